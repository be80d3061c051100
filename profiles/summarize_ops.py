"""Per-kernel HBM fractions of the secondary operators, from three rocprofv3 runs of
tools/bench_ops.py (`--kernel-trace --stats`, `--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`).

  python profiles/summarize_ops.py <kernel_stats.csv> <fetch counter_collection.csv> \
      <write counter_collection.csv> <rows> <out.json>

For every kernel: algorithmic bytes per launch (every input read once, every output written once;
the per-row figures below times the rows the launch handles), the counter bytes per launch
((2 x FETCH_SIZE + WRITE_SIZE) x 1024, as in summarize_pmc.py), the average launch time of the
trace run, and achieved = algorithmic bytes / time against the 8 TB/s HBM peak.
"""
import collections
import csv
import json
import re
import sys

PEAK = 8.0e12

# kernel -> (algorithmic bytes per row, which launches: rows handled relative to --rows)
# accelerometer table: int32 key, int64 ts, three fp64 channels; sort words are 8 bytes, row ids 4.
ALGO = {
    # EMA: sort word + carried metric in, EMA + row id out
    "ema_kernel": 8 + 8 + 8 + 4,
    # resample(mean, 3 cols): row id + 3 gathered metrics in (outputs are per bucket: negligible)
    "bucket_reduce_kernel": 4 + 3 * 8,
    "bucket_count_kernel": 8,
    "bucket_starts_kernel": 8,
    # withGroupedStats(1 col): row id + metric in
    "bucket_stats_kernel": 4 + 8,
    # lookback(L=10, F=3) on min(rows, 1e8) rows: word + id + 3 metrics in, 10 x 3 fp64 + row id out
    "lookback_write_kernel": 8 + 4 + 3 * 8 + 10 * 3 * 8 + 4,
    "lookback_count_kernel": 8,
    "lookback_slots_kernel": 8 + 4,
    # shuffle routing: key in; then key + (key, ts, fp64, flag) in and out
    "route_count_kernel": 4,
    "route_move_kernel": 4 + 2 * (4 + 8 + 8 + 1),
}
LOOKBACK_ROWS = 100_000_000


def short(name):
    m = re.search(r"tmp::(\w+)", name)
    return m.group(1) if m else None


def counters(path, counter):
    per = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                k = short(r["Kernel_Name"])
                if k:
                    per[k].append(float(r["Counter_Value"]))
    return per


def main():
    stats, fetch, write, rows, out = sys.argv[1:6]
    rows = int(rows)
    F, W = counters(fetch, "FETCH_SIZE"), counters(write, "WRITE_SIZE")
    res = {}
    with open(stats) as f:
        for r in csv.DictReader(f):
            k = short(r["Name"])
            if k not in ALGO:
                continue
            n = min(rows, LOOKBACK_ROWS) if k.startswith("lookback") else rows
            e = res.setdefault(k, dict(calls=0, total_ms=0.0))
            e["calls"] += int(r["Calls"])
            e["total_ms"] += int(r["TotalDurationNs"]) / 1e6
            e["rows_per_launch"] = n
            e["algorithmic_bytes_per_launch"] = ALGO[k] * n
    for k, e in res.items():
        e["avg_launch_ms"] = round(e.pop("total_ms") / e["calls"], 4)
        if F.get(k) and W.get(k):
            f_kib = sum(F[k]) / len(F[k])
            w_kib = sum(W[k]) / len(W[k])
            e["fetch_size_kib_per_launch"] = round(f_kib, 1)
            e["write_size_kib_per_launch"] = round(w_kib, 1)
            e["hbm_bytes_per_launch"] = int((2 * f_kib + w_kib) * 1024)
            e["traffic_over_algorithmic"] = round(e["hbm_bytes_per_launch"] / e["algorithmic_bytes_per_launch"], 2)
        e["achieved_gb_s"] = round(e["algorithmic_bytes_per_launch"] / e["avg_launch_ms"] / 1e6, 1)
        e["frac_of_hbm_peak"] = round(e["algorithmic_bytes_per_launch"] / (e["avg_launch_ms"] * 1e-3) / PEAK, 4)
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k in sorted(res):
        print(f"{k:26s} {res[k]}")


if __name__ == "__main__":
    main()
