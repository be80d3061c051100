"""Per-kernel HBM fractions of the secondary operators, from three rocprofv3 runs of
tools/bench_ops.py (`--kernel-trace --stats`, `--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`).

  python profiles/summarize_ops.py <kernel_trace.csv> <fetch counter_collection.csv> \
      <write counter_collection.csv> <rows> <out.json>

For every kernel: algorithmic bytes per launch (every input read once, every output written once;
the per-row figures below times the rows the launch handles), the counter bytes per launch
((2 x FETCH_SIZE + WRITE_SIZE) x 1024, as in summarize_pmc.py), the launch time in the trace run
(heaviest launch of the kernel), and achieved = algorithmic bytes / time against the 8 TB/s HBM peak.
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
    "ema_kernel": 8 + 8 + 8,  # sort word + metric in, EMA out (the row ids are left in out_perm by the ordering step)
    # resample(mean, 3 cols): the 3 metric columns, already in sorted order (outputs are per bucket)
    "bucket_reduce_kernel": 3 * 8,
    # resample(mean) without ordering: key + ts + the 3 metric columns in (outputs are per bucket)
    "dense_mean_kernel": 4 + 8 + 3 * 8,
    # withGroupedStats(1 col) without ordering: key + ts + the metric column in, per sweep
    "dense_stats_kernel": 4 + 8 + 8,
    # resample(floor) without ordering: key + ts in, per sweep
    "dense_pick_kernel": 4 + 8,
    "bucket_count_kernel": 8,
    "bucket_starts_kernel": 8,
    # withGroupedStats(1 col): the metric column in sorted order
    "bucket_stats_kernel": 8,
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
    trace, fetch, write, rows, out = sys.argv[1:6]
    rows = int(rows)
    F, W = counters(fetch, "FETCH_SIZE"), counters(write, "WRITE_SIZE")
    # per kernel the launches of the trace run; the operators are called with different column
    # counts and bucket sizes, so the figures are those of the heaviest launch (the one the
    # per-row byte counts above describe), not an average over unlike launches
    dur = collections.defaultdict(list)
    with open(trace) as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            if k in ALGO:
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    res = {}
    for k, ms in dur.items():
        n = min(rows, LOOKBACK_ROWS) if k.startswith("lookback") else rows
        e = dict(launches=len(ms), rows_per_launch=n, algorithmic_bytes_per_launch=ALGO[k] * n,
                 launch_ms=round(max(ms), 4))
        if F.get(k) and W.get(k):
            f_kib, w_kib = max(F[k]), max(W[k])
            e["fetch_size_kib_per_launch"] = round(f_kib, 1)
            e["write_size_kib_per_launch"] = round(w_kib, 1)
            e["hbm_bytes_per_launch"] = int((2 * f_kib + w_kib) * 1024)
            e["traffic_over_algorithmic"] = round(e["hbm_bytes_per_launch"] / e["algorithmic_bytes_per_launch"], 2)
        e["achieved_gb_s"] = round(e["algorithmic_bytes_per_launch"] / e["launch_ms"] / 1e6, 1)
        e["frac_of_hbm_peak"] = round(e["algorithmic_bytes_per_launch"] / (e["launch_ms"] * 1e-3) / PEAK, 4)
        res[k] = e
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k in sorted(res):
        print(f"{k:26s} {res[k]}")


if __name__ == "__main__":
    main()
