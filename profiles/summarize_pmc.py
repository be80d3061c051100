"""Turn rocprofv3 output of `bench.py` into the per-kernel summaries kept in this directory.

  python profiles/summarize_pmc.py <kernel_stats.csv> <fetch counter_collection.csv> \
      <write counter_collection.csv> <out.json>

HBM traffic per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 bytes: rocprofv3 reports both
counters in KiB, and on gfx950 FETCH_SIZE tallies the 128-byte requests of coalesced streaming
reads at 64 bytes (checked here on radix_hist_kernel, which reads exactly 8 bytes per row:
FETCH_SIZE x 1024 comes out at 4 bytes per row).  Kernels dominated by random 8- or 32-byte
reads (asof_merge_kernel) are over-counted by that factor on their random part; the raw
counters are kept next to the corrected figure.
"""
import collections
import csv
import json
import re
import sys


def short(name):
    m = re.search(r"tmp::(\w+)(<([^>]*)>)?", name)
    if not m:
        return None
    base, args = m.group(1), (m.group(3) or "")
    if base == "radix_scatter_kernel":
        parts = [a.strip() for a in args.split(",")]
        npay = parts[2]
        if len(parts) > 6 and parts[6] in ("true", "1"):  # the pass that builds the sort words itself
            base += "+words32" if len(parts) > 7 and parts[7] in ("true", "1") else "+words"
        return base + ("+carried" if npay.startswith("-") else "+payload" if npay != "0" else "")
    if base == "radix_hist_kernel":
        gen = args.split(",")[-1].strip()
        return base + ("+keys" if gen == "3" else "+words" if gen in ("true", "1", "2") else "")
    if base == "range_stats_kernel":
        parts = [a.strip() for a in args.split(",")]
        if len(parts) > 3 and parts[3] in ("true", "1"):   # reads 32-bit sort words
            return "range_stats_kernel<T>/words32"
        if len(parts) > 2 and parts[2] in ("true", "1"):   # builds the words from the table's columns
            return "range_stats_kernel<T>+words"
        return "range_stats_kernel<T>"
    if base == "gather_kernel":
        return "gather_kernel<%s>" % args.replace("unsigned long", "uint64_t")
    return base


def counters(path, counter):
    per = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            k = short(r["Kernel_Name"])
            if k:
                per[k].append(float(r["Counter_Value"]))
    return per


def main():
    stats, fetch, write, out = sys.argv[1:5]
    F, W = counters(fetch, "FETCH_SIZE"), counters(write, "WRITE_SIZE")
    timing = {}
    with open(stats) as f:
        for r in csv.DictReader(f):
            k = short(r["Name"])
            if not k:
                continue
            t = timing.setdefault(k, dict(calls=0, total_ns=0))
            t["calls"] += int(r["Calls"])
            t["total_ns"] += int(r["TotalDurationNs"])
    res = {}
    for k in sorted(set(F) | set(W)):
        f_kib = sum(F[k]) / max(len(F[k]), 1)
        w_kib = sum(W[k]) / max(len(W[k]), 1)
        e = dict(launches_per_step=len(F[k]), fetch_size_kib_per_launch=round(f_kib, 1),
                 write_size_kib_per_launch=round(w_kib, 1),
                 hbm_bytes_per_launch=int((2 * f_kib + w_kib) * 1024))
        if k in timing:
            e["avg_launch_ms_rocprof"] = round(timing[k]["total_ns"] / timing[k]["calls"] / 1e6, 4)
            e["calls_rocprof"] = timing[k]["calls"]
        res[k] = e
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k, e in res.items():
        print(f"{k:32s} {e}")


if __name__ == "__main__":
    main()
