"""Per-kernel SQ counter totals from one `rocprofv3 --pmc SQ_...` pass of bench.py (one step).

usage: summarize_sq.py <counter_collection.csv> <kernel_stats.csv> <out.json>
valu_util = SQ_INSTS_VALU / (SIMDs x kernel cycles / 4): a wave64 VALU instruction occupies its
SIMD for 4 cycles, so this is the share of the chip's vector issue slots the kernel filled
(1024 SIMDs, 2.4 GHz).  Kernel time is the rocprof average x launches of the step.
"""
import collections
import csv
import json
import sys

from summarize_pmc import short as short_name


def main(counter_csv, stats_csv, out_json):
    totals = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(counter_csv)):
        name = short_name(r["Kernel_Name"])
        if name is None:  # not a kernel of this library (the data generator's)
            continue
        totals[name][r["Counter_Name"]] += float(r["Counter_Value"])
        if (r["Dispatch_Id"], name) not in seen:
            seen.add((r["Dispatch_Id"], name))
            launches[name] += 1
    avg_ns = {}
    for r in csv.DictReader(open(stats_csv)):
        n = short_name(r["Name"])
        if n is None:
            continue
        calls, tot = float(r["Calls"]), float(r["TotalDurationNs"])
        if n in avg_ns:
            calls, tot = calls + avg_ns[n][0], tot + avg_ns[n][1]
        avg_ns[n] = (calls, tot)
    out = {}
    for name, c in sorted(totals.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
        rec = {k: v for k, v in c.items()}
        rec["launches_per_step"] = launches[name]
        if name in avg_ns and avg_ns[name][0]:
            ms = avg_ns[name][1] / avg_ns[name][0] * launches[name] / 1e6
            rec["ms_per_step_rocprof"] = round(ms, 3)
            rec["valu_util"] = round(c.get("SQ_INSTS_VALU", 0) / (1024 * 2.4e9 * ms / 1e3 / 4), 3)
        out[name] = rec
    json.dump(out, open(out_json, "w"), indent=1)
    for name, rec in out.items():
        if rec.get("ms_per_step_rocprof", 0) > 1:
            print("%-32s %7.2f ms  valu_util %.2f" % (name, rec["ms_per_step_rocprof"], rec.get("valu_util", 0)))


if __name__ == "__main__":
    main(*sys.argv[1:4])
