"""Per-kernel HBM figures of the kernels tools/bench_more.py runs, from three rocprofv3 runs of it
(`--kernel-trace`, `--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`) and the JSON line it prints.

  python profiles/summarize_more.py <kernel_trace.csv> <fetch counter_collection.csv> \
      <write counter_collection.csv> <bench_more.json> <out.json>

Same columns as the other *_ops_pmc.json files: algorithmic bytes per launch, counter bytes per
launch ((2 x FETCH_SIZE + WRITE_SIZE) x 1024; the factor 2 is the gfx950 FETCH_SIZE correction),
time of the heaviest launch, achieved = algorithmic bytes / time against the 8 TB/s HBM peak.
"""
import collections
import csv
import json
import re
import sys

PEAK = 8.0e12


def short(name):
    m = re.search(r"tmp::(\w+)", name)
    return m.group(1) if m else None


def main():
    trace, fetch, write, algo_json, out = sys.argv[1:6]
    with open(algo_json) as f:
        algo = json.loads(f.readlines()[-1])["algorithmic_bytes_per_launch"]
    heavy, launches = {}, collections.Counter()
    with open(trace) as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            if k in algo:
                ns = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                launches[k] += 1
                heavy[k] = max(heavy.get(k, 0), ns)
    ctr = {}
    for path, name in ((fetch, "FETCH_SIZE"), (write, "WRITE_SIZE")):
        per = collections.defaultdict(list)
        with open(path) as f:
            for r in csv.DictReader(f):
                k = short(r["Kernel_Name"])
                if r["Counter_Name"] == name and k in algo:
                    per[k].append(float(r["Counter_Value"]))
        ctr[name] = {k: max(v) for k, v in per.items()}   # the heaviest launch, like the time
    res = {}
    for k, b in sorted(algo.items()):
        if k not in heavy:
            continue
        ms = heavy[k] / 1e6
        fk, wk = ctr["FETCH_SIZE"].get(k, 0.0), ctr["WRITE_SIZE"].get(k, 0.0)
        hbm = (2 * fk + wk) * 1024
        res[k] = dict(launches=launches[k], launch_ms=round(ms, 4), algorithmic_bytes_per_launch=int(b),
                      fetch_size_kib_per_launch=round(fk, 1), write_size_kib_per_launch=round(wk, 1),
                      hbm_bytes_per_launch=int(hbm), traffic_over_algorithmic=round(hbm / b, 2) if b else None,
                      achieved_gb_s=round(b / ms / 1e6, 1), frac_of_hbm_peak=round(b / (ms * 1e-3) / PEAK, 4))
    with open(out, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    for k, v in res.items():
        print("%-24s %8.3f ms  %7.1f GB/s  frac %.3f  traffic/algorithmic %s" % (k, v["launch_ms"], v["achieved_gb_s"],
                                                                                v["frac_of_hbm_peak"], v["traffic_over_algorithmic"]))


if __name__ == "__main__":
    main()
