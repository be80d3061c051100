"""withRangeStats alone on the headline tick table (bench.py's), in three row orders: time-ordered
feed, row-permuted, and `--runs N` time-ordered runs back to back (what a rank holds after the key
shuffle).  Prints ms per call and the per-kernel device time of each.

  python tools/bench_stats.py [--rows 1000000000] [--runs 8]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tempo_amd import _lib, ops  # noqa: E402


ANY_ORDER = False


def timed(t, reps=2):
    f = lambda: ops.range_stats(t["key"], t["ts"], t["x"], t["x_valid"], 100, 600, any_order=ANY_ORDER)
    f()
    torch.cuda.synchronize()
    _lib.profile_reset()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        o = f()
        del o
    e1.record()
    torch.cuda.synchronize()
    kern = {}
    for name, _, ms in _lib.profile_read():
        kern[name] = kern.get(name, 0.0) + ms / reps
    return round(e0.elapsed_time(e1) / reps, 3), {k: round(v, 3) for k, v in sorted(kern.items(), key=lambda kv: -kv[1])[:9]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000_000)
    ap.add_argument("--runs", type=int, default=8)
    ap.add_argument("--density", type=float, default=1.0, help="ticks per second and symbol (bench.py: 1)")
    ap.add_argument("--any-order", action="store_true", help="let the library return the rows in any order")
    ap.add_argument("--orders", default="time_ordered,runs,permuted", help="which row orders to time")
    args = ap.parse_args()
    global ANY_ORDER
    ANY_ORDER = args.any_order
    dev = torch.device("cuda", 0)
    _lib.profile_enable(True)
    n = args.rows
    tk = bench.gen_ticks(dev, n, 100, 2345, shards=1.0 / args.density)
    out = {}
    orders = args.orders.split(",")
    if "time_ordered" in orders:
        out["time_ordered"] = timed(tk)
    # N runs: rank r of N keeps every N-th block of 1/N of the rows... simplest faithful stand-in:
    # split the feed into N interleaved sub-feeds (row i goes to run i mod N) and lay them back to back
    idx = torch.arange(n, device=dev).view(-1, args.runs).t().reshape(-1) if n % args.runs == 0 else None
    if idx is not None and "runs" in orders:
        runs = {k: v[idx] for k, v in tk.items()}
        del idx
        out["runs_%d" % args.runs] = timed(runs)
        del runs
    if "permuted" in orders:
        p = torch.randperm(n, device=dev)
        perm = {k: v[p] for k, v in tk.items()}
        del p
        out["permuted"] = timed(perm)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
