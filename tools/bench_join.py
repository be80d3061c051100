"""The headline join alone (bench.py's trades x quotes, skew mode, fused table output): ms per call
and per-kernel device time.  Also the command to put under rocprofv3 when only the join's kernels
are of interest.

  python tools/bench_join.py [--rows 1000000000] [--reps 3]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tempo_amd import _lib, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000_000, help="left + right rows")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--order", choices=["random", "sorted", "time"], default="random",
                    help="row order of both sides: as generated (random), by (key, ts), or by ts alone")
    ap.add_argument("--no-skew", action="store_true", help="plain join: no tsPartitionVal brackets")
    ap.add_argument("--copy-left", action="store_true",
                    help="have the join write key / ts / left payload of the result even when the left table is "
                         "in order already (default: such a table is used by reference)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    n = args.rows // 2
    q = bench.gen_trades_quotes(dev, n, n, 10_000, 4567)
    if args.order != "random":
        for side, cols in (("l", ["key_l", "ts_l", "trade_pr"]), ("r", ["key_r", "ts_r", "bid", "ask", "bid_valid", "ask_valid"])):
            o = torch.argsort(q["ts_" + side], stable=True)
            if args.order == "sorted":
                o = o[torch.argsort(q["key_" + side][o], stable=True)]
            for c in cols:
                q[c] = q[c][o].contiguous()
            del o
    torch.cuda.synchronize()
    if not args.no_profile:
        _lib.profile_enable(True)

    def run():
        return ops.asof_join_table(q["key_l"], q["ts_l"], q["key_r"], q["ts_r"], 10_000, [q["trade_pr"]],
                                   [q["ts_r"], q["bid"], q["ask"]], [None, q["bid_valid"], q["ask_valid"]],
                                   skip_nulls=True, ts_partition_val=None if args.no_skew else 10, fraction=0.1,
                                   left_by_reference=not args.copy_left)

    in_place = run()["in_place"]
    torch.cuda.synchronize()
    _lib.profile_reset()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        j = run()
        del j
    e1.record()
    torch.cuda.synchronize()
    kern = {}
    if not args.no_profile:
        for name, _, ms in _lib.profile_read():
            k = kern.setdefault(name, [0, 0.0])
            k[0] += 1
            k[1] += ms / args.reps
    print(json.dumps(dict(rows=args.rows, order=args.order, left_in_place=in_place, ms_per_call=round(e0.elapsed_time(e1) / args.reps, 3),
                          kernels={k: dict(launches=v[0] // args.reps, ms=round(v[1], 3))
                                   for k, v in sorted(kern.items(), key=lambda kv: -kv[1][1])})))


if __name__ == "__main__":
    main()
