"""withRangeStats over one, two, three and four metric columns of the headline tick table shape
(tmp_range_stats_cols: one ordering, the columns travelling with the rows, one statistics kernel per
column): ms per call and per kernel.

  python tools/bench_stats_cols.py [--rows 100000000]
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
    ap.add_argument("--rows", type=int, default=100_000_000)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    tk = bench.gen_ticks(dev, args.rows, 100, 2345)
    cols = [tk["x"], tk["x"] * 1.5 + 3.0, tk["x"] - 7.0, tk["x"] * 0.25]
    _lib.profile_enable(True)
    out = {"rows": args.rows}
    for nc in (1, 2, 3, 4):
        f = lambda: ops.range_stats_cols(tk["key"], tk["ts"], cols[:nc], [tk["x_valid"]] + [None] * (nc - 1), 100, 600)
        f()
        torch.cuda.synchronize()
        _lib.profile_reset()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            o = f()
            del o
        e1.record()
        torch.cuda.synchronize()
        kern = {}
        for name, _, ms in _lib.profile_read():
            kern[name] = kern.get(name, 0.0) + ms / args.reps
        out["cols_%d" % nc] = dict(ms_per_call=round(e0.elapsed_time(e1) / args.reps, 3),
                                   kernels={k: round(v, 3) for k, v in sorted(kern.items(), key=lambda kv: -kv[1])[:6]})
    out["ratio_3_to_1"] = round(out["cols_3"]["ms_per_call"] / out["cols_1"]["ms_per_call"], 2)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
