"""The kernels the headline and secondary benches do not reach, each run once or twice on inputs
that take its path: long frames of withRangeStats (wide_*), withRangeStats over (key, second) cells
(cell_runs / cell_rows), interpolate (interp_*), the Arrow import's conversion kernel, the write
layout (event_cols), the match-record gather, the one-sweep join of ordered tables (sj_*), and the
narrowing / widening of the columns a key shuffle sends (wire_rows4).
Prints one JSON line: per kernel the algorithmic bytes one launch moves (every input read once,
every output written once) -- what profiles/summarize_more.py sets against rocprofv3's counters.

  python tools/bench_more.py [--rows 100000000]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tempo_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=100_000_000)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    n = args.rows
    algo = {}

    # ---- long frames: one series, 100 rows per second, 600 s frames = 60 000 rows (region holds 1280)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    ts = bench.BASE_US + torch.cumsum(torch.empty(n, dtype=torch.float64, device=dev).exponential_(100.0, generator=g), 0).mul_(1e6).to(torch.int64)
    x = 100 + torch.randn(n, generator=g, device=dev, dtype=torch.float64)
    ops.range_stats(None, ts, x, None, 1, 600)
    algo["wide_chunk_kernel"] = n * (8 + 8 + 8 + 8 + 4 + 4 * 8)      # word + x in; two prefixes, count, four extremes out
    algo["wide_rows_kernel"] = n * (8 + 1 + 61 + 2 * 64)            # word + flag in, seven outputs; two random 64-byte prefix rows
    algo["wide_pivot_kernel"] = (n // 65536 + 1) * 64 * 16
    # ---- (key, second) cells: 8 keys, 50 rows per second and key, time-ordered
    key = torch.randint(0, 8, (n,), generator=g, device=dev, dtype=torch.int32)
    ts2 = bench.BASE_US + torch.sort(torch.randint(0, (n // 400) * 1_000_000, (n,), generator=g, device=dev, dtype=torch.int64)).values
    r = ops.range_stats(key, ts2, x, None, 8, 600, any_order=True)
    assert bool((r["perm"][:1000] == torch.arange(1000, device=dev, dtype=torch.int32)).all()), "cells path not taken"
    cells = 8 * (n // 400)
    algo["cell_rows_kernel"] = n * (4 + 8 + 8 + 61 + 4)            # key, ts, x in; seven outputs + row id out (+ 1-2 cell records, cached)
    algo["cell_runs_kernel"] = cells * (8 * 6 + 2 * 64)
    del r
    # ---- interpolate: minute samples of 9 series upsampled to seconds
    m = max(n // 60, 1000)
    sk = torch.arange(m, device=dev, dtype=torch.int32) % 9
    st = bench.BASE_US + (torch.arange(m, device=dev, dtype=torch.int64) // 9) * 60_000_000
    o = torch.argsort(sk, stable=True)
    sk, st = sk[o].contiguous(), st[o].contiguous()
    sv = torch.randn(m, generator=g, device=dev, dtype=torch.float64)
    out = ops.interpolate(sk, st, 9, 1_000_000, "linear", [sv], [None])
    algo["interp_fill_kernel"] = out["n"] * (21 + 10) + m * 21
    algo["interp_count_kernel"] = m * 12
    del out
    # ---- Arrow import with conversion: float32 with NULLs -> float64 + byte mask
    import pyarrow as pa
    k = min(n, 50_000_000)
    vals = np.random.default_rng(1).normal(size=k).astype(np.float32)
    arr = pa.array(vals, mask=np.random.default_rng(2).random(k) < 0.1)
    ops.arrow_import(arr, "f64")
    algo["arrow_convert_kernel"] = k * (4 + 8 + 1) + k // 8
    # ---- write layout
    ops.write_layout(key, ts2, 8)
    algo["event_cols_kernel"] = n * (8 + 4 + 8)
    # ---- gather of joined values from match records
    idx = torch.randint(-1, n, (n,), generator=g, device=dev, dtype=torch.int32)
    ops.gather(x, idx)
    algo["gather_kernel"] = n * (4 + 8 + 8 + 1)
    del idx
    # ---- one-sweep join of ordered tables (half the rows per side)
    h = n // 2
    q = bench.gen_trades_quotes(dev, h, h, 10_000, 4567)
    for side, cols in (("l", ["key_l", "ts_l", "trade_pr"]), ("r", ["key_r", "ts_r", "bid", "ask", "bid_valid", "ask_valid"])):
        o = torch.argsort(q["ts_" + side], stable=True)
        o = o[torch.argsort(q["key_" + side][o], stable=True)]
        for c in cols:
            q[c] = q[c][o].contiguous()
    j = ops.asof_join_table(q["key_l"], q["ts_l"], q["key_r"], q["ts_r"], 10_000, [q["trade_pr"]],
                            [q["ts_r"], q["bid"], q["ask"]], [None, q["bid_valid"], q["ask_valid"]],
                            skip_nulls=True, ts_partition_val=10, fraction=0.1, left_by_reference=True)
    assert j["in_place"]
    algo["sj_merge_kernel"] = h * (12 + 14 + 16 + 3 * 9)            # left key/ts; right key/ts/2 masks; bid, ask; 3 x (value + flag) out
    algo["sj_tile_last_kernel"] = h * 2
    algo["sj_bounds_kernel"] = (h // 512) * 8
    del j, q
    # ---- the wire form of the key shuffle: key id + NULL flag in one byte, the time as a 5-byte offset
    from tempo_amd.dist import WirePlan
    flag = (torch.rand(n, generator=g, device=dev) >= 0.005).to(torch.uint8)
    plan = WirePlan(3, [1], {0: (int(ts2[0]), 5)})
    wire = plan.encode(key, [ts2, flag])
    back_key, back = plan.decode(wire, [torch.int64, torch.uint8])
    assert torch.equal(back_key, key) and torch.equal(back[0], ts2) and torch.equal(back[1], flag)
    algo["wire_rows4_kernel"] = n * (4 + 8 + 1 + 1 + 4 + 1)        # either way: key, ts, flag on one side; byte, low word, high byte on the other
    torch.cuda.synchronize()
    print(json.dumps(dict(rows=n, algorithmic_bytes_per_launch=algo)))


if __name__ == "__main__":
    main()
