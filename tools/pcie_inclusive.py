"""PCIe-inclusive rate of the headline path: inputs start in pinned host memory, results end in
pinned host memory.  One JSON line.  (bench.py's `value` is the device-resident rate.)

  python tools/pcie_inclusive.py [--rows 200000000]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tempo_amd import ops  # noqa: E402


def to_host(d):
    return {k: torch.empty(v.shape, dtype=v.dtype, pin_memory=True).copy_(v) for k, v in d.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=200_000_000, help="rows per operator")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    n = args.rows
    tq_h = to_host(bench.gen_trades_quotes(dev, n // 2, n // 2, 10_000, 4567))
    tk_h = to_host(bench.gen_ticks(dev, n, 100, 2345))
    torch.cuda.empty_cache()
    in_bytes = sum(v.numel() * v.element_size() for d in (tq_h, tk_h) for v in d.values())

    def run():
        q = {k: v.to(dev, non_blocking=True) for k, v in tq_h.items()}
        j = ops.asof_join_table(q["key_l"], q["ts_l"], q["key_r"], q["ts_r"], 10_000, [q["trade_pr"]],
                                [q["ts_r"], q["bid"], q["ask"]], [None, q["bid_valid"], q["ask_valid"]],
                                skip_nulls=True, ts_partition_val=10, fraction=0.1)
        outs = [j["perm"], j["key"], j["ts"], j["left"][0]] + [t for pair in j["right"] for t in pair]
        t = {k: v.to(dev, non_blocking=True) for k, v in tk_h.items()}
        r = ops.range_stats(t["key"], t["ts"], t["x"], t["x_valid"], 100, 600)
        outs += list(r.values())
        host = [torch.empty(o.shape, dtype=o.dtype, pin_memory=True) for o in outs]
        for h, o in zip(host, outs):
            h.copy_(o, non_blocking=True)
        torch.cuda.synchronize()
        return sum(h.numel() * h.element_size() for h in host)

    out_bytes = run()
    t0 = time.perf_counter()
    reps = 2
    for _ in range(reps):
        run()
    dt = (time.perf_counter() - t0) / reps
    print(json.dumps(dict(rows_per_operator=n, seconds=round(dt, 4), rows_per_s=round(2 * n / dt),
                          host_to_device_GB=round(in_bytes / 1e9, 2), device_to_host_GB=round(out_bytes / 1e9, 2),
                          pcie_GBps=round((in_bytes + out_bytes) / dt / 1e9, 1))))


if __name__ == "__main__":
    main()
