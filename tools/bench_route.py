"""Device side of the key shuffle on its own: ops.partition_move of the bench's three table shapes to
1, 2, 4 and 8 owners (hashed), timed with events.  python tools/bench_route.py [--rows N]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tempo_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=500_000_000)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    n = a.rows
    g = torch.Generator(device=dev).manual_seed(7)
    key = torch.randint(0, 10_000, (n,), dtype=torch.int32, device=dev, generator=g)
    ts = torch.randint(0, 1 << 40, (n,), dtype=torch.int64, device=dev, generator=g)
    x = torch.rand(n, dtype=torch.float64, device=dev, generator=g)
    f = (x > 0.5).to(torch.uint8)
    shapes = {"trades (key, ts, f64)": [key, ts, x], "quotes (key, ts, 2 f64, 2 flags)": [key, ts, x, x, f, f],
              "ticks (key, ts, f64, flag)": [key, ts, x, f]}
    for name, cols in shapes.items():
        for parts in (1, 2, 4, 8):
            ops.partition_move(key, cols, parts)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                ops.partition_move(key, cols, parts)
            e1.record()
            torch.cuda.synchronize()
            print(json.dumps({"table": name, "parts": parts, "rows": n, "ms": round(e0.elapsed_time(e1) / a.reps, 3)}),
                  flush=True)


if __name__ == "__main__":
    main()
