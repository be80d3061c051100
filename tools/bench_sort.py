"""The ordering step alone (tmp_sort_order) on one side of the headline join: Zipf(1.2) keys over
10 000 symbols, uniform microsecond timestamps over a day, rows in random order.  Prints one JSON
line with ms per call and the per-kernel device time.

  python tools/bench_sort.py [--rows 500000000] [--keys 10000] [--reps 3]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tempo_amd import _lib, ops  # noqa: E402

BASE_US = 1_596_240_000_000_000
DAY_US = 86_400_000_000


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=500_000_000)
    ap.add_argument("--keys", type=int, default=10_000)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--check", action="store_true", help="verify the permutation against torch (small sizes)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(4567)
    n = args.rows
    w = 1.0 / torch.arange(1, args.keys + 1, dtype=torch.float64, device=dev) ** 1.2
    cdf = torch.cumsum(w / w.sum(), 0)
    key = torch.empty(n, dtype=torch.int32, device=dev)
    for s in range(0, n, 1 << 26):
        m = min(1 << 26, n - s)
        u = torch.rand(m, generator=g, device=dev, dtype=torch.float64)
        key[s:s + m] = torch.searchsorted(cdf, u).clamp_(max=args.keys - 1).to(torch.int32)
    ts = BASE_US + torch.randint(0, DAY_US, (n,), generator=g, device=dev, dtype=torch.int64)
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    perm = ops.sort_order(key, ts, args.keys)
    torch.cuda.synchronize()
    if args.check:
        word = (key.to(torch.int64) << 40) | (ts - BASE_US)
        want = torch.sort(word, stable=True).indices.to(torch.int32)
        assert torch.equal(perm, want), "permutation differs from torch.sort"
    del perm
    _lib.profile_reset()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        p = ops.sort_order(key, ts, args.keys)
        del p
    e1.record()
    torch.cuda.synchronize()
    kern = {}
    for name, _, ms in _lib.profile_read():
        k = kern.setdefault(name, [0, 0.0])
        k[0] += 1
        k[1] += ms / args.reps
    print(json.dumps(dict(rows=n, keys=args.keys, ms_per_call=round(e0.elapsed_time(e1) / args.reps, 3),
                          checked=bool(args.check),
                          kernels={k: dict(launches=v[0] // args.reps, ms=round(v[1], 3))
                                   for k, v in sorted(kern.items(), key=lambda kv: -kv[1][1])})))


if __name__ == "__main__":
    main()
