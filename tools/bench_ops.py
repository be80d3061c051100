"""Secondary operators on an accelerometer-shaped table (one MI355X, synthetic data
resident in HBM).  Prints one JSON line per operator: rows/s, ms per call and the per-kernel
device time from the library's own launch events.

  python tools/bench_ops.py [--rows 500000000] [--reps 3] [--no-profile]

--no-profile leaves the library's own launch events off (for runs under rocprofv3, which times
and counts the kernels itself).
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tempo_amd import _lib, ops  # noqa: E402

BASE_US = 1_577_836_800_000_000


def accelerometer(dev, n, users=9, seed=3456):
    """9 users, 5 ms sampling with +-1 ms jitter, three fp64 channels, rows in time order."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    per = n // users
    n = per * users
    slot = torch.arange(per, device=dev, dtype=torch.int64).repeat_interleave(users)
    key = torch.arange(users, device=dev, dtype=torch.int32).repeat(per)
    jitter = torch.randint(-1000, 1001, (n,), generator=g, device=dev, dtype=torch.int64)
    ts = BASE_US + slot * 5000 + jitter
    cols = [torch.randn(n, generator=g, device=dev, dtype=torch.float64) * s + m
            for m, s in ((-1.70, 3.86), (0.16, 1.55), (8.93, 2.26))]
    return key, ts, cols, users


REPS = 3


def timed(fn, reps=None):
    reps = reps or REPS
    fn()
    torch.cuda.synchronize()
    _lib.profile_reset()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
        del out
    e1.record()
    torch.cuda.synchronize()
    kern = {}
    for name, _, ms in _lib.profile_read():
        kern[name] = kern.get(name, 0.0) + ms / reps
    return e0.elapsed_time(e1) / reps, {k: round(v, 3) for k, v in sorted(kern.items(), key=lambda kv: -kv[1])[:6]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=500_000_000)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--no-profile", action="store_true")
    args = ap.parse_args()
    global REPS
    REPS = args.reps
    dev = torch.device("cuda", 0)
    _lib.profile_enable(not args.no_profile)
    key, ts, cols, nk = accelerometer(dev, args.rows)
    n = ts.numel()
    torch.cuda.synchronize()

    def report(name, ms, kern, rows, **extra):
        print(json.dumps(dict(op=name, rows=rows, ms=round(ms, 2), rows_per_s=round(rows / ms * 1e3),
                              kernels_ms=kern, **extra)), flush=True)

    ms, kern = timed(lambda: ops.resample(key, ts, nk, 60_000_000, "mean", cols, [None] * 3))
    report("resample(min, mean) 3 cols", ms, kern, n)
    ms, kern = timed(lambda: ops.resample(key, ts, nk, 86_400_000_000, "mean", cols, [None] * 3))
    report("resample(day, mean) 3 cols", ms, kern, n)
    ms, kern = timed(lambda: ops.resample(key, ts, nk, 60_000_000, "floor", cols, [None] * 3))
    report("resample(min, floor) 3 cols", ms, kern, n)
    ms, kern = timed(lambda: ops.grouped_stats(key, ts, nk, 60_000_000, cols[:1], [None]))
    report("withGroupedStats(1 min) 1 col", ms, kern, n)
    ms, kern = timed(lambda: ops.ema(key, ts, cols[0], None, nk, window=30))
    report("EMA(window=30)", ms, kern, n)
    m = min(n, 100_000_000)
    ms, kern = timed(lambda: ops.lookback(key[:m], ts[:m], [c[:m] for c in cols], [None] * 3, nk, 10))
    report("withLookbackFeatures(L=10, F=3)", ms, kern, m)
    # the shuffle's device side: rows routed to 8 owners, key + ts + one fp64 + one flag column moved
    flag = (cols[1] > 0).to(torch.uint8)
    ms, kern = timed(lambda: ops.partition_move(key, [key, ts, cols[0], flag], 8))
    report("shuffle routing (8 owners; key, ts, fp64, flag)", ms, kern, n)
    del flag
    # interpolate: one-minute means upsampled to one second (60x)
    r = ops.resample(key, ts, nk, 60_000_000, "mean", cols[:1], [None])
    sk, st, sv = r["key"].contiguous(), r["start"].contiguous(), r["f64"][0].contiguous()
    ms, kern = timed(lambda: ops.interpolate(sk, st, nk, 1_000_000, "linear", [sv], [None]))
    out_rows = ops.interpolate(sk, st, nk, 1_000_000, "linear", [sv], [None])["n"]
    report("interpolate(linear) 1 min -> 1 s", ms, kern, out_rows, sampled_rows=int(st.numel()))


if __name__ == "__main__":
    main()
