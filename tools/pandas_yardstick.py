"""A familiar CPU yardstick next to the oracle-based cpu_baseline: single-thread pandas on the same
kind of tables -- merge_asof(by=symbol) for the AS-OF join and groupby(symbol).rolling("600s") for
withRangeStats -- at 10^6 and 10^7 rows.  Not the reference (no PySpark here); prints one JSON line.

  python tools/pandas_yardstick.py [--rows 1000000 10000000]
"""
import argparse
import json
import os
import time

import numpy as np
import pandas as pd

BASE = np.datetime64("2020-08-01T00:00:00", "us")


def tables(n, seed):
    rng = np.random.default_rng(seed)
    nk = 10_000
    w = 1.0 / np.arange(1, nk + 1) ** 1.2
    cdf = np.cumsum(w / w.sum())
    sym = lambda m: np.searchsorted(cdf, rng.random(m)).clip(0, nk - 1).astype(np.int32)
    day = 86_400_000_000
    trades = pd.DataFrame({"symbol": sym(n // 2), "event_ts": BASE + rng.integers(0, day, n // 2).astype("timedelta64[us]"),
                           "trade_pr": rng.normal(100, 1, n // 2)})
    quotes = pd.DataFrame({"symbol": sym(n // 2), "event_ts": BASE + rng.integers(0, day, n // 2).astype("timedelta64[us]"),
                           "bid": rng.normal(99, 1, n // 2), "ask": rng.normal(101, 1, n // 2)})
    gaps = rng.exponential(0.01, n)
    ticks = pd.DataFrame({"symbol": rng.integers(0, 100, n).astype(np.int32),
                          "event_ts": BASE + (np.cumsum(gaps) * 1e6).astype("timedelta64[us]"),
                          "x": 100 + np.cumsum(rng.normal(0, 1, n)) * 0.01})
    return trades, quotes, ticks


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, nargs="+", default=[1_000_000, 10_000_000])
    args = ap.parse_args()
    out = {"pandas": pd.__version__, "threads": 1, "cpu_count": os.cpu_count(), "runs": []}
    for n in args.rows:
        trades, quotes, ticks = tables(n, 7)
        t0 = time.perf_counter()
        left = trades.sort_values("event_ts", kind="stable")
        right = quotes.sort_values("event_ts", kind="stable")
        pd.merge_asof(left, right, on="event_ts", by="symbol")
        t1 = time.perf_counter()
        r = ticks.set_index("event_ts").groupby("symbol")["x"].rolling("600s")
        r.mean(); r.std(); r.min(); r.max(); r.sum(); r.count()
        t2 = time.perf_counter()
        out["runs"].append({"rows_per_operator": n, "merge_asof_s": round(t1 - t0, 3), "rolling_600s_six_stats_s": round(t2 - t1, 3),
                            "rows_per_s": round(2 * n / (t2 - t0))})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
