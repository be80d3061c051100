"""The public TSDF API with a clock on it: BASELINE configs[0] (asofJoin of 10^6 trades against 10^6
quotes, 100 symbols) and withRangeStats(600 s) on 10^7 ticks, host columns in, host table out --
pandas-backed and Arrow-backed -- with the time spent moving columns to the device, inside the device
operator, moving results back and in the remaining host work (key encoding, result assembly) told
apart; beside them single-thread pandas (merge_asof / groupby.rolling) and the C oracle on the same
tables.  Prints one JSON line (kept as profiles/r03_api_cfg1.json).

  python tools/bench_api.py [--join-rows 1000000] [--stats-rows 10000000] [--reps 3]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import pandas as pd
import pyarrow as pa
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tempo_amd import TSDF, arrow_io, columns, ops  # noqa: E402

BASE = np.datetime64("2020-08-01T00:00:00", "us")
PCIE_GBPS = 50.0  # what pinned copies reach on the box (profiles/r01_pcie_inclusive.json: 53 GB/s)


def tables(n_join, n_stats, seed=1234):
    rng = np.random.default_rng(seed)
    syms = np.array(["S%03d" % i for i in range(100)], dtype=object)
    span = int(6.5 * 3600 * 1e6)

    def side(n):
        ts = BASE + (rng.integers(0, span // 1000, n) * 1000).astype("timedelta64[us]")  # ms-rounded: ties occur
        return syms[rng.integers(0, 100, n)], ts

    ks, ts = side(n_join)
    trades = pd.DataFrame({"symbol": ks, "event_ts": ts, "trade_pr": 100 + rng.normal(0, 1, n_join).cumsum() * 0.01})
    ks, ts = side(n_join)
    bid = 100 + rng.normal(0, 1, n_join).cumsum() * 0.01
    ask = bid + 0.05
    bid[rng.random(n_join) < 0.01] = np.nan
    ask[rng.random(n_join) < 0.01] = np.nan
    quotes = pd.DataFrame({"symbol": ks, "event_ts": ts, "bid_pr": bid, "ask_pr": ask})
    x = 100 + rng.normal(0, 1, n_stats).cumsum() * 0.01
    x[rng.random(n_stats) < 0.005] = np.nan
    ticks = pd.DataFrame({"symbol": syms[rng.integers(0, 100, n_stats)],
                          "event_ts": BASE + (rng.exponential(0.01, n_stats).cumsum() * 1e6).astype("timedelta64[us]"),
                          "x": x})
    return trades, quotes, ticks


class Phases:
    """Wall time inside the calls that move or compute, each bracketed by a device synchronisation."""

    def __init__(self):
        self.t = dict(h2d=0.0, device=0.0, d2h=0.0)
        self.bytes = dict(h2d=0, d2h=0)
        self._orig = {}

    def _wrap(self, owner, name, phase, count_bytes=None):
        fn = getattr(owner, name)
        self._orig[(owner, name)] = fn

        def timed(*a, **k):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = fn(*a, **k)
            torch.cuda.synchronize()
            self.t[phase] += time.perf_counter() - t0
            if count_bytes:
                self.bytes[phase] += count_bytes(out, a)
            return out
        setattr(owner, name, timed)

    def __enter__(self):
        tens = lambda o: o.numel() * o.element_size() if isinstance(o, torch.Tensor) else 0
        self._wrap(columns, "to_dev", "h2d", lambda out, a: tens(out))
        self._wrap(ops, "arrow_import", "h2d", lambda out, a: tens(out[0]) + tens(out[1]))
        for name in ("asof_join_table", "range_stats_cols", "range_stats", "sort_order"):
            self._wrap(ops, name, "device")
        self._wrap(torch.Tensor, "cpu", "d2h", lambda out, a: tens(out) if a[0].is_cuda else 0)
        self._wrap(arrow_io, "to_host", "d2h", lambda out, a: sum(o.nbytes for o in out if o is not None))
        return self

    def __exit__(self, *exc):
        for (owner, name), fn in self._orig.items():
            setattr(owner, name, fn)


def run(label, fn, reps):
    fn()  # warm-up (library load, allocator)
    best = None
    for _ in range(reps):
        with Phases() as ph:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = fn()
            torch.cuda.synchronize()
            total = time.perf_counter() - t0
        rec = dict(total_ms=total * 1e3, h2d_ms=ph.t["h2d"] * 1e3, device_ms=ph.t["device"] * 1e3,
                   d2h_ms=ph.t["d2h"] * 1e3, h2d_bytes=ph.bytes["h2d"], d2h_bytes=ph.bytes["d2h"])
        rec["host_other_ms"] = rec["total_ms"] - rec["h2d_ms"] - rec["device_ms"] - rec["d2h_ms"]
        rec["pcie_ms_at_50GBps"] = (rec["h2d_bytes"] + rec["d2h_bytes"]) / PCIE_GBPS / 1e6
        rec["host_phases_over_pcie_time"] = (rec["total_ms"] - rec["device_ms"]) / max(rec["pcie_ms_at_50GBps"], 1e-9)
        del out
        if best is None or rec["total_ms"] < best["total_ms"]:
            best = rec
    return {k: (round(v, 3) if isinstance(v, float) else v) for k, v in best.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--join-rows", type=int, default=1_000_000)
    ap.add_argument("--stats-rows", type=int, default=10_000_000)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()
    trades, quotes, ticks = tables(args.join_rows, args.stats_rows)
    at, aq, ak = (pa.Table.from_pandas(t, preserve_index=False) for t in (trades, quotes, ticks))
    out = dict(join_rows_per_side=args.join_rows, stats_rows=args.stats_rows, symbols=100)

    def join(l, r):
        return TSDF(l, partition_cols=["symbol"]).asofJoin(TSDF(r, partition_cols=["symbol"])).df

    def stats(t):
        return TSDF(t, partition_cols=["symbol"]).withRangeStats(colsToSummarize=["x"], rangeBackWindowSecs=600).df

    def stats_arrow(t):
        return TSDF(t, partition_cols=["symbol"]).withRangeStats(colsToSummarize=["x"], rangeBackWindowSecs=600).table

    def join_arrow(l, r):
        return TSDF(l, partition_cols=["symbol"]).asofJoin(TSDF(r, partition_cols=["symbol"])).table

    out["asofJoin_pandas_in_pandas_out"] = run("join/pandas", lambda: join(trades, quotes), args.reps)
    out["asofJoin_arrow_in_arrow_out"] = run("join/arrow", lambda: join_arrow(at, aq), args.reps)
    out["withRangeStats_pandas_in_pandas_out"] = run("stats/pandas", lambda: stats(ticks), args.reps)
    out["withRangeStats_arrow_in_arrow_out"] = run("stats/arrow", lambda: stats_arrow(ak), args.reps)
    for k in ("asofJoin_pandas_in_pandas_out", "asofJoin_arrow_in_arrow_out"):
        out[k]["rows_per_s"] = round(2 * args.join_rows / (out[k]["total_ms"] * 1e-3))
    for k in ("withRangeStats_pandas_in_pandas_out", "withRangeStats_arrow_in_arrow_out"):
        out[k]["rows_per_s"] = round(args.stats_rows / (out[k]["total_ms"] * 1e-3))

    if not args.no_cpu:
        # single-thread pandas on the same tables
        t0 = time.perf_counter()
        pd.merge_asof(trades.sort_values("event_ts", kind="stable"), quotes.sort_values("event_ts", kind="stable"),
                      on="event_ts", by="symbol")
        t1 = time.perf_counter()
        r = ticks.set_index("event_ts").groupby("symbol")["x"].rolling("600s")
        r.mean(); r.std(); r.min(); r.max(); r.sum(); r.count()
        t2 = time.perf_counter()
        out["pandas_single_thread"] = dict(merge_asof_ms=round((t1 - t0) * 1e3, 1), rolling_600s_six_stats_ms=round((t2 - t1) * 1e3, 1),
                                           note="merge_asof has no per-column NULL skipping; rolling has no zscore")
        # the C oracle (the checker, timed here as the CPU port of the reference plan), one thread
        from oracle import oracle
        key_l, uniq = pd.factorize(pd.concat([trades["symbol"], quotes["symbol"]]))
        kl, kr = key_l[:len(trades)].astype(np.int32), key_l[len(trades):].astype(np.int32)
        tl = trades["event_ts"].to_numpy().astype("datetime64[us]").astype(np.int64)
        tr = quotes["event_ts"].to_numpy().astype("datetime64[us]").astype(np.int64)
        vb = (~quotes["bid_pr"].isna()).to_numpy().astype(np.uint8)
        va = (~quotes["ask_pr"].isna()).to_numpy().astype(np.uint8)
        t0 = time.perf_counter()
        oracle.asof_join(kl, tl, kr, tr, valid_r=[None, vb, va])
        t1 = time.perf_counter()
        kk = pd.factorize(ticks["symbol"])[0].astype(np.int32)
        tt = ticks["event_ts"].to_numpy().astype("datetime64[us]").astype(np.int64)
        xv = (~ticks["x"].isna()).to_numpy().astype(np.uint8)
        oracle.range_stats(kk, tt, np.nan_to_num(ticks["x"].to_numpy()), xv, 600)
        t2 = time.perf_counter()
        out["c_oracle_single_thread"] = dict(asof_join_ms=round((t1 - t0) * 1e3, 1), range_stats_ms=round((t2 - t1) * 1e3, 1))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
