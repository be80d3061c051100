"""Random pandas tables through the public TSDF API on the GPU, against the same API calls answered
by the CPU oracle (the operators of tempo_amd.ops swapped for tests/oracle_ops.py, as the CPU suite
does): column naming, prefixes, NULL handling, partition-column encoding and result assembly on
top of the kernels.  Frames are compared as multisets of rows with floats rounded to 5 decimals;
frames that differ there are compared again cell by cell to 1e-9 (a value on a rounding boundary).

  python tools/fuzz_tsdf.py [--seconds 200] [--seed 1] [--log gpurun_out/fuzz_tsdf.log]
"""
import argparse
import contextlib
import json
import os
import sys
import time
import traceback
import warnings

import numpy as np
import pandas as pd
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_ops  # noqa: E402
import tsdf_golden_cases as cases  # noqa: E402
from tempo_amd import TSDF, columns, ops  # noqa: E402

SWAPPED = ("sort_order", "asof_join", "asof_join_table", "gather", "range_stats", "range_stats_cols", "ema", "lookback", "resample",
           "grouped_stats", "interpolate", "write_layout")


@contextlib.contextmanager
def oracle_behind_the_api():
    saved = {n: getattr(ops, n) for n in SWAPPED}
    dev = columns.device
    try:
        columns.device = lambda: torch.device("cpu")
        for n in SWAPPED:
            setattr(ops, n, getattr(oracle_ops, n))
        yield
    finally:
        columns.device = dev
        for n, f in saved.items():
            setattr(ops, n, f)


def table(rng, n, n_sym, two_keys, with_seq=False):
    df = pd.DataFrame({"symbol": rng.choice(["S%d" % i for i in range(n_sym)], n)})
    if two_keys:
        df["venue"] = rng.integers(0, 3, n).astype(np.int64)
    step = int(rng.choice([1, 1000, 250_000]))
    df["event_ts"] = pd.to_datetime(1_596_240_000_000 + rng.integers(0, int(rng.choice([30, 4000, 400_000])), n) * step, unit="ms")
    pr = np.round(100 + rng.normal(0, 2, n), int(rng.choice([1, 6])))
    pr[rng.random(n) < float(rng.choice([0.0, 0.1, 0.6]))] = np.nan
    df["trade_pr"] = pr
    df["qty"] = rng.integers(1, 500, n).astype(np.int64)
    if with_seq:
        df["seq_nb"] = rng.integers(0, 6, n).astype(np.int64)
    return df


def frames(out):
    df = out.df if hasattr(out, "df") else out
    df = df.copy()
    for c in df.columns:
        if df[c].dtype == object and len(df) and not isinstance(df[c].iloc[0], str):
            df[c] = df[c].map(lambda a: repr(np.round(np.asarray(a, dtype=np.float64), 5).tolist()))
    return df


def one(rng):
    n = int(rng.choice([rng.integers(0, 5), rng.integers(5, 400), rng.integers(400, 20_000)]))
    n_sym = int(rng.choice([1, 3, 30]))
    two = bool(rng.random() < 0.3)
    part = ["symbol", "venue"] if two else ["symbol"]
    op = str(rng.choice(["asof", "asof", "range_stats", "ema", "lookback", "resample", "grouped", "bars"]))
    if op == "asof":
        with_seq = bool(rng.random() < 0.3)
        left, right = table(rng, n, n_sym, two), table(rng, max(1, int(n * rng.random() * 2)), n_sym, two, with_seq)
        right = right.rename(columns={"trade_pr": "bid_pr", "qty": "bid_sz"})
        kw = dict(right_prefix=str(rng.choice(["right", "q"])), skipNulls=bool(rng.random() < 0.7))
        if rng.random() < 0.3:
            kw["left_prefix"] = "l"
        if rng.random() < 0.35:
            kw.update(tsPartitionVal=int(rng.choice([1, 10, 600])), fraction=float(rng.choice([0.1, 0.5])))
        yield "asofJoin n=%d+%d sym=%d two=%s seq=%s %r" % (len(left), len(right), n_sym, two, with_seq, kw)
        call = lambda: TSDF(left, partition_cols=part).asofJoin(
            TSDF(right, partition_cols=part, sequence_col="seq_nb" if with_seq else None), **kw)
    else:
        df = table(rng, max(n, 1), n_sym, two)
        t = lambda: TSDF(df, partition_cols=part)
        if op == "range_stats":
            w = int(rng.choice([1, 30, 1000]))
            cols = [] if rng.random() < 0.5 else ["trade_pr"]
            yield "withRangeStats n=%d sym=%d two=%s w=%d cols=%r" % (len(df), n_sym, two, w, cols)
            call = lambda: t().withRangeStats(colsToSummarize=cols, rangeBackWindowSecs=w)
        elif op == "ema":
            w = int(rng.choice([1, 5, 30]))
            yield "EMA n=%d sym=%d two=%s window=%d" % (len(df), n_sym, two, w)
            call = lambda: t().EMA("trade_pr", window=w, exp_factor=0.3)
        elif op == "lookback":
            L, exact = int(rng.choice([1, 3, 8])), bool(rng.random() < 0.5)
            yield "withLookbackFeatures n=%d sym=%d two=%s L=%d exact=%s" % (len(df), n_sym, two, L, exact)
            call = lambda: t().withLookbackFeatures(["trade_pr", "qty"], L, exactSize=exact)
        elif op == "resample":
            freq = str(rng.choice(["sec", "min", "5 minutes", "hr", "day"]))
            func = str(rng.choice(["floor", "ceil", "min", "max", "mean"]))
            # fill=True lays out every bucket between a key's first and last row: only on spans where
            # that grid stays small (the span is drawn inside table(): up to three years)
            span_s = float((df["event_ts"].max() - df["event_ts"].min()).total_seconds())
            fill = bool(rng.random() < 0.25) and freq in ("min", "5 minutes", "hr", "day") and span_s < 40 * 86_400
            yield "resample n=%d sym=%d two=%s freq=%s func=%s fill=%s" % (len(df), n_sym, two, freq, func, fill)
            call = lambda: t().resample(freq=freq, func=func, fill=fill or None)
        elif op == "grouped":
            freq = str(rng.choice(["1 min", "1 hr", "1 day"]))
            yield "withGroupedStats n=%d sym=%d two=%s freq=%s" % (len(df), n_sym, two, freq)
            call = lambda: t().withGroupedStats(freq=freq)
        else:
            freq = str(rng.choice(["min", "hr"]))
            yield "calc_bars n=%d sym=%d two=%s freq=%s" % (len(df), n_sym, two, freq)
            call = lambda: t().calc_bars(freq=freq, metricCols=["trade_pr", "qty"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        try:
            got = frames(call())
            raised = None
        except (ValueError, TypeError, NotImplementedError) as err:
            got, raised = None, (type(err).__name__, str(err)[:200])
        with oracle_behind_the_api():
            try:
                want = frames(call())
                want_raised = None
            except (ValueError, TypeError, NotImplementedError) as err:
                want, want_raised = None, (type(err).__name__, str(err)[:200])
    assert raised == want_raised, "exceptions differ: %r vs %r" % (raised, want_raised)
    if got is not None:
        assert list(got.columns) == list(want.columns), (list(got.columns), list(want.columns))
        assert [str(d) for d in got.dtypes] == [str(d) for d in want.dtypes], (got.dtypes.tolist(), want.dtypes.tolist())
        try:
            cases.assert_frames_equal(got, want, decimals=5)
        except AssertionError:
            # name the first differing cells instead of printing both frames
            cols = sorted(got.columns)
            key = [c for c in cols if got[c].dtype.kind not in "f"]
            a = got.sort_values(key, kind="stable").reset_index(drop=True)
            b = want.sort_values(key, kind="stable").reset_index(drop=True)
            notes = []
            for c in cols:
                if a[c].dtype.kind == "f":
                    x, y = a[c].to_numpy(np.float64), b[c].to_numpy(np.float64)
                    d = ~((np.isnan(x) & np.isnan(y)) | (np.abs(x - y) <= 1e-9 * np.maximum(1.0, np.abs(y))))
                    if d.any():
                        i = int(np.flatnonzero(d)[0])
                        notes.append("%s: %d cells differ, first row %d: %r vs %r" % (c, int(d.sum()), i, x[i], y[i]))
                elif not a[c].equals(b[c]):
                    notes.append("%s: differs" % c)
            if notes:  # otherwise the frames differ below 1e-9 only: a value on a rounding boundary
                raise AssertionError("; ".join(notes))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=200.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--log", default=os.path.join(ROOT, "gpurun_out", "fuzz_tsdf.log"))
    args = ap.parse_args()
    os.makedirs(os.path.dirname(args.log), exist_ok=True)
    t_end, done, bad = time.time() + args.seconds, 0, []
    with open(args.log, "w") as log:
        while time.time() < t_end:
            rng = np.random.default_rng(args.seed * 1_000_003 + done)
            desc, t0 = "", time.time()
            try:
                for desc in one(rng):
                    log.write("%5d start %s\n" % (done + 1, desc))
                    log.flush()
                status = "ok"
            except Exception as err:  # noqa: BLE001
                status = "BAD"
                text = [t.strip() for t in str(err).strip().splitlines() if t.strip()]
                bad.append(dict(case=done, what=desc, error=("%s: %s" % (type(err).__name__, " / ".join(text[:3])))[:700],
                                trace=traceback.format_exc().splitlines()[-8:]))
            done += 1
            log.write("%5d %-5s %.2fs %s\n" % (done, status, time.time() - t0, desc))
            log.flush()
    print(json.dumps(dict(cases=done, disagreements=len(bad), seed=args.seed, first_bad=bad[:8])))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
