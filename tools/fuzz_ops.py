"""Differential fuzzing of every operator against the CPU oracle.

Each case draws a table (size from 0 rows to a little over 2^20, 1 to 5000 keys, one of the row
orders the ordering step has shortcuts for, timestamps on a coarse or fine grid so that ties are
common or rare, NULL rates from none to almost all, NaN / infinite values now and then) and the
operator's parameters, runs the HIP path through tempo_amd.ops and the oracle on the same host
arrays, and compares: indices, rows, buckets, counts, extremes, EMA, lookback features and
interpolated values bit for bit; means, sums and standard deviations to 1e-9.

  python tools/fuzz_ops.py [--seconds 300] [--seed 1] [--log gpurun_out/fuzz_ops.log]

Prints one line per case to the log (so a long run shows progress), a summary on stdout, and exits
with status 1 if any case disagreed.  The oracle is the checker here; nothing in the library uses it.
"""
import argparse
import json
import os
import sys
import time
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_ops as orc  # noqa: E402  (the oracle behind the same call signatures as tempo_amd.ops)
from tempo_amd import ops  # noqa: E402

BASE = 1_596_240_000_000_000
RTOL = 1e-9
LAST = {"desc": ""}


def note(desc):
    """Remember what the case is before anything is compared, so that a failure can name it."""
    LAST["desc"] = desc
    return desc


def dev(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a))


def draw_size(rng):
    kind = rng.integers(0, 10)
    if kind == 0:
        return int(rng.integers(0, 4))
    if kind <= 5:
        return int(rng.integers(4, 40_000))
    if kind <= 7:
        return int(rng.integers(40_000, 400_000))
    return int(rng.integers(1_040_000, 1_300_000))   # past the 2^20-row thresholds of the shortcuts


def draw_table(rng, n, n_keys, pre_epoch=False):
    """key, ts in one of the row orders: random, (key, ts)-sorted, time-ordered, a few time-ordered
    runs, every key in time order with the keys interleaved."""
    shape = str(rng.choice(["random", "sorted", "time", "runs", "per_key"]))
    step_us = int(rng.choice([1, 1000, 1_000_000, 250_000]))
    span = int(rng.choice([30, 5_000, 3_000_000]))
    key = rng.integers(0, n_keys, n).astype(np.int32)
    if n_keys > 1 and rng.random() < 0.3:  # a skewed key column
        key = np.minimum(key, rng.integers(0, n_keys, n)).astype(np.int32)
    ts = (BASE if not pre_epoch else -BASE // 7) + rng.integers(0, span, n) * step_us
    if n > 1:
        if shape == "sorted":
            o = np.lexsort((ts, key))
            key, ts = key[o], ts[o]
        elif shape == "time":
            ts = np.sort(ts)
        elif shape == "runs":
            k = int(rng.integers(2, min(20, n) + 1))
            cuts = np.r_[0, np.sort(rng.choice(np.arange(1, n), k - 1, replace=False)), n]
            for a, b in zip(cuts[:-1], cuts[1:]):
                ts[a:b] = np.sort(ts[a:b])
        elif shape == "per_key":
            where = np.argsort(key, kind="stable")
            ts_new = np.empty_like(ts)
            ts_new[where] = ts[np.lexsort((ts, key))]
            ts = ts_new
    return key, ts, shape


def draw_mask(rng, n):
    p = float(rng.choice([0.0, 0.01, 0.3, 0.95]))
    if rng.random() < 0.25:
        return None
    m = (rng.random(n) >= p).astype(np.uint8)
    if n > 50 and rng.random() < 0.3:  # a long run of NULLs
        a = int(rng.integers(0, n - 10))
        m[a:a + int(rng.integers(10, max(11, n // 3)))] = 0
    return m


def draw_metric(rng, n, allow_nonfinite=True):
    kind = rng.integers(0, 5)
    if kind == 0:
        x = np.round(rng.normal(0, 3, n), 1)              # coarse: many ties
    elif kind == 1:
        x = 1e6 + np.cumsum(rng.normal(0, 1e-3, n))       # tiny spread on a large level
    elif kind == 2:
        x = np.full(n, float(rng.normal()))               # constant
    else:
        x = 100 + np.cumsum(rng.normal(0, 1, n)) * 0.01
    if allow_nonfinite and n and rng.random() < 0.2:
        at = rng.integers(0, n, max(1, n // 500))
        x[at] = rng.choice([np.nan, np.inf, -np.inf], at.size)
    return x


def draw_seq(rng, n):
    if rng.random() < 0.6:
        return None, None
    return rng.integers(-5, 50, n), (None if rng.random() < 0.3 else (rng.random(n) > 0.2).astype(np.uint8))


def eq(a, b, what, zero_sign=True):
    a = a.cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    if a.shape != b.shape:
        raise AssertionError("%s: shape %r vs %r" % (what, a.shape, b.shape))
    if a.dtype.kind == "f":
        same = (a.view(np.int64) == b.view(np.int64)) | ((a == b) & (a != 0)) | (np.isnan(a) & np.isnan(b))
        same |= (a == 0) & (b == 0) & ((np.signbit(a) == np.signbit(b)) | (not zero_sign))
    else:
        same = a.astype(np.int64) == b.astype(np.int64)
    if not same.all():
        at = int(np.flatnonzero(~same)[0])
        raise AssertionError("%s: %d of %d differ, first at %d: %r vs %r" % (what, int((~same).sum()), a.size, at,
                                                                          a.reshape(-1)[at], b.reshape(-1)[at]))


def close(a, b, what, atol=0.0, rtol=RTOL):
    a, b = a.cpu().numpy(), (b.cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b))
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, equal_nan=True, err_msg=what)


def frame_rows(key, sec, w):
    """Rows all the RANGE frames of w seconds hold together (what a row-by-row evaluation visits)."""
    o = np.lexsort((sec, key))
    comp = key[o].astype(np.int64) * (1 << 40) + (sec[o] - sec.min())
    start = np.searchsorted(comp, comp - w)
    return float((np.arange(len(comp)) - start + 1).sum())


# ---- one function per operator: draws a case, returns a short description; raises on mismatch

def case_asof(rng):
    n_keys = int(rng.choice([1, 3, 40, 5000]))
    nl, nr = draw_size(rng), draw_size(rng)
    kl, tl, sl = draw_table(rng, nl, n_keys)
    kr, tr, sr = draw_table(rng, nr, n_keys)
    seq, seqv = draw_seq(rng, nr)
    masks = [m for m in (draw_mask(rng, nr) for _ in range(int(rng.integers(0, 4)))) if m is not None]
    skew = rng.random() < 0.4
    tpv = (int(rng.choice([1, 3, 10, 600])) if rng.random() < 0.6 else float(rng.choice([0.5, 7.5]))) if skew else None
    frac = float(rng.choice([0.1, 0.5, 0.9]))
    secs = bool(rng.random() < 0.15)
    if secs:
        tl, tr = tl // 1_000_000, tr // 1_000_000
    desc = LAST["desc"] = "asof nl=%d(%s) nr=%d(%s) keys=%d masks=%d seq=%s skew=%r frac=%g secs=%s" % (
        nl, sl, nr, sr, n_keys, len(masks), seq is not None, tpv, frac, secs)
    use_key = n_keys > 1 or rng.random() < 0.5
    a = (dev(kl) if use_key else None, dev(tl), dev(kr) if use_key else None, dev(tr), n_keys)
    h = (host(kl) if use_key else None, host(tl), host(kr) if use_key else None, host(tr), n_keys)
    if not use_key:
        kl[:], kr[:] = 0, 0
    kw = dict(ts_is_seconds=secs, ts_partition_val=tpv, fraction=frac)
    if rng.random() < 0.5:
        row, midx, miss = ops.asof_join(*a, seq_r=dev(seq), seq_r_valid=dev(seqv), masks=[dev(m) for m in masks], **kw)
        erow, emidx, emiss = orc.asof_join(*h, seq_r=host(seq), seq_r_valid=host(seqv), masks=[host(m) for m in masks], **kw)
        eq(row, erow, "row index")
        eq(midx.reshape(len(masks), nl), emidx, "skip-null indices")
        eq(miss, emiss, "missing flags")
        return desc + " [records]"
    ncol = int(rng.integers(1, 5))
    rcols = [rng.normal(0, 1, nr) if rng.random() < 0.7 else rng.integers(-9, 9, nr) for _ in range(ncol)]
    col_mask = [None if (not masks or rng.random() < 0.3) else int(rng.integers(0, len(masks))) for _ in range(ncol)]
    lcols = [rng.normal(0, 1, nl) for _ in range(int(rng.integers(0, 3)))]
    skip = bool(rng.random() < 0.7)
    dmasks, hmasks = [dev(m) for m in masks], [host(m) for m in masks]
    j = ops.asof_join_table(*a, [dev(c) for c in lcols], [dev(c) for c in rcols],
                            [None if s is None else dmasks[s] for s in col_mask], skip_nulls=skip,
                            seq_r=dev(seq), seq_r_valid=dev(seqv), **kw)
    e = orc.asof_join_table(*h, [host(c) for c in lcols], [host(c) for c in rcols],
                            [None if s is None else hmasks[s] for s in col_mask], skip_nulls=skip,
                            seq_r=host(seq), seq_r_valid=host(seqv), **kw)
    eq(j["perm"], e["perm"], "perm")
    eq(j["key"] if use_key else torch.zeros_like(j["perm"]), e["key"], "key")
    eq(j["ts"], e["ts"], "ts")
    for c in range(len(lcols)):
        eq(j["left"][c], e["left"][c], "left column %d" % c)
    for c in range(ncol):
        ok = e["right"][c][1].numpy() != 0
        eq(j["right"][c][1], e["right"][c][1], "right validity %d" % c)
        eq(j["right"][c][0].cpu()[torch.from_numpy(ok)], e["right"][c][0][torch.from_numpy(ok)], "right column %d" % c)
    return desc + " [table cols=%d skip=%s]" % (ncol, skip)


def case_asof_ordered(rng):
    """Both sides in (key, ts[, seq NULLS FIRST]) order: the one-sweep join. Side sizes far apart,
    masks with long NULL runs, few timestamps per key (many ties), skew brackets, records and
    fused-table output, the left table by reference."""
    n_keys = int(rng.choice([1, 2, 40, 5000, 300_000]))
    nl, nr = draw_size(rng), draw_size(rng)
    lop = rng.random()
    if lop < 0.2:
        nl = int(rng.integers(1, 40))         # a handful of left rows over a long right side
    elif lop < 0.4:
        nr = int(rng.integers(1, 40))
    step_us = int(rng.choice([1, 1000, 1_000_000, 250_000]))
    span = int(rng.choice([3, 30, 5_000, 3_000_000]))
    kl = rng.integers(0, n_keys, nl).astype(np.int32)
    kr = rng.integers(0, n_keys, nr).astype(np.int32)
    split = rng.random()
    if n_keys > 1 and split < 0.25:           # the sides share few keys
        kl, kr = np.minimum(kl, n_keys // 2).astype(np.int32), np.maximum(kr, n_keys // 2 - 1).astype(np.int32)
    elif n_keys > 1 and split < 0.5:          # one heavy key
        kl = np.where(rng.random(nl) < 0.8, n_keys // 3, kl).astype(np.int32)
        kr = np.where(rng.random(nr) < 0.8, n_keys // 3, kr).astype(np.int32)
    shift = int(rng.choice([0, 0, span // 2 + 1, -(span // 2 + 1)]))  # right side partly before / after the left
    tl = BASE + rng.integers(0, span, nl) * step_us
    tr = BASE + (rng.integers(0, span, nr) + shift) * step_us
    seq, seqv = draw_seq(rng, nr)
    if seq is not None and seqv is None:
        seqv = np.ones(nr, np.uint8)
    masks = [m for m in (draw_mask(rng, nr) for _ in range(int(rng.integers(0, 5)))) if m is not None][:4]
    o = np.lexsort((tl, kl))
    kl, tl = kl[o], tl[o]
    o = np.lexsort((tr, kr)) if seq is None else np.lexsort((np.where(seqv == 1, seq, -10**9), tr, kr))
    kr, tr = kr[o], tr[o]
    if seq is not None:
        seq, seqv = seq[o], seqv[o]
    skew = rng.random() < 0.35
    tpv = (int(rng.choice([1, 3, 10, 600])) if rng.random() < 0.6 else float(rng.choice([0.5, 7.5]))) if skew else None
    frac = float(rng.choice([0.1, 0.5, 0.9]))
    secs = bool(rng.random() < 0.15)
    if secs:
        tl, tr = tl // 1_000_000, tr // 1_000_000
    desc = LAST["desc"] = "asof_ordered nl=%d nr=%d keys=%d span=%d step=%d shift=%d masks=%d seq=%s skew=%r frac=%g secs=%s" % (
        nl, nr, n_keys, span, step_us, shift, len(masks), seq is not None, tpv, frac, secs)
    if nl == 0 or nr == 0:
        return desc + " [skipped: empty]"
    use_key = n_keys > 1 or rng.random() < 0.5
    a = (dev(kl) if use_key else None, dev(tl), dev(kr) if use_key else None, dev(tr), n_keys)
    h = (host(kl) if use_key else None, host(tl), host(kr) if use_key else None, host(tr), n_keys)
    kw = dict(ts_is_seconds=secs, ts_partition_val=tpv, fraction=frac)
    if rng.random() < 0.4:
        row, midx, miss = ops.asof_join(*a, seq_r=dev(seq), seq_r_valid=dev(seqv), masks=[dev(m) for m in masks], **kw)
        erow, emidx, emiss = orc.asof_join(*h, seq_r=host(seq), seq_r_valid=host(seqv), masks=[host(m) for m in masks], **kw)
        eq(row, erow, "row index")
        eq(midx.reshape(len(masks), nl), emidx, "skip-null indices")
        eq(miss, emiss, "missing flags")
        return desc + " [records]"
    ncol = int(rng.integers(1, 7))
    rcols = [rng.normal(0, 1, nr) if rng.random() < 0.7 else rng.integers(-9, 9, nr) for _ in range(ncol)]
    col_mask = [None if (not masks or rng.random() < 0.3) else int(rng.integers(0, len(masks))) for _ in range(ncol)]
    lcols = [rng.normal(0, 1, nl) for _ in range(int(rng.integers(0, 3)))]
    skip = bool(rng.random() < 0.7)
    by_ref = bool(rng.random() < 0.5)
    dmasks, hmasks = [dev(m) for m in masks], [host(m) for m in masks]
    dl = [dev(c) for c in lcols]
    j = ops.asof_join_table(*a, dl, [dev(c) for c in rcols],
                            [None if s is None else dmasks[s] for s in col_mask], skip_nulls=skip,
                            seq_r=dev(seq), seq_r_valid=dev(seqv), left_by_reference=by_ref, **kw)
    e = orc.asof_join_table(*h, [host(c) for c in lcols], [host(c) for c in rcols],
                            [None if s is None else hmasks[s] for s in col_mask], skip_nulls=skip,
                            seq_r=host(seq), seq_r_valid=host(seqv), **kw)
    if by_ref and not j["in_place"] and tpv is None:
        raise AssertionError("an ordered left table was not used by reference")
    if j["in_place"]:
        eq(e["perm"], np.arange(nl), "expected order is the input order")
        if lcols and j["left"][0].data_ptr() != dl[0].data_ptr():
            raise AssertionError("left columns were copied")
    else:
        eq(j["perm"], e["perm"], "perm")
    if use_key:
        eq(j["key"], e["key"], "key")
    eq(j["ts"], e["ts"], "ts")
    for c in range(len(lcols)):
        eq(j["left"][c], e["left"][c], "left column %d" % c)
    for c in range(ncol):
        ok = e["right"][c][1].numpy() != 0
        eq(j["right"][c][1], e["right"][c][1], "right validity %d" % c)
        eq(j["right"][c][0].cpu()[torch.from_numpy(ok)], e["right"][c][0][torch.from_numpy(ok)], "right column %d" % c)
    return desc + " [table cols=%d skip=%s by_ref=%s in_place=%s]" % (ncol, skip, by_ref, j["in_place"])


def case_range_stats(rng):
    n_keys = int(rng.choice([1, 3, 40, 1000]))
    n = draw_size(rng)
    key, ts, shape = draw_table(rng, n, n_keys)
    w = int(rng.choice([0, 1, 10, 600, 5000]))
    # the oracle walks every frame row by row: keep the rows it visits below ~3e8 (shorter
    # frames first, then fewer rows)
    while n > 4 and os.environ.get("FUZZ_NO_BOUND") != "1" and frame_rows(key, ts // 1_000_000, w) > 3e8:
        if w > 0:
            w //= 10
        else:
            n //= 2
            key, ts = key[:n], ts[:n]
    is_int = rng.random() < 0.25
    x = rng.integers(-10**9, 10**9, n) if is_int else draw_metric(rng, n)
    xv = draw_mask(rng, n)
    desc = LAST["desc"] = "range_stats n=%d(%s) keys=%d w=%d int=%s" % (n, shape, n_keys, w, is_int)
    if n == 0:
        return desc + " [skipped: empty]"
    r = ops.range_stats(dev(key), dev(ts), dev(x), dev(xv), n_keys, w)
    e = orc.range_stats(host(key), host(ts), host(x), host(xv), n_keys, w)
    eq(r["perm"], e["perm"], "perm")
    eq(r["valid"], e["valid"], "valid flags")
    v = e["valid"].numpy()
    ok, ok2, okz = torch.from_numpy((v & 1) != 0), torch.from_numpy((v & 2) != 0), torch.from_numpy((v & 4) != 0)
    eq(r["count"], e["count"], "count")
    # a frame holding both -0.0 and 0.0: Spark's ordering calls them equal and keeps whichever came
    # first, so the sign of a zero extreme is not pinned
    eq(r["min"].cpu()[ok], e["min"][ok], "min", zero_sign=False)
    eq(r["max"].cpu()[ok], e["max"][ok], "max", zero_sign=False)
    scale = float(np.nanmax(np.abs(np.where(np.isfinite(x), x, 0)))) if n else 1.0
    close(r["mean"].cpu()[ok], e["mean"][ok], "mean", atol=1e-12 * scale)
    if is_int:
        eq(r["sum"].cpu()[ok], e["sum"][ok], "sum")
    else:
        close(r["sum"].cpu()[ok], e["sum"][ok], "sum", atol=1e-9 * scale)
    close(r["stddev"].cpu()[ok2], e["stddev"][ok2], "stddev", atol=1e-12 * max(scale, 1.0))
    return desc


def case_ema(rng):
    n_keys = int(rng.choice([1, 5, 300]))
    n = draw_size(rng)
    key, ts, shape = draw_table(rng, n, n_keys)
    seq, seqv = draw_seq(rng, n)
    x, xv = draw_metric(rng, n, allow_nonfinite=False), draw_mask(rng, n)
    window = int(rng.choice([1, 2, 30, 200, 1024]))
    alpha = float(rng.choice([0.2, 0.05, 0.9]))
    desc = LAST["desc"] = "ema n=%d(%s) keys=%d window=%d seq=%s" % (n, shape, n_keys, window, seq is not None)
    if n == 0:
        return desc + " [skipped: empty]"
    perm, out = ops.ema(dev(key), dev(ts), dev(x), dev(xv), n_keys, window, alpha, seq=dev(seq), seq_valid=dev(seqv))
    eperm, eout = orc.ema(host(key), host(ts), host(x), host(xv), n_keys, window, alpha, seq=host(seq), seq_valid=host(seqv))
    eq(perm, eperm, "perm")
    eq(out, eout, "ema")
    return desc


def case_lookback(rng):
    n_keys = int(rng.choice([1, 5, 300]))
    n = min(draw_size(rng), 300_000)
    key, ts, shape = draw_table(rng, n, n_keys)
    seq, seqv = draw_seq(rng, n)
    f = int(rng.integers(1, 5))
    xs = [draw_metric(rng, n, allow_nonfinite=False) for _ in range(f)]
    xv = [draw_mask(rng, n) for _ in range(f)]
    L = int(rng.choice([1, 3, 10, 40]))
    exact = bool(rng.random() < 0.5)
    desc = LAST["desc"] = "lookback n=%d(%s) keys=%d L=%d F=%d exact=%s seq=%s" % (n, shape, n_keys, L, f, exact, seq is not None)
    if n == 0:
        return desc + " [skipped: empty]"
    rows, lens, feat, fv = ops.lookback(dev(key), dev(ts), [dev(c) for c in xs], [dev(v) for v in xv], n_keys, L,
                                        exact_size=exact, seq=dev(seq), seq_valid=dev(seqv))
    erows, elens, efeat, efv = orc.lookback(host(key), host(ts), [host(c) for c in xs], [host(v) for v in xv], n_keys, L,
                                            exact_size=exact, seq=host(seq), seq_valid=host(seqv))
    eq(rows, erows, "rows")
    if not exact:
        eq(lens, elens, "history lengths")
    if fv is None:  # no NULLs among the inputs and only full histories kept: every feature is valid
        assert bool((efv != 0).all()), "oracle marks NULL features where the library returned no validity array"
    else:
        eq(fv.reshape(-1), efv.reshape(-1), "feature validity")
    eq(feat.reshape(-1), efeat.reshape(-1), "features")
    return desc


def draw_cols(rng, n):
    k = int(rng.integers(0, 4))
    cols = [(rng.integers(-10**6, 10**6, n) if rng.random() < 0.3 else draw_metric(rng, n, allow_nonfinite=False))
            for _ in range(k)]
    return cols, [draw_mask(rng, n) for _ in range(k)]


def case_resample(rng):
    n_keys = int(rng.choice([1, 9, 300, 3000]))
    n = draw_size(rng)
    key, ts, shape = draw_table(rng, n, n_keys, pre_epoch=bool(rng.random() < 0.2))
    dur = int(rng.choice([1_000_000, 60_000_000, 3_600_000_000, 86_400_000_000]))
    func = str(rng.choice(["mean", "min", "max", "floor", "ceil"]))
    cols, valid = draw_cols(rng, n)
    desc = LAST["desc"] = "resample n=%d(%s) keys=%d dur=%d func=%s cols=%d" % (n, shape, n_keys, dur, func, len(cols))
    r = ops.resample(dev(key), dev(ts), n_keys, dur, func, [dev(c) for c in cols], [dev(v) for v in valid])
    e = orc.resample(host(key), host(ts), n_keys, dur, func, [host(c) for c in cols], [host(v) for v in valid])
    assert r["n"] == e["n"], "bucket count %d vs %d" % (r["n"], e["n"])
    eq(r["key"], e["key"], "key")
    eq(r["start"], e["start"], "bucket start")
    if func in ("floor", "ceil"):
        eq(r["row"], e["row"], "picked row")
        return desc
    if len(cols) == 0:
        return desc
    eq(r["valid"], e["valid"], "validity")
    ok = torch.from_numpy(e["valid"].numpy() != 0)
    if func == "mean":
        g, w = r["f64"].cpu(), e["f64"]
        close(g[ok], w[ok], "mean", atol=1e-12)
    else:
        eq(r["f64"].cpu()[ok], e["f64"][ok], "f64 extremes", zero_sign=False)
        eq(r["i64"].cpu()[ok], e["i64"][ok], "i64 extremes")
    return desc


def case_grouped(rng):
    n_keys = int(rng.choice([1, 7, 300, 3000]))
    n = draw_size(rng)
    key, ts, shape = draw_table(rng, n, n_keys)
    dur = int(rng.choice([1_000_000, 60_000_000, 86_400_000_000]))
    cols, valid = draw_cols(rng, n)
    desc = LAST["desc"] = "grouped_stats n=%d(%s) keys=%d dur=%d cols=%d" % (n, shape, n_keys, dur, len(cols))
    r = ops.grouped_stats(dev(key), dev(ts), n_keys, dur, [dev(c) for c in cols], [dev(v) for v in valid])
    e = orc.grouped_stats(host(key), host(ts), n_keys, dur, [host(c) for c in cols], [host(v) for v in valid])
    assert r["n"] == e["n"], "bucket count %d vs %d" % (r["n"], e["n"])
    eq(r["key"], e["key"], "key")
    eq(r["start"], e["start"], "bucket start")
    for c in range(len(cols)):
        eq(r["count"][c], e["count"][c], "count %d" % c)
        cnt = e["count"][c].numpy()
        ok, ok2 = torch.from_numpy(cnt > 0), torch.from_numpy(cnt > 1)
        eq(r["min"][c].cpu()[ok], e["min"][c][ok], "min %d" % c, zero_sign=False)
        eq(r["max"][c].cpu()[ok], e["max"][c][ok], "max %d" % c, zero_sign=False)
        scale = float(np.max(np.abs(cols[c]))) if n else 1.0
        if e["is_i64"][c]:
            eq(r["sum"][c].cpu()[ok], e["sum"][c][ok], "sum %d" % c)
        else:
            close(r["sum"][c].cpu()[ok], e["sum"][c][ok], "sum %d" % c, atol=1e-9 * scale)
        close(r["mean"][c].cpu()[ok], e["mean"][c][ok], "mean %d" % c, atol=1e-12 * scale)
        close(r["stddev"][c].cpu()[ok2], e["stddev"][c][ok2], "stddev %d" % c, atol=1e-12 * max(scale, 1.0))
    return desc


def case_interpolate(rng):
    n_keys = int(rng.choice([1, 11, 200]))
    freq = int(rng.choice([1_000_000, 30_000_000, 3_600_000_000]))
    buckets = int(rng.choice([1, 5, 700, 5000]))
    p_keep = float(rng.choice([0.05, 0.4, 0.9, 1.0]))
    keys, stamps = [], []
    for k in range(n_keys):
        keep = np.flatnonzero(rng.random(buckets) < p_keep)
        keys.append(np.full(keep.size, k, np.int32))
        stamps.append(1_577_836_800_000_000 + keep.astype(np.int64) * freq)
    key, ts = np.concatenate(keys), np.concatenate(stamps)
    o = rng.permutation(key.size)
    key, ts = key[o], ts[o]
    n = key.size
    k = int(rng.integers(1, 4))
    cols = [rng.normal(100, 5, n) for _ in range(k)]
    valid = [draw_mask(rng, n) for _ in range(k)]
    method = str(rng.choice(["zero", "null", "bfill", "ffill", "linear"]))
    desc = LAST["desc"] = "interpolate n=%d keys=%d freq=%d method=%s cols=%d" % (n, n_keys, freq, method, k)
    r = ops.interpolate(dev(key), dev(ts), n_keys, freq, method, [dev(c) for c in cols], [dev(v) for v in valid])
    e = orc.interpolate(host(key), host(ts), n_keys, freq, method, [host(c) for c in cols], [host(v) for v in valid])
    assert r["n"] == e["n"], "row count %d vs %d" % (r["n"], e["n"])
    eq(r["src"], e["src"], "source row")
    eq(r["key"], e["key"], "key")
    eq(r["ts"], e["ts"], "ts")
    eq(r["ts_interp"], e["ts_interp"], "ts flag")
    for c in range(k):
        eq(r["valid"][c], e["valid"][c], "validity %d" % c)
        eq(r["interp"][c], e["interp"][c], "interp flag %d" % c)
        ok = torch.from_numpy(e["valid"][c].numpy() == 1)
        eq(r["val"][c].cpu()[ok], e["val"][c][ok], "value %d" % c)
    return desc


def case_asof_wide(rng):
    """Tables whose key, time and sequence fields or validity bits do not fit one 64-bit sort word:
    many keys, a year of microseconds, a sequence column, up to 8 nullable right columns."""
    n_keys = int(rng.choice([300_000, 1_000_000]))
    nl, nr = min(draw_size(rng), 300_000), min(draw_size(rng), 300_000)
    year = 365 * 86_400 * 1_000_000
    kl, kr = rng.integers(0, n_keys, nl).astype(np.int32), rng.integers(0, n_keys, nr).astype(np.int32)
    if rng.random() < 0.5:  # few distinct keys out of a large dictionary: ties and real matches
        kl, kr = (kl % 50) * (n_keys // 50), (kr % 50) * (n_keys // 50)
    tl, tr = BASE + rng.integers(0, year, nl), BASE + rng.integers(0, year, nr)
    seq, seqv = (rng.integers(0, 1 << 20, nr), (rng.random(nr) > 0.2).astype(np.uint8)) if rng.random() < 0.6 else (None, None)
    masks = [(rng.random(nr) >= 0.3).astype(np.uint8) for _ in range(int(rng.integers(0, 9)))]
    desc = LAST["desc"] = "asof_wide nl=%d nr=%d keys=%d masks=%d seq=%s" % (nl, nr, n_keys, len(masks), seq is not None)
    rcols = [rng.normal(0, 1, nr) for _ in range(max(1, len(masks)))]
    col_mask = [c if c < len(masks) else None for c in range(len(rcols))]
    dmasks, hmasks = [dev(m) for m in masks], [host(m) for m in masks]
    j = ops.asof_join_table(dev(kl), dev(tl), dev(kr), dev(tr), n_keys, [], [dev(c) for c in rcols],
                            [None if c is None else dmasks[c] for c in col_mask], seq_r=dev(seq), seq_r_valid=dev(seqv))
    e = orc.asof_join_table(host(kl), host(tl), host(kr), host(tr), n_keys, [], [host(c) for c in rcols],
                            [None if c is None else hmasks[c] for c in col_mask], seq_r=host(seq), seq_r_valid=host(seqv))
    eq(j["perm"], e["perm"], "perm")
    eq(j["key"], e["key"], "key")
    eq(j["ts"], e["ts"], "ts")
    for c in range(len(rcols)):
        ok = torch.from_numpy(e["right"][c][1].numpy() != 0)
        eq(j["right"][c][1], e["right"][c][1], "right validity %d" % c)
        eq(j["right"][c][0].cpu()[ok], e["right"][c][0][ok], "right column %d" % c)
    return desc


def case_partition(rng):
    """The shuffle's routing: rows bucketed by owner (hash or owner table), every column moved."""
    import host_partition as hp
    n = max(1, draw_size(rng))
    n_keys = int(rng.choice([1, 3, 40, 5000]))
    n_parts = int(rng.choice([1, 2, 3, 8, 64]))
    key = rng.integers(0, n_keys, n).astype(np.int32)
    table = rng.integers(0, n_parts, n_keys).astype(np.int32) if rng.random() < 0.5 else None
    cols = [rng.integers(0, 1 << 60, n), rng.normal(0, 1, n), (rng.random(n) > 0.5).astype(np.uint8),
            rng.integers(0, 1 << 30, n).astype(np.int32)][:int(rng.integers(1, 5))]
    desc = LAST["desc"] = "partition n=%d keys=%d parts=%d table=%s cols=%d" % (n, n_keys, n_parts, table is not None, len(cols))
    owner = hp.owners_of(key, n_parts, table)
    perm = np.argsort(owner, kind="stable")
    counts = np.bincount(owner, minlength=n_parts)
    outs, got_counts = ops.partition_move(dev(key), [dev(c) for c in cols], n_parts, owner_table=dev(table))
    eq(got_counts, counts, "counts")
    for c in range(len(cols)):
        eq(outs[c], cols[c][perm], "column %d" % c)
    p2, c2 = ops.hash_partition(dev(key), n_parts, owner_table=dev(table))
    eq(c2, counts, "counts (perm form)")
    eq(p2, perm, "perm")
    return desc


def case_write_layout(rng):
    n_keys = int(rng.choice([1, 9, 300]))
    n = max(1, min(draw_size(rng), 400_000))
    key, ts, shape = draw_table(rng, n, n_keys, pre_epoch=bool(rng.random() < 0.2))
    ts = ts + rng.integers(0, 40, n) * 86_400_000_000   # several days
    desc = LAST["desc"] = "write_layout n=%d(%s) keys=%d" % (n, shape, n_keys)
    got = ops.write_layout(dev(key), dev(ts), n_keys)
    want = orc.write_layout(host(key), host(ts), n_keys)
    for name, g, w in zip(("event_dt", "event_time", "perm", "dt_range"), got, want):
        eq(g, w, name)
    return desc


def case_wire(rng):
    """The wire form of a key shuffle: narrowed by the encode kernel to exactly the bytes a numpy
    restatement gives, widened by the decode kernel back to the table."""
    from tempo_amd.dist import WirePlan
    n = max(1, draw_size(rng))
    key_bits = int(rng.choice([1, 3, 7, 8, 14, 20, 31]))
    n_masks = int(rng.integers(0, min(8, 32 - key_bits) + 1)) if key_bits < 31 else 0
    widths = [int(w) for w in rng.choice([4, 5, 6, 8], int(rng.integers(0, 4)))]
    off = int(rng.integers(0, 4))           # views that start off the 16-byte grid take the row-by-row kernel
    m = n + off
    key = rng.integers(0, 1 << key_bits, m).astype(np.int32)
    masks = [(rng.random(m) > 0.3).astype(np.uint8) for _ in range(n_masks)]
    ints, mins = [], []
    for w in widths:
        lo = int(rng.integers(-(1 << 50), 1 << 50))
        span = (1 << 62) if w == 8 else (1 << (8 * w))
        c = lo + rng.integers(0, span, m)
        c[off], c[m - 1] = lo, lo + span - 1
        ints.append(c)
        mins.append(lo)
    other = rng.normal(0, 1, m)
    cols = ints + masks + [other]
    desc = LAST["desc"] = "wire n=%d off=%d key_bits=%d masks=%d ints=%r" % (n, off, key_bits, n_masks, widths)
    plan = WirePlan(key_bits, list(range(len(ints), len(ints) + n_masks)), {i: (mins[i], widths[i]) for i in range(len(ints))})
    d_key, d_cols = dev(key)[off:], [dev(c)[off:] for c in cols]
    wire = plan.encode(d_key, d_cols)
    word = key[off:].astype(np.int64)
    for b, f in enumerate(masks):
        word |= f[off:].astype(np.int64) << (key_bits + b)
    width = key_bits + n_masks
    want = [word.astype(np.uint8 if width <= 8 else (np.uint16 if width <= 16 else np.uint32))]
    for c, lo, w in zip(ints, mins, widths):
        if w == 8:
            want.append(c[off:])
            continue
        rel = (c[off:] - lo).astype(np.uint64)
        want.append((rel & 0xffffffff).astype(np.uint32))
        if w > 4:
            want.append((rel >> 32).astype(np.uint8 if w == 5 else np.uint16))
    want.append(other[off:])
    assert len(wire) == len(want), "%d wire columns, %d expected" % (len(wire), len(want))
    for i, (g, w) in enumerate(zip(wire, want)):
        g = g.cpu().numpy()
        assert g.dtype.itemsize == w.dtype.itemsize, "wire column %d: %d-byte elements, %d expected" % (i, g.dtype.itemsize, w.dtype.itemsize)
        eq(g.view(w.dtype), w, "wire column %d" % i)
    back_key, back = plan.decode(wire, [c.dtype for c in d_cols])
    eq(back_key, key[off:], "key after decode")
    for i, c in enumerate(cols):
        eq(back[i], c[off:], "column %d after decode" % i)
    return desc


CASES = [("asof", case_asof, 4), ("range_stats", case_range_stats, 3), ("ema", case_ema, 1), ("lookback", case_lookback, 1),
         ("resample", case_resample, 2), ("grouped_stats", case_grouped, 2), ("interpolate", case_interpolate, 1)]
# a pool of its own (--extra), so that the case a seed of the first pool names stays the same
EXTRA_CASES = [("asof_wide", case_asof_wide, 1), ("partition", case_partition, 1), ("write_layout", case_write_layout, 1)]
# a third pool (--ordered): joins of two tables that are both in order
ORDERED_CASES = [("asof_ordered", case_asof_ordered, 1)]
# a fourth pool (--wire): the narrowing / widening kernels of the key shuffle
WIRE_CASES = [("wire", case_wire, 1)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cases", type=int, default=0, help="stop after this many cases (0 = by time only)")
    ap.add_argument("--case-seed", type=int, default=0, help="run this one case again (the seed a log line names)")
    ap.add_argument("--extra", action="store_true",
                    help="the second pool: joins of tables wider than one sort word, shuffle routing, write layout")
    ap.add_argument("--ordered", action="store_true", help="the third pool: joins of two tables already in (key, ts) order")
    ap.add_argument("--wire", action="store_true", help="the fourth pool: wire encoding of the key shuffle's columns")
    ap.add_argument("--only", default="", help="comma-separated operator names")
    ap.add_argument("--log", default=os.path.join(ROOT, "gpurun_out", "fuzz_ops.log"))
    args = ap.parse_args()
    os.makedirs(os.path.dirname(args.log), exist_ok=True)
    pool = [c for c in (WIRE_CASES if args.wire else ORDERED_CASES if args.ordered else EXTRA_CASES if args.extra else CASES) if not args.only or c[0] in args.only.split(",")]
    weights = np.array([c[2] for c in pool], dtype=np.float64)
    weights /= weights.sum()
    t_end = time.time() + args.seconds
    done, bad, per_op = 0, [], {}
    with open(args.log, "w") as log:
        if args.case_seed:
            args.cases = 1
        while time.time() < t_end and (args.cases == 0 or done < args.cases):
            case_seed = args.case_seed or args.seed * 1_000_003 + done
            rng = np.random.default_rng(case_seed)
            LAST["desc"] = ""
            name, fn, _ = pool[int(rng.choice(len(pool), p=weights))]
            t0 = time.time()
            log.write("%5d start    %-13s seed=%d\n" % (done + 1, name, case_seed))
            log.flush()
            try:
                desc = fn(rng)
                status = "ok"
            except Exception as err:  # noqa: BLE001 (a harness: every failure is a finding)
                text = [t.strip() for t in str(err).strip().splitlines() if t.strip()]
                desc = "%s: %s | %s" % (type(err).__name__, " / ".join(text[:7]), LAST["desc"])
                status = "MISMATCH" if isinstance(err, AssertionError) else "ERROR"
                bad.append(dict(op=name, case_seed=case_seed, status=status, what=desc,
                                trace=traceback.format_exc().splitlines()[-6:]))
            per_op.setdefault(name, [0, 0])[0 if status == "ok" else 1] += 1
            done += 1
            log.write("%5d %-8s %-13s seed=%d %.2fs %s\n" % (done, status, name, case_seed, time.time() - t0, desc))
            log.flush()
            torch.cuda.synchronize()
    print(json.dumps(dict(cases=done, disagreements=len(bad), per_operator={k: dict(ok=v[0], bad=v[1]) for k, v in per_op.items()},
                          seed=args.seed, seconds=args.seconds, first_bad=bad[:10])))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
