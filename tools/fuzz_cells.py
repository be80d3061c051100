"""Differential check of the cell paths (resample mean / min / max / floor / ceil, withGroupedStats,
withRangeStats with any_order) against the ordered paths of the same library, on random tables:
row counts, keys, bucket widths, spans, row orders, NULL fractions and special values are drawn per
case.  Both sides run on the GPU; the switches are read by the library at call time.
python tools/fuzz_cells.py [--cases N] [--seed S]      exit status 1 on the first disagreement"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tempo_amd import _lib, ops  # noqa: E402

OFF = ["TEMPO_NO_DENSE_MEAN", "TEMPO_NO_DENSE_STATS", "TEMPO_NO_DENSE_PICK", "TEMPO_NO_RANGE_CELLS"]


def dev(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()


TAKEN = {}  # how many calls went through each cell kernel (the library's own launch log)


def both(fn):
    for k in OFF:
        os.environ.pop(k, None)
    os.environ["TEMPO_DENSE_MIN_ROWS"] = "1"
    _lib.profile_enable(True)
    _lib.profile_reset()
    a = fn()
    torch.cuda.synchronize()
    for name in {name for name, _, _ in _lib.profile_read()}:
        if name.startswith("dense_") or name.startswith("cell_"):
            TAKEN[name] = TAKEN.get(name, 0) + 1
    _lib.profile_enable(False)
    for k in OFF:
        os.environ[k] = "1"
    b = fn()
    for k in OFF:
        os.environ.pop(k, None)
    return a, b


def same(a, b, what, where=None, exact=True, scale=1.0):
    a, b = a.cpu().numpy(), b.cpu().numpy()
    if where is not None:
        a, b = a[where], b[where]
    if exact:
        ok = np.array_equal(a, b, equal_nan=a.dtype.kind == "f")
    else:
        ok = np.allclose(a, b, rtol=1e-9, atol=1e-12 * scale, equal_nan=True)
    if not ok:
        bad = np.flatnonzero(~np.isclose(a.astype(np.float64), b.astype(np.float64), rtol=1e-9, atol=1e-12 * scale,
                                         equal_nan=True))
        raise AssertionError("%s differs at %d places, first %s: %r vs %r" % (
            what, len(bad), bad[:3], a[bad[:3]], b[bad[:3]]))


def table(rng):
    n = int(rng.choice([1000, 5000, 70_000, 300_000, 600_000]))
    nk = int(rng.choice([1, 2, 9, 100, 700, 2048, 3000]))
    dur = int(rng.choice([1_000_000, 60_000_000, 3_600_000_000, 86_400_000_000]))
    span = int(dur * rng.choice([0.5, 3, 40, 1000, 20000]))
    ts = int(rng.choice([0, -5 * dur, 1_596_240_000_000_000, -span // 2])) + np.sort(rng.integers(0, max(span, 2), n))
    if rng.random() < 0.3:
        ts = (ts // 1_000_000) * 1_000_000
    key = rng.integers(0, nk, n).astype(np.int32)
    if rng.random() < 0.2:
        key = (key // 3 * 3).astype(np.int32)          # gaps in the dictionary
    order = rng.choice(["time", "sorted", "random", "runs"])
    if order == "sorted":
        o = np.lexsort((ts, key))
    elif order == "random":
        o = rng.permutation(n)
    elif order == "runs":
        r = int(rng.integers(2, 9))
        o = np.concatenate([np.arange(i, n, r) for i in range(r)])
    else:
        o = np.arange(n)
    key, ts = key[o], ts[o]
    return n, nk, dur, span, key, ts, order


def column(rng, n, kind):
    if kind == "i64":
        x = rng.integers(-10**12, 10**12, n)
    elif kind == "walk":
        x = 100 + np.cumsum(rng.normal(0, 0.01, n))
    elif kind == "const":
        x = np.full(n, 7.25)
    else:
        x = np.round(rng.normal(-1.7, 3.86, n), 2)
    p = rng.choice([0.0, 0.05, 0.6, 1.0])
    v = None if p == 0.0 else (rng.random(n) >= p).astype(np.uint8)
    return x, v


def one_case(seed):
    rng = np.random.default_rng(seed)
    n, nk, dur, span, key, ts, order = table(rng)
    tag = "seed %d: n=%d keys=%d dur=%d span=%d order=%s" % (seed, n, nk, dur, span, order)
    k_dev = None if nk == 1 and rng.random() < 0.5 else dev(key)
    t_dev = dev(ts)
    nc = int(rng.integers(1, 5))
    cols, valid = [], []
    for _ in range(nc):
        x, v = column(rng, n, rng.choice(["i64", "walk", "const", "norm"]))
        if x.dtype.kind == "f" and rng.random() < 0.3:
            x = x.copy()
            x[rng.random(n) < 0.001] = np.inf
            x[rng.random(n) < 0.001] = -np.inf
        cols.append(x)
        valid.append(v)
    dc, dv = [dev(c) for c in cols], [dev(v) for v in valid]
    for func in ("mean", "min", "max", "floor", "ceil"):
        a, b = both(lambda: ops.resample(k_dev, t_dev, nk, dur, func, dc, dv))
        w = tag + " resample(%s, %d cols)" % (func, nc)
        assert a["n"] == b["n"], w
        same(a["key"], b["key"], w + " key")
        same(a["start"], b["start"], w + " start")
        if func in ("floor", "ceil"):
            same(a["row"], b["row"], w + " row")
            continue
        same(a["valid"], b["valid"], w + " valid")
        ok = a["valid"].cpu().numpy() != 0
        for c in range(nc):
            if a["is_i64"][c] and func != "mean":
                same(a["i64"][c], b["i64"][c], w + " i64[%d]" % c, ok[c])
            else:
                same(a["f64"][c], b["f64"][c], w + " f64[%d]" % c, ok[c], exact=func != "mean",
                     scale=1e12 if a["is_i64"][c] else 100.0)
    # grouped stats: NaNs as well
    gcols = []
    for x in cols[:2]:
        if x.dtype.kind == "f" and rng.random() < 0.5:
            x = x.copy()
            x[rng.random(n) < 0.0005] = np.nan
        gcols.append(x)
    gd = [dev(c) for c in gcols]
    a, b = both(lambda: ops.grouped_stats(k_dev, t_dev, nk, dur, gd, dv[:2]))
    w = tag + " grouped_stats"
    assert a["n"] == b["n"], w
    same(a["key"], b["key"], w + " key")
    same(a["start"], b["start"], w + " start")
    for c in range(len(gd)):
        same(a["count"][c], b["count"][c], w + " count[%d]" % c)
        cnt = a["count"][c].cpu().numpy()
        ok, ok2 = cnt > 0, cnt > 1
        scale = 1e12 if a["is_i64"][c] else 100.0
        same(a["min"][c], b["min"][c], w + " min[%d]" % c, ok)
        same(a["max"][c], b["max"][c], w + " max[%d]" % c, ok)
        same(a["sum"][c], b["sum"][c], w + " sum[%d]" % c, ok, exact=bool(a["is_i64"][c]), scale=scale * 1e4)
        same(a["mean"][c], b["mean"][c], w + " mean[%d]" % c, ok, exact=False, scale=scale)
        same(a["stddev"][c], b["stddev"][c], w + " stddev[%d]" % c, ok2, exact=False, scale=scale)
    # range stats, any_order: compared row by row
    x, v = cols[0], valid[0]
    if x.dtype.kind == "f":
        wsecs = int(rng.choice([0, 1, 10, 600, 100000]))
        xd, vd = dev(x), dev(v)
        a, b = both(lambda: ops.range_stats(k_dev, t_dev, xd, vd, nk, wsecs, any_order=True))
        w = tag + " range_stats(W=%d)" % wsecs
        pa, pb = a["perm"].cpu().numpy().astype(np.int64), b["perm"].cpu().numpy().astype(np.int64)
        ia, ib = np.argsort(pa, kind="stable"), np.argsort(pb, kind="stable")
        assert np.array_equal(pa[ia], np.arange(n)) and np.array_equal(pb[ib], np.arange(n)), w + " perm"
        va, vb = a["valid"].cpu().numpy()[ia], b["valid"].cpu().numpy()[ib]
        assert np.array_equal(va, vb), w + " valid"

        def by_row(r, i, name):
            return torch.from_numpy(r[name].cpu().numpy()[i])
        same(by_row(a, ia, "count"), by_row(b, ib, "count"), w + " count")
        ok, ok2, okz = (va & 1) != 0, (va & 2) != 0, (va & 4) != 0
        same(by_row(a, ia, "min"), by_row(b, ib, "min"), w + " min", ok)
        same(by_row(a, ia, "max"), by_row(b, ib, "max"), w + " max", ok)
        same(by_row(a, ia, "mean"), by_row(b, ib, "mean"), w + " mean", ok, exact=False, scale=100.0)
        same(by_row(a, ia, "sum"), by_row(b, ib, "sum"), w + " sum", ok, exact=False, scale=1e7)
        same(by_row(a, ia, "stddev"), by_row(b, ib, "stddev"), w + " stddev", ok2, exact=False, scale=100.0)
        # zscore = (x - mean) / stddev magnifies the last bits of a tiny stddev: compared through stddev
        za, zb = by_row(a, ia, "zscore").numpy()[okz], by_row(b, ib, "zscore").numpy()[okz]
        sa = by_row(a, ia, "stddev").numpy()[okz]
        assert np.allclose(za * sa, zb * sa, rtol=1e-6, atol=1e-9, equal_nan=True), w + " zscore"
    return tag


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    for s in range(a.seed, a.seed + a.cases):
        try:
            print(one_case(s), "ok", flush=True)
        except AssertionError as e:
            print("FAIL", e, flush=True)
            sys.exit(1)
    print("calls through the cell kernels:", dict(sorted(TAKEN.items())), flush=True)


if __name__ == "__main__":
    main()
