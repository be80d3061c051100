"""C oracle (sort-and-scan formulation) against the definition-level brute force on small
random inputs: ties, NULLs, sequence numbers, skew brackets, empty sides."""
import math

import numpy as np
import pytest

from oracle import brute

BASE = 1_596_240_000_000_000  # 2020-08-01 00:00:00 UTC in microseconds


def _rand_asof(rng, nL, nR, nkeys, span_us, step_us, C, null_p, with_seq):
    key_l = rng.integers(0, nkeys, nL).astype(np.int32)
    key_r = rng.integers(0, nkeys, nR).astype(np.int32)
    ts_l = BASE + rng.integers(0, span_us // step_us + 1, nL) * step_us
    ts_r = BASE + rng.integers(0, span_us // step_us + 1, nR) * step_us
    valid = [(rng.random(nR) >= null_p).astype(np.uint8) for _ in range(C)]
    if C > 1:
        valid[0] = None  # a column without NULLs
    seq = seq_valid = None
    if with_seq:
        seq = rng.integers(-3, 4, nR)
        seq_valid = (rng.random(nR) >= 0.2).astype(np.uint8)
    return key_l, ts_l, key_r, ts_r, seq, seq_valid, valid


@pytest.mark.parametrize("seed", range(12))
def test_asof_matches_brute(orc, seed):
    rng = np.random.default_rng(seed)
    nL, nR = int(rng.integers(0, 60)), int(rng.integers(0, 60))
    with_seq = seed % 3 == 1
    skew = seed % 4 >= 2
    key_l, ts_l, key_r, ts_r, seq, seq_valid, valid = _rand_asof(
        rng, nL, nR, 3, 40_000_000, 500_000, 3, 0.3, with_seq)
    for skip_nulls in (True, False):
        if skew and not skip_nulls:
            continue
        kw = dict(seq_r=seq, seq_r_valid=seq_valid, valid_r=valid, skip_nulls=skip_nulls)
        if skew:
            kw.update(ts_partition_val=[10, 3, 7.5][seed % 3], fraction=[0.1, 0.5, 0.25][seed % 3])
        idx, missing = orc.asof_join(key_l, ts_l, key_r, ts_r, **kw)
        bidx, bmissing = brute.asof_join(key_l, ts_l, key_r, ts_r, **kw)
        # rows tying on every order key are resolved by input order in both
        assert idx.tolist() == bidx
        assert missing.tolist() == bmissing


@pytest.mark.parametrize("seed", range(8))
def test_range_stats_matches_brute(orc, seed):
    rng = np.random.default_rng(100 + seed)
    n = int(rng.integers(1, 80))
    key = rng.integers(0, 3, n).astype(np.int32)
    ts = BASE + rng.integers(0, 60, n) * 700_000 - (5_000_000 if seed == 0 else 0)
    as_int = seed % 2 == 1
    x = rng.integers(-50, 50, n) if as_int else np.round(rng.normal(100, 5, n), 3)
    xv = (rng.random(n) >= 0.2).astype(np.uint8)
    W = int(rng.integers(0, 12))
    out = orc.range_stats(key, ts, x, xv, W)
    rows = brute.range_stats(key.tolist(), ts.tolist(), x.tolist(), xv.tolist(), W)
    assert out["perm"].tolist() == [r["row"] for r in rows]
    for p, r in enumerate(rows):
        assert out["count"][p] == r["count"]
        assert bool(out["stat_valid"][p]) == (r["count"] > 0)
        if r["count"]:
            assert out["min"][p] == r["min"] and out["max"][p] == r["max"]
            assert out["sum"][p] == r["sum"]
            assert out["mean"][p] == r["mean"]
        assert bool(out["stddev_valid"][p]) == (r["stddev"] is not None)
        if r["stddev"] is not None:
            assert out["stddev"][p] == r["stddev"]
        assert bool(out["zscore_valid"][p]) == (r["zscore"] is not None)
        if r["zscore"] is not None:
            assert out["zscore"][p] == r["zscore"]


@pytest.mark.parametrize("seed", range(6))
def test_ema_and_lookback_match_brute(orc, seed):
    rng = np.random.default_rng(200 + seed)
    n = int(rng.integers(1, 70))
    key = rng.integers(0, 3, n).astype(np.int32)
    ts = BASE + rng.integers(0, 50, n) * 900_000
    seq = rng.integers(0, 5, n) if seed % 2 else None
    seq_valid = (rng.random(n) >= 0.2).astype(np.uint8) if seed % 2 else None
    x = rng.normal(0, 3, n)
    xv = (rng.random(n) >= 0.25).astype(np.uint8)
    window = int(rng.integers(1, 9))
    perm, out = orc.ema(key, ts, x, xv, window, 0.2, seq=seq, seq_valid=seq_valid)
    bperm, bout = brute.ema(key, ts, x, xv, window, 0.2, seq=seq, seq_valid=seq_valid)
    assert perm.tolist() == bperm
    assert out.tolist() == bout  # bit-exact: same weights, same add order
    y = rng.normal(5, 1, n)
    L = int(rng.integers(1, 6))
    perm, ln, feat, fv = orc.lookback(key, ts, [x, y], [xv, None], L, seq=seq, seq_valid=seq_valid)
    bperm, bfeat = brute.lookback(key, ts, [x, y], [xv, None], L, seq=seq, seq_valid=seq_valid)
    assert perm.tolist() == bperm
    for p in range(n):
        assert ln[p] == len(bfeat[p])
        for l in range(ln[p]):
            for f in range(2):
                want = bfeat[p][l][f]
                assert bool(fv[p, l, f]) == (want is not None)
                if want is not None:
                    assert feat[p, l, f] == want


@pytest.mark.parametrize("seed", range(6))
def test_resample_matches_brute(orc, seed):
    rng = np.random.default_rng(300 + seed)
    n = int(rng.integers(1, 90))
    key = rng.integers(0, 3, n).astype(np.int32)
    ts = BASE + rng.integers(-40, 40, n) * 7_000_000
    if seed == 0:
        ts = ts - 2 * BASE  # negative epoch times exercise the floor-mod bucket rule
    a = np.round(rng.normal(0, 3, n), 2)
    b = rng.integers(-5, 5, n)
    av = (rng.random(n) >= 0.3).astype(np.uint8)
    dur = 60_000_000
    buckets = brute.resample_buckets(key, ts, dur)
    for func in ("mean", "min", "max", "floor", "ceil"):
        r = orc.resample(key, ts, dur, func, [a, b], [av, None])
        assert r["n"] == len(buckets)
        for bi, ((k, start), rows) in enumerate(buckets.items()):
            assert r["key"][bi] == k and r["start"][bi] == start
            if func in ("floor", "ceil"):
                def tup(i):
                    return (int(ts[i]), (0, 0.0) if not av[i] else (1, float(a[i])), int(b[i]))
                best = min(rows, key=tup) if func == "floor" else max(rows, key=tup)
                assert tup(r["row"][bi]) == tup(best)
                continue
            va = [float(a[i]) for i in rows if av[i]]
            vb = [int(b[i]) for i in rows]
            assert bool(r["valid"][0][bi]) == bool(va)
            if func == "mean":
                if va:
                    assert math.isclose(r["f64"][0][bi], sum(va) / len(va), rel_tol=1e-12)
                assert math.isclose(r["f64"][1][bi], sum(vb) / len(vb), rel_tol=1e-12)
            else:
                pick = min if func == "min" else max
                if va:
                    assert r["f64"][0][bi] == pick(va)
                assert r["i64"][1][bi] == pick(vb)
