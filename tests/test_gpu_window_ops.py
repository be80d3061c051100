"""HIP EMA / withLookbackFeatures / resample / key bucketing against the CPU oracle, through
the C ABI.  EMA and lookback are bit-exact (same taps, same add order / pure copies); resample
bucket membership and min/max/floor/ceil are bit-exact, means within 1e-9 relative."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu

BASE = 1_596_240_000_000_000


def _dev(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope="module")
def ops():
    from tempo_amd import ops
    return ops


def _series(rng, n, nkeys, with_seq, sort=False, step_us=400_000, span=3000):
    key = rng.integers(0, nkeys, n).astype(np.int32)
    ts = BASE + rng.integers(0, span, n) * step_us
    seq = rng.integers(-5, 50, n) if with_seq else None
    seq_valid = (rng.random(n) >= 0.1).astype(np.uint8) if with_seq else None
    if sort:
        o = np.lexsort((ts, key))
        key, ts = key[o], ts[o]
    return key, ts, seq, seq_valid


@pytest.mark.parametrize("seed,window,with_seq", [(0, 30, False), (1, 5, True), (2, 1, False), (3, 200, True)])
def test_ema_bit_exact(orc, ops, seed, window, with_seq):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 40_000))
    key, ts, seq, seq_valid = _series(rng, n, 6, with_seq)
    x = rng.normal(0, 3, n)
    xv = (rng.random(n) >= 0.1).astype(np.uint8)
    perm, out = ops.ema(_dev(key), _dev(ts), _dev(x), _dev(xv), 6, window=window, exp_factor=0.2,
                        seq=_dev(seq), seq_valid=_dev(seq_valid))
    operm, oout = orc.ema(key, ts, x, xv, window, 0.2, seq=seq, seq_valid=seq_valid)
    assert np.array_equal(perm.cpu().numpy().astype(np.int64), operm)
    assert np.array_equal(out.cpu().numpy(), oout)  # same weights, same add order


def test_ema_golden_scala_vector(ops, golden):
    from conftest import ts_us
    fx = golden["ema_scala"]
    sym = [r[0] for r in fx["data"]]
    key = np.array([sorted(set(sym)).index(s) for s in sym], np.int32)
    ts = np.array([ts_us(r[1]) for r in fx["data"]])
    x = np.array([r[2] for r in fx["data"]])
    perm, out = ops.ema(_dev(key), _dev(ts), _dev(x), None, 2, window=fx["python_window"],
                        exp_factor=fx["exp_factor"])
    assert perm.cpu().numpy().tolist() == list(range(6))
    assert out.cpu().numpy().tolist() == fx["expected_ema"]


def test_ema_presorted_single_series(orc, ops):
    rng = np.random.default_rng(5)
    n = 50_000
    ts = BASE + np.arange(n) * 5_000  # 5 ms sampling: many rows share a floor-second
    x = rng.normal(-1.7, 3.86, n)
    perm, out = ops.ema(None, _dev(ts), _dev(x), None, 1, window=30)
    operm, oout = orc.ema(None, ts, x, None, 30, 0.2)
    assert np.array_equal(perm.cpu().numpy().astype(np.int64), operm)
    assert np.array_equal(out.cpu().numpy(), oout)


@pytest.mark.parametrize("seed,L,exact,with_seq", [(0, 10, True, False), (1, 3, False, True), (2, 1, True, True),
                                                   (3, 40, False, False)])
def test_lookback(orc, ops, seed, L, exact, with_seq):
    rng = np.random.default_rng(20 + seed)
    n = int(rng.integers(1, 30_000))
    key, ts, seq, seq_valid = _series(rng, n, 5, with_seq)
    xs = [rng.normal(0, 1, n), rng.normal(5, 2, n), rng.integers(0, 9, n).astype(np.float64)]
    xv = [None, (rng.random(n) >= 0.2).astype(np.uint8), None]
    rows, lens, feat, fv = ops.lookback(_dev(key), _dev(ts), [_dev(x) for x in xs], [_dev(v) for v in xv], 5, L,
                                        exact_size=exact, seq=_dev(seq), seq_valid=_dev(seq_valid))
    operm, olen, ofeat, ofv = orc.lookback(key, ts, xs, xv, L, seq=seq, seq_valid=seq_valid)
    if exact:
        keep = olen == L
        assert np.array_equal(rows.cpu().numpy().astype(np.int64), operm[keep])
        assert np.array_equal(feat.cpu().numpy(), ofeat[keep])
        assert np.array_equal(fv.cpu().numpy(), ofv[keep])
    else:
        assert np.array_equal(rows.cpu().numpy().astype(np.int64), operm)
        assert np.array_equal(lens.cpu().numpy(), olen)
        assert np.array_equal(feat.cpu().numpy(), ofeat)
        assert np.array_equal(fv.cpu().numpy(), ofv)


@pytest.mark.parametrize("func", ["mean", "min", "max", "floor", "ceil"])
@pytest.mark.parametrize("seed", [0, 1])
def test_resample(orc, ops, func, seed):
    rng = np.random.default_rng(40 + seed)
    n = int(rng.integers(1, 60_000))
    key = rng.integers(0, 9, n).astype(np.int32)
    ts = BASE + rng.integers(-2000, 20_000, n) * 5_000 * (1 + seed * 200)
    if seed == 1:
        ts = ts - 2 * BASE  # pre-epoch timestamps: floor-mod bucket rule
    a = np.round(rng.normal(-1.7, 3.86, n), 1)  # coarse values: ties in the struct ordering
    b = rng.integers(-3, 4, n)
    c = rng.normal(8.93, 2.26, n)
    av = (rng.random(n) >= 0.2).astype(np.uint8)
    cols, valid = [a, b, c], [av, None, None]
    dur = 60_000_000
    r = ops.resample(_dev(key), _dev(ts), 9, dur, func, [_dev(x) for x in cols], [_dev(v) for v in valid])
    o = orc.resample(key, ts, dur, func, cols, valid)
    assert r["n"] == o["n"]
    assert np.array_equal(r["key"].cpu().numpy(), o["key"])
    assert np.array_equal(r["start"].cpu().numpy(), o["start"])  # bucket membership bit-exact
    if func in ("floor", "ceil"):
        assert np.array_equal(r["row"].cpu().numpy().astype(np.int64), o["row"])
        return
    assert np.array_equal(r["valid"].cpu().numpy(), o["valid"])
    if func == "mean":
        np.testing.assert_allclose(r["f64"].cpu().numpy(), o["f64"], rtol=1e-9, atol=1e-12)
    else:
        assert np.array_equal(r["f64"].cpu().numpy(), o["f64"])
        assert np.array_equal(r["i64"].cpu().numpy(), o["i64"])


def test_resample_golden(ops, golden):
    from conftest import ts_us
    fx = golden["resample"]
    ts = np.array([ts_us(r[2]) for r in fx["data"]])
    key = np.zeros(len(ts), np.int32)
    pr = np.array([r[3] for r in fx["data"]], np.float32).astype(np.float64)
    pr2 = np.array([r[4] for r in fx["data"]], np.float32).astype(np.float64)
    r = ops.resample(_dev(key), _dev(ts), 1, 60_000_000, "floor", [_dev(pr), _dev(pr2)], [None, None])
    e = fx["floor_min"]["expected"]
    assert r["n"] == len(e)
    rows = r["row"].cpu().numpy()
    for b, erow in enumerate(e):
        assert int(r["start"][b]) == ts_us(erow[1])
        assert np.float32(pr[rows[b]]) == np.float32(erow[2]) and pr2[rows[b]] == erow[4]
    r = ops.resample(_dev(key), _dev(ts), 1, 300_000_000, "mean", [_dev(pr), _dev(pr2)], [None, None])
    e = fx["mean_5min"]["expected"]
    assert r["n"] == len(e)
    f = r["f64"].cpu().numpy()
    for b, erow in enumerate(e):
        assert int(r["start"][b]) == ts_us(erow[1])
        assert round(float(f[0][b]), 2) == erow[3] and f[1][b] == erow[4]


def test_hash_partition_matches_host_owner(ops):
    from tempo_amd import _lib
    rng = np.random.default_rng(7)
    n, parts = 100_000, 8
    key = rng.integers(0, 5000, n).astype(np.int32)
    perm, counts = ops.hash_partition(_dev(key), parts)
    perm, counts = perm.cpu().numpy(), counts.cpu().numpy()
    owner = np.array([_lib.lib().tmp_key_owner(int(k), parts) for k in range(5000)])[key]
    assert np.array_equal(counts, np.bincount(owner, minlength=parts))
    assert np.array_equal(perm, np.argsort(owner, kind="stable"))


def test_ema_time_ordered_feed(orc, ops):
    rng = np.random.default_rng(31)
    n = 60_000
    key = rng.integers(0, 9, n).astype(np.int32)
    ts = BASE + np.sort(rng.integers(0, 500_000, n)) * 5_000
    x = rng.normal(0, 1, n)
    perm, out = ops.ema(_dev(key), _dev(ts), _dev(x), None, 9, window=30)
    operm, oout = orc.ema(key, ts, x, None, 30, 0.2)
    assert np.array_equal(perm.cpu().numpy().astype(np.int64), operm)
    assert np.array_equal(out.cpu().numpy(), oout)


@pytest.mark.parametrize("aligned", [True, False])
def test_sort_order_seconds_cast_extreme_timestamps(ops, aligned):
    """cast(ts as long) = floor(ts / 1e6) over the whole int64 range, both signs, at bucket edges."""
    rng = np.random.default_rng(5)
    n = 200_003
    ts = rng.integers(-2**63, 2**63 - 1, n, dtype=np.int64, endpoint=True)
    edges = rng.integers(-9_000_000_000_000, 9_000_000_000_000, 20_000) * 1_000_000
    ts[:60_000] = np.concatenate([edges - 1, edges, edges + 1])
    ts[60_000:60_004] = [np.iinfo(np.int64).min, np.iinfo(np.int64).max, -1, 0]
    rng.shuffle(ts)
    key = rng.integers(0, 5, n).astype(np.int32)
    off = 0 if aligned else 1       # an odd element offset takes the row-at-a-time scan
    kd, td = _dev(key)[off:], _dev(ts)[off:]
    perm = ops.sort_order(kd, td, 5, floor_seconds=True).cpu().numpy().astype(np.int64)
    key, ts = key[off:], ts[off:]
    sec = np.floor_divide(ts, 1_000_000)
    assert np.array_equal(perm, np.lexsort((np.arange(key.size), sec, key)))


@pytest.mark.parametrize("seed,dur", [(0, 60_000_000), (1, 1_000_000), (2, 86_400_000_000)])
def test_grouped_stats(orc, ops, seed, dur):
    """withGroupedStats core against the oracle: buckets, count, min, max bit-exact (int sums too);
    mean / double sum / stddev within 1e-9 relative."""
    rng = np.random.default_rng(seed)
    n, nk = 60_000, 7
    key = rng.integers(0, nk, n).astype(np.int32)
    ts = 1_596_240_000_000_000 + rng.integers(0, 3 * 3600 * 1_000_000, n)
    xd = 100 + np.cumsum(rng.normal(0, 0.01, n))
    xd[rng.random(n) < 0.001] = np.nan
    xi = rng.integers(-10**12, 10**12, n)
    vd = (rng.random(n) > 0.05).astype(np.uint8)
    vi = (rng.random(n) > 0.5).astype(np.uint8)
    r = ops.grouped_stats(_dev(key), _dev(ts), nk, dur, [_dev(xd), _dev(xi), _dev(xd)], [_dev(vd), _dev(vi), None])
    e = orc.grouped_stats(key, ts, dur, [xd, xi, xd], [vd, vi, None])
    assert r["n"] == e["n"]
    assert np.array_equal(r["key"].cpu().numpy(), e["key"]) and np.array_equal(r["start"].cpu().numpy(), e["start"])
    for c in range(3):
        cnt = r["count"][c].cpu().numpy()
        assert np.array_equal(cnt, e["count"][c]), c
        ok, ok2 = cnt > 0, cnt > 1
        for name in ("min", "max"):
            assert np.array_equal(r[name][c].cpu().numpy()[ok], e[name][c][ok], equal_nan=True), (c, name)
        if e["is_i64"][c]:
            assert np.array_equal(r["sum"][c].cpu().numpy()[ok], e["sum"][c][ok])
        else:
            np.testing.assert_allclose(r["sum"][c].cpu().numpy()[ok], e["sum"][c][ok], rtol=1e-9, equal_nan=True)
        np.testing.assert_allclose(r["mean"][c].cpu().numpy()[ok], e["mean"][c][ok], rtol=1e-9, equal_nan=True)
        scale = 1e12 if e["is_i64"][c] else 100.0
        np.testing.assert_allclose(r["stddev"][c].cpu().numpy()[ok2], e["stddev"][c][ok2], rtol=1e-9,
                                   atol=1e-12 * scale, equal_nan=True)


def test_grouped_stats_empty_and_single_series(orc, ops):
    e = ops.grouped_stats(None, _dev(np.zeros(0, np.int64)), 1, 60_000_000, [], [])
    assert e["n"] == 0
    ts = 1_596_240_000_000_000 + np.arange(10) * 30_000_000
    x = np.arange(10, dtype=np.float64)
    r = ops.grouped_stats(None, _dev(ts), 1, 60_000_000, [_dev(x)], [None])
    assert r["n"] == 5 and r["count"][0].cpu().numpy().tolist() == [2] * 5
    assert r["mean"][0].cpu().numpy().tolist() == [0.5, 2.5, 4.5, 6.5, 8.5]
    np.testing.assert_allclose(r["stddev"][0].cpu().numpy(), np.full(5, np.sqrt(0.5)), rtol=1e-12)


def _sampled_table(rng, n_keys, buckets, freq, p_keep, p_null, n_cols):
    """A sampled table: per key a random subset of a regular grid, values with NULLs."""
    keys, stamps = [], []
    for k in range(n_keys):
        keep = np.flatnonzero(rng.random(buckets) < p_keep)
        keys.append(np.full(keep.size, k, np.int32))
        stamps.append(1_577_836_800_000_000 + keep.astype(np.int64) * freq)
    key, ts = np.concatenate(keys), np.concatenate(stamps)
    o = rng.permutation(key.size)
    key, ts = key[o], ts[o]
    cols = [rng.normal(100, 5, key.size) for _ in range(n_cols)]
    valid = [(rng.random(key.size) >= p_null).astype(np.uint8) for _ in range(n_cols)]
    return key, ts, cols, valid


@pytest.mark.parametrize("method", ["zero", "null", "bfill", "ffill", "linear"])
@pytest.mark.parametrize("freq,p_keep,p_null", [(30_000_000, 0.4, 0.3), (1_000_000, 0.9, 0.05), (3_600_000_000, 0.1, 0.6)])
def test_interpolate(orc, ops, method, freq, p_keep, p_null):
    """Interpolation core against the oracle: rows, timestamps, flags and every filled value are
    bit-exact (same fp64 operations in the same order, no contraction)."""
    rng = np.random.default_rng(int(freq % 97) + len(method))
    key, ts, cols, valid = _sampled_table(rng, 11, 700, freq, p_keep, p_null, 3)
    valid[2] = None
    r = ops.interpolate(_dev(key), _dev(ts), 11, freq, method, [_dev(c) for c in cols],
                        [None if v is None else _dev(v) for v in valid])
    e = orc.interpolate(key, ts, freq, method, cols, valid)
    assert r["n"] == e["n"]
    assert np.array_equal(r["src"].cpu().numpy(), e["src"])
    assert np.array_equal(r["key"].cpu().numpy(), e["key"]) and np.array_equal(r["ts"].cpu().numpy(), e["ts"])
    assert np.array_equal(r["ts_interp"].cpu().numpy(), e["ts_interp"])
    for c in range(3):
        ok = r["valid"][c].cpu().numpy()
        assert np.array_equal(ok, e["valid"][c]), c
        assert np.array_equal(r["interp"][c].cpu().numpy(), e["interp"][c]), c
        got, want = r["val"][c].cpu().numpy(), e["val"][c]
        assert np.array_equal(got[ok == 1].view(np.int64), want[ok == 1].view(np.int64)), c


def test_interpolate_edges(orc, ops):
    e = ops.interpolate(None, _dev(np.zeros(0, np.int64)), 1, 1_000_000, "linear", [], [])
    assert e["n"] == 0
    # one row per key: every key yields exactly its own row
    key = np.arange(5, dtype=np.int32)
    ts = np.full(5, 1_577_836_800_000_000, np.int64)
    x = np.arange(5, dtype=np.float64)
    r = ops.interpolate(_dev(key), _dev(ts), 5, 60_000_000, "linear", [_dev(x)], [None])
    assert r["n"] == 5 and r["val"][0].cpu().numpy().tolist() == x.tolist()
    assert not r["ts_interp"].cpu().numpy().any()
    # no key column, pre-epoch timestamps (unix_timestamp floors), a wide gap
    ts = np.array([-3_600_000_000, -1_800_000_000, 7_200_000_000], np.int64)
    x = np.array([1.0, 0.0, 9.0])
    v = np.array([1, 0, 1], np.uint8)
    r = ops.interpolate(None, _dev(ts), 1, 600_000_000, "linear", [_dev(x)], [_dev(v)])
    o = orc.interpolate(None, ts, 600_000_000, "linear", [x], [v])
    assert r["n"] == o["n"] == 19
    assert np.array_equal(r["val"][0].cpu().numpy().view(np.int64), o["val"][0].view(np.int64))
    # rows closer than the frequency are rejected, as Spark's sequence() rejects them
    with pytest.raises(ValueError, match="closer than freq_us"):
        ops.interpolate(None, _dev(np.array([0, 10, 20], np.int64)), 1, 15, "zero", [], [])


@pytest.mark.parametrize("func", ["mean", "min", "max"])
def test_resample_big_buckets(orc, ops, func):
    """Buckets of more than 32 768 rows are cut into tiles and merged: day buckets over an
    intraday series, next to small buckets in the same call."""
    rng = np.random.default_rng(21)
    n, nk = 400_000, 3
    key = rng.integers(0, nk, n).astype(np.int32)
    day = 86_400_000_000
    ts = 1_596_240_000_000_000 + rng.integers(0, 3 * day, n)
    ts[key == 2] = 1_596_240_000_000_000 + rng.integers(0, 200, (key == 2).sum()) * day   # key 2: tiny buckets
    xd = 50 + np.cumsum(rng.normal(0, 0.01, n))
    xi = rng.integers(-10**9, 10**9, n)
    vd = (rng.random(n) > 0.1).astype(np.uint8)
    r = ops.resample(_dev(key), _dev(ts), nk, day, func, [_dev(xd), _dev(xi)], [_dev(vd), None])
    e = orc.resample(key, ts, day, func, [xd, xi], [vd, None])
    assert r["n"] == e["n"] and np.array_equal(r["start"].cpu().numpy(), e["start"])
    assert np.array_equal(r["valid"].cpu().numpy(), e["valid"])
    if func == "mean":
        np.testing.assert_allclose(r["f64"].cpu().numpy(), e["f64"], rtol=1e-9)
    else:
        assert np.array_equal(r["f64"].cpu().numpy()[0], e["f64"][0])
        assert np.array_equal(r["i64"].cpu().numpy()[1], e["i64"][1])


def test_grouped_stats_big_buckets(orc, ops):
    rng = np.random.default_rng(22)
    n = 300_000
    day = 86_400_000_000
    ts = 1_596_240_000_000_000 + np.sort(rng.integers(0, 2 * day, n))
    xd = 1000 + np.cumsum(rng.normal(0, 0.01, n)) + 0.001 * np.arange(n)
    xi = rng.integers(-10**12, 10**12, n)
    vd = (rng.random(n) > 0.2).astype(np.uint8)
    r = ops.grouped_stats(None, _dev(ts), 1, day, [_dev(xd), _dev(xi)], [_dev(vd), None])
    e = orc.grouped_stats(None, ts, day, [xd, xi], [vd, None])
    assert r["n"] == e["n"] == 2
    for c in range(2):
        assert np.array_equal(r["count"][c].cpu().numpy(), e["count"][c])
        assert np.array_equal(r["min"][c].cpu().numpy(), e["min"][c]) and np.array_equal(r["max"][c].cpu().numpy(), e["max"][c])
        np.testing.assert_allclose(r["mean"][c].cpu().numpy(), e["mean"][c], rtol=1e-9)
        np.testing.assert_allclose(r["stddev"][c].cpu().numpy(), e["stddev"][c], rtol=1e-9)
    assert np.array_equal(r["sum"][1].cpu().numpy(), e["sum"][1])
    np.testing.assert_allclose(r["sum"][0].cpu().numpy(), e["sum"][0], rtol=1e-9)


def test_lookback_very_long_history(orc, ops):
    """L * F beyond the blocked writer's range (16384 elements per row): the general writer."""
    rng = np.random.default_rng(77)
    n, L = 3000, 5600
    key = np.zeros(n, np.int32)
    ts = BASE + np.arange(n) * 1_000_000
    xs = [rng.normal(0, 1, n), rng.normal(5, 2, n), rng.normal(1, 1, n)]
    xv = [None, (rng.random(n) >= 0.2).astype(np.uint8), None]
    rows, lens, feat, fv = ops.lookback(_dev(key), _dev(ts), [_dev(x) for x in xs], [_dev(v) for v in xv], 1, L,
                                        exact_size=False)
    operm, olen, ofeat, ofv = orc.lookback(key, ts, xs, xv, L)
    assert np.array_equal(rows.cpu().numpy().astype(np.int64), operm)
    assert np.array_equal(lens.cpu().numpy(), olen)
    assert np.array_equal(feat.cpu().numpy(), ofeat) and np.array_equal(fv.cpu().numpy(), ofv)


@pytest.mark.parametrize("feed", ["time_ordered", "sorted", "random_few_cells", "random_many_cells", "many_keys",
                                  "single_series"])
def test_resample_mean_without_ordering(orc, ops, feed, monkeypatch):
    """mean over a table of (key, bucket) cells, rows added where they lie: time-ordered and sorted
    feeds, a random one whose cells fit a block's table, many keys (the block tables fill up and
    are flushed), a table without keys; a random feed over many cells takes the ordered path.
    NULLs, an integer column, timestamps before the epoch."""
    monkeypatch.setenv("TEMPO_DENSE_MIN_ROWS", "1")
    rng = np.random.default_rng(91)
    n = 400_000
    nk = {"many_keys": 700, "random_few_cells": 4, "single_series": 1}.get(feed, 9)
    dur = 60_000_000
    span = {"random_few_cells": 100 * dur, "many_keys": 300 * dur}.get(feed, 2000 * dur)
    ts = -3 * dur + np.sort(rng.integers(0, span, n))          # starts before the epoch
    key = rng.integers(0, nk, n).astype(np.int32)
    if feed == "sorted":
        o = np.lexsort((ts, key)); key, ts = key[o], ts[o]
    if feed.startswith("random"):
        o = rng.permutation(n); key, ts = key[o], ts[o]
    a = rng.normal(-1.7, 3.86, n)
    b = rng.integers(-10**9, 10**9, n)
    c = 1e6 + rng.normal(0, 1e-3, n)
    av = (rng.random(n) >= 0.3).astype(np.uint8)
    cols, valid = [a, b, c], [av, None, None]
    k_dev = None if feed == "single_series" else _dev(key)
    r = ops.resample(k_dev, _dev(ts), nk, dur, "mean", [_dev(x) for x in cols], [_dev(v) for v in valid])
    e = orc.resample(None if feed == "single_series" else key, ts, dur, "mean", cols, valid)
    assert r["n"] == e["n"]
    assert np.array_equal(r["key"].cpu().numpy(), e["key"])
    assert np.array_equal(r["start"].cpu().numpy(), e["start"])
    assert np.array_equal(r["valid"].cpu().numpy(), e["valid"])
    np.testing.assert_allclose(r["f64"].cpu().numpy(), e["f64"], rtol=1e-9, atol=1e-12)
    assert not r["i64"].cpu().numpy().any()


@pytest.mark.parametrize("feed", ["time_ordered", "sorted", "random_few_cells", "random_many_cells", "many_keys",
                                  "single_series"])
@pytest.mark.parametrize("dur", [60_000_000, 86_400_000_000])
def test_grouped_stats_without_ordering(orc, ops, feed, dur, monkeypatch):
    """withGroupedStats over a table of (key, bucket) cells, two sweeps over the rows as they lie:
    counts, minima, maxima and integer sums bit-exact, mean / double sum / stddev within 1e-9 --
    NULLs, NaNs (greatest under Spark's ordering), an integer column, timestamps before the
    epoch; a random feed over many cells takes the ordered path."""
    monkeypatch.setenv("TEMPO_DENSE_MIN_ROWS", "1")
    rng = np.random.default_rng(97)
    n = 300_000
    nk = {"many_keys": 700, "random_few_cells": 4, "single_series": 1}.get(feed, 9)
    minute = 60_000_000
    span = {"random_few_cells": 100 * minute, "many_keys": 300 * minute}.get(feed, 2000 * minute)
    ts = -3 * minute + np.sort(rng.integers(0, span, n))
    key = rng.integers(0, nk, n).astype(np.int32)
    if feed == "sorted":
        o = np.lexsort((ts, key)); key, ts = key[o], ts[o]
    if feed.startswith("random"):
        o = rng.permutation(n); key, ts = key[o], ts[o]
    xd = 100 + np.cumsum(rng.normal(0, 0.01, n))
    xd[rng.random(n) < 0.0005] = np.nan
    xd[rng.random(n) < 0.0005] = -np.inf
    xi = rng.integers(-10**12, 10**12, n)
    vd = (rng.random(n) > 0.05).astype(np.uint8)
    vi = (rng.random(n) > 0.5).astype(np.uint8)
    k_dev = None if feed == "single_series" else _dev(key)
    r = ops.grouped_stats(k_dev, _dev(ts), nk, dur, [_dev(xd), _dev(xi)], [_dev(vd), _dev(vi)])
    e = orc.grouped_stats(None if feed == "single_series" else key, ts, dur, [xd, xi], [vd, vi])
    assert r["n"] == e["n"]
    assert np.array_equal(r["key"].cpu().numpy(), e["key"]) and np.array_equal(r["start"].cpu().numpy(), e["start"])
    for c in range(2):
        cnt = r["count"][c].cpu().numpy()
        assert np.array_equal(cnt, e["count"][c]), c
        ok, ok2 = cnt > 0, cnt > 1
        for name in ("min", "max"):
            assert np.array_equal(r[name][c].cpu().numpy()[ok], e[name][c][ok], equal_nan=True), (c, name)
        if e["is_i64"][c]:
            assert np.array_equal(r["sum"][c].cpu().numpy()[ok], e["sum"][c][ok])
        else:
            np.testing.assert_allclose(r["sum"][c].cpu().numpy()[ok], e["sum"][c][ok], rtol=1e-9, equal_nan=True)
        np.testing.assert_allclose(r["mean"][c].cpu().numpy()[ok], e["mean"][c][ok], rtol=1e-9, equal_nan=True)
        scale = 1e12 if e["is_i64"][c] else 100.0
        np.testing.assert_allclose(r["stddev"][c].cpu().numpy()[ok2], e["stddev"][c][ok2], rtol=1e-9,
                                   atol=1e-12 * scale, equal_nan=True)


@pytest.mark.parametrize("func", ["min", "max"])
@pytest.mark.parametrize("feed", ["time_ordered", "sorted", "random_few_cells", "random_many_cells", "many_keys"])
def test_resample_extremes_without_ordering(orc, ops, func, feed, monkeypatch):
    """resample(min / max) of one or two columns from the table of (key, bucket) cells: extremes
    as integer atomics on order-preserving codes, bit-exact against the oracle (doubles with NULLs
    and infinities, an integer column)."""
    monkeypatch.setenv("TEMPO_DENSE_MIN_ROWS", "1")
    rng = np.random.default_rng(101)
    n = 300_000
    nk = {"many_keys": 700, "random_few_cells": 4}.get(feed, 9)
    dur = 60_000_000
    span = {"random_few_cells": 100 * dur, "many_keys": 300 * dur}.get(feed, 2000 * dur)
    ts = -3 * dur + np.sort(rng.integers(0, span, n))
    key = rng.integers(0, nk, n).astype(np.int32)
    if feed == "sorted":
        o = np.lexsort((ts, key)); key, ts = key[o], ts[o]
    if feed.startswith("random"):
        o = rng.permutation(n); key, ts = key[o], ts[o]
    a = np.round(rng.normal(-1.7, 3.86, n), 2)
    a[rng.random(n) < 0.001] = np.inf
    a[rng.random(n) < 0.001] = -np.inf
    b = rng.integers(-10**15, 10**15, n)
    av = (rng.random(n) >= 0.3).astype(np.uint8)
    bv = (rng.random(n) >= 0.9).astype(np.uint8)      # mostly NULL: some buckets have no value at all
    r = ops.resample(_dev(key), _dev(ts), nk, dur, func, [_dev(a), _dev(b)], [_dev(av), _dev(bv)])
    e = orc.resample(key, ts, dur, func, [a, b], [av, bv])
    assert r["n"] == e["n"]
    assert np.array_equal(r["key"].cpu().numpy(), e["key"]) and np.array_equal(r["start"].cpu().numpy(), e["start"])
    assert np.array_equal(r["valid"].cpu().numpy(), e["valid"])
    assert np.array_equal(r["f64"].cpu().numpy(), e["f64"])
    assert np.array_equal(r["i64"].cpu().numpy(), e["i64"])


@pytest.mark.parametrize("func", ["floor", "ceil"])
@pytest.mark.parametrize("feed", ["time_ordered", "sorted", "random_few_cells", "many_keys", "ties", "no_columns"])
def test_resample_pick_without_ordering(orc, ops, func, feed, monkeypatch):
    """floor / ceil from the table of (key, bucket) cells: the row with the cell's earliest /
    latest timestamp.  Rows that tie on it are decided by their columns, which sends the call the
    ordered way ("ties"); without columns the earliest of them in the table wins either way."""
    monkeypatch.setenv("TEMPO_DENSE_MIN_ROWS", "1")
    rng = np.random.default_rng(103)
    n = 200_000
    nk = {"many_keys": 700, "random_few_cells": 4}.get(feed, 9)
    dur = 60_000_000
    span = {"random_few_cells": 100 * dur, "many_keys": 300 * dur}.get(feed, 2000 * dur)
    ts = -3 * dur + np.sort(rng.integers(0, span, n))
    if feed in ("ties", "no_columns"):
        ts = (ts // 1_000_000) * 1_000_000            # whole seconds: many rows share a timestamp
    key = rng.integers(0, nk, n).astype(np.int32)
    if feed == "sorted":
        o = np.lexsort((ts, key)); key, ts = key[o], ts[o]
    if feed.startswith("random"):
        o = rng.permutation(n); key, ts = key[o], ts[o]
    a = np.round(rng.normal(0, 2, n), 1)
    b = rng.integers(-5, 6, n)
    av = (rng.random(n) >= 0.2).astype(np.uint8)
    cols, valid = ([], []) if feed == "no_columns" else ([a, b], [av, None])
    r = ops.resample(_dev(key), _dev(ts), nk, dur, func, [_dev(x) for x in cols], [_dev(v) for v in valid])
    e = orc.resample(key, ts, dur, func, cols, valid)
    assert r["n"] == e["n"]
    assert np.array_equal(r["key"].cpu().numpy(), e["key"]) and np.array_equal(r["start"].cpu().numpy(), e["start"])
    assert np.array_equal(r["row"].cpu().numpy().astype(np.int64), e["row"])


def test_grouped_stats_big_buckets_whose_tiles_open_with_nulls(orc, ops, monkeypatch):
    """Buckets of several hundred thousand rows are reduced tile by tile; the sums of a tile are
    taken around a value of the tile even when its first rows are NULL.  A constant column has a
    standard deviation of exactly 0 (as Spark's recurrence gives), and values far from 0 with a
    small spread keep nine digits -- checked against an extended-precision evaluation."""
    monkeypatch.setenv("TEMPO_NO_DENSE_STATS", "1")   # the ordered path's big-bucket kernels
    rng = np.random.default_rng(8)
    n = 400_000
    ts = 1_596_240_000_000_000 + np.sort(rng.integers(0, 20 * 3600 * 1_000_000, n))   # one day, one key
    const = np.full(n, 0.7306188)
    level = 1e6 + np.cumsum(rng.normal(0, 1e-3, n))
    v = (rng.random(n) >= 0.3).astype(np.uint8)
    v[::512] = 0            # every tile of the ordered rows opens with a NULL, whatever its size
    v[100_000:130_000] = 0  # and some with nothing but NULLs
    r = ops.grouped_stats(None, _dev(ts), 1, 86_400_000_000, [_dev(const), _dev(level)], [_dev(v), _dev(v)])
    e = orc.grouped_stats(np.zeros(n, np.int32), ts, 86_400_000_000, [const, level], [v, v])
    assert r["n"] == e["n"] == 1 and int(r["count"][0][0]) == int(v.sum())
    assert float(r["stddev"][0][0]) == 0.0 and float(e["stddev"][0][0]) == 0.0
    vals = level[v != 0].astype(np.longdouble)
    exact = float(np.sqrt(((vals - vals.mean()) ** 2).sum() / (len(vals) - 1)))
    assert abs(float(r["stddev"][1][0]) - exact) <= 1e-9 * exact, (float(r["stddev"][1][0]), exact)
    assert abs(float(r["mean"][1][0]) - float(vals.mean())) <= 1e-12 * 1e6
    assert float(r["min"][1][0]) == float(vals.min()) and float(r["max"][1][0]) == float(vals.max())
