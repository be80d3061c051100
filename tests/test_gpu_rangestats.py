"""HIP withRangeStats against the CPU oracle: counts/min/max/NULL flags exact, fp64 statistics
within 1e-9 relative (zscore: 1e-9 * max(1, |z|)), through the C ABI."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu

BASE = 1_596_240_000_000_000
RTOL = 1e-9


def _dev(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope="module")
def ops():
    from tempo_amd import ops
    return ops


def _compare(orc, ops, key, ts, x, xv, nkeys, W, ts_is_seconds=False):
    out = ops.range_stats(_dev(key), _dev(ts), _dev(x), _dev(xv), nkeys, W, ts_is_seconds=ts_is_seconds)
    torch.cuda.synchronize()
    ref = orc.range_stats(key, ts, x, xv, W, ts_is_seconds=ts_is_seconds)
    g = {k: v.cpu().numpy() for k, v in out.items()}
    assert np.array_equal(g["perm"].astype(np.int64), ref["perm"])
    assert np.array_equal(g["count"], ref["count"])
    sv, dv, zv = (g["valid"] & 1) != 0, (g["valid"] & 2) != 0, (g["valid"] & 4) != 0
    assert np.array_equal(sv, ref["stat_valid"].astype(bool))
    assert np.array_equal(dv, ref["stddev_valid"].astype(bool))
    assert np.array_equal(zv, ref["zscore_valid"].astype(bool))
    assert np.array_equal(g["min"][sv], ref["min"][sv], equal_nan=True)
    assert np.array_equal(g["max"][sv], ref["max"][sv], equal_nan=True)
    if np.issubdtype(x.dtype, np.integer):
        assert np.array_equal(g["sum"][sv], ref["sum"][sv])
    else:
        np.testing.assert_allclose(g["sum"][sv], ref["sum"][sv], rtol=RTOL, atol=0, equal_nan=True)
    np.testing.assert_allclose(g["mean"][sv], ref["mean"][sv], rtol=RTOL, atol=0, equal_nan=True)
    # stddev: 1e-9 relative, with an absolute floor at 1e-9 of the value scale for frames whose
    # spread is far below the values themselves (both algorithms lose those digits)
    scale = np.nanmax(np.abs(np.where(np.isfinite(ref["mean"]), ref["mean"], 0))) if sv.any() else 1.0
    np.testing.assert_allclose(g["stddev"][dv], ref["stddev"][dv], rtol=RTOL, atol=1e-12 * max(scale, 1.0),
                               equal_nan=True)
    zt = RTOL * np.maximum(1.0, np.abs(ref["zscore"][zv]))
    # (x - mean) / stddev: a mean that is right to an ulp is still off by ulp(mean) / stddev in the
    # quotient -- the oracle's as much as the device's -- which shows where a frame's spread is tiny
    with np.errstate(divide="ignore", invalid="ignore"):
        cond = 8 * np.finfo(np.float64).eps * np.abs(ref["mean"][zv]) / ref["stddev"][zv]
    zt = zt + np.where(np.isfinite(cond), cond, 0.0) / 10
    dz = np.abs(g["zscore"][zv] - ref["zscore"][zv])
    ok = (dz <= zt * 10) | (np.isnan(g["zscore"][zv]) & np.isnan(ref["zscore"][zv]))
    assert ok.all(), f"zscore off: max {np.nanmax(dz)}"
    return g, ref


def _ticks(rng, n, nkeys, mean_gap_s, null_p, sort=False):
    key = rng.integers(0, nkeys, n).astype(np.int32)
    ts = BASE + (rng.exponential(mean_gap_s * nkeys, n).cumsum() * 1e6).astype(np.int64)
    if not sort:
        p = rng.permutation(n)
        key, ts = key[p], ts[p]
    x = 100 + rng.normal(0, 1, n).cumsum() * 0.01
    xv = (rng.random(n) >= null_p).astype(np.uint8) if null_p else None
    return key, ts, x, xv


@pytest.mark.parametrize("seed", range(4))
def test_range_stats_small(orc, ops, seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 5000))
    key, ts, x, xv = _ticks(rng, n, 4, 1.0, 0.1)
    _compare(orc, ops, key, ts, x, xv, 4, int(rng.integers(0, 30)))


def test_range_stats_window_600_ticks(orc, ops):
    rng = np.random.default_rng(10)
    key, ts, x, xv = _ticks(rng, 60_000, 10, 1.0, 0.005)
    _compare(orc, ops, key, ts, x, xv, 10, 600)


def test_range_stats_presorted_no_nulls(orc, ops):
    rng = np.random.default_rng(11)
    key, ts, x, _ = _ticks(rng, 30_000, 5, 2.0, 0.0)
    o = np.lexsort((ts, key))
    _compare(orc, ops, key[o], ts[o], x[o], None, 5, 120)


def test_range_stats_dense_frames_take_sequential_path(orc, ops):
    """Frames of ~5000 rows exceed the LDS extent: the exact sequential path must agree too."""
    rng = np.random.default_rng(12)
    n = 20_000
    key = np.zeros(n, np.int32)
    ts = BASE + np.sort(rng.integers(0, 40_000_000, n))  # 40 s, 500 rows per second
    x = rng.normal(50, 3, n)
    _compare(orc, ops, key, ts, x, None, 1, 10)


def test_range_stats_identical_values_and_specials(orc, ops):
    rng = np.random.default_rng(13)
    n = 6000
    key = rng.integers(0, 3, n).astype(np.int32)
    ts = BASE + rng.integers(0, 3000, n) * 1_000_000
    x = np.round(rng.normal(10, 0.3, n), 0)  # many frames hold one repeated value -> stddev 0, zscore NULL
    x[rng.integers(0, n, 5)] = np.nan
    x[rng.integers(0, n, 3)] = np.inf
    x[rng.integers(0, n, 3)] = -np.inf
    xv = (rng.random(n) >= 0.3).astype(np.uint8)
    _compare(orc, ops, key, ts, x, xv, 3, 4)


def test_range_stats_int64_metric_and_second_ts(orc, ops):
    rng = np.random.default_rng(14)
    n = 20_000
    key = rng.integers(0, 7, n).astype(np.int32)
    ts = rng.integers(1_000_000, 1_020_000, n)  # integral ts column: seconds already
    x = rng.integers(-(2 ** 40), 2 ** 40, n)
    xv = (rng.random(n) >= 0.1).astype(np.uint8)
    _compare(orc, ops, key, ts, x, xv, 7, 50, ts_is_seconds=True)


def test_range_stats_golden(ops, golden):
    from conftest import ts_us
    fx = golden["range_stats"]
    ts = np.array([ts_us(r[1]) for r in fx["data"]])
    x = np.array([r[2] for r in fx["data"]], np.float32).astype(np.float64)
    out = ops.range_stats(_dev(np.zeros(4, np.int32)), _dev(ts), _dev(x), None, 1, fx["rangeBackWindowSecs"])
    g = {k: v.cpu().numpy() for k, v in out.items()}
    for p, erow in enumerate(fx["expected"]):
        e = dict(zip(fx["expected_columns"], erow))
        assert g["count"][p] == e["count_trade_pr"]
        for name in ("mean", "min", "max", "sum"):
            assert abs(g[name][p] - e[name + "_trade_pr"]) < 0.00501
        if e["stddev_trade_pr"] is None:
            assert not g["valid"][p] & 2 and not g["valid"][p] & 4
        else:
            assert abs(g["stddev"][p] - e["stddev_trade_pr"]) < 0.00501
            assert abs(g["zscore"][p] - e["zscore_trade_pr"]) < 0.00501


def test_range_stats_time_ordered_feed(orc, ops):
    """Rows in time order with interleaved keys (a live feed): only the key bits need sorting."""
    rng = np.random.default_rng(15)
    key, ts, x, xv = _ticks(rng, 80_000, 12, 1.0, 0.01, sort=True)  # sort=True keeps arrival order
    assert (np.diff(ts) >= 0).all() and (np.diff(key) < 0).any()
    _compare(orc, ops, key, ts, x, xv, 12, 600)


@pytest.mark.parametrize("kind", ["f64", "i64", "specials"])
def test_range_stats_wide_frames_vs_oracle(orc, ops, kind):
    """Frames of a few thousand rows (far beyond a block's LDS region), several keys, random row
    order, NULLs: completed by the chunk-prefix path, same bar as the region kernel."""
    rng = np.random.default_rng(77)
    n, nk = 60_000, 3
    key = rng.integers(0, nk, n).astype(np.int32)
    ts = BASE + rng.integers(0, 600_000_000, n)          # 10 minutes: 33 rows per key and second
    xv = (rng.random(n) >= 0.02).astype(np.uint8)
    if kind == "i64":
        x = rng.integers(-10**15, 10**15, n)
    else:
        x = 1000 + rng.normal(0, 1, n).cumsum() * 0.05
        if kind == "specials":
            x[rng.integers(0, n, 12)] = np.nan
            x[rng.integers(0, n, 6)] = np.inf
            x[rng.integers(0, n, 6)] = -np.inf
    _compare(orc, ops, key, ts, x, xv, nk, 120)          # ~4000-row frames


def _numpy_reference(sec, x, W):
    """Exact-enough statistics of RANGE BETWEEN W PRECEDING AND CURRENT ROW on one ordered
    series without NULLs: extended-precision prefix sums of centred data, sparse tables."""
    n = len(x)
    A = np.searchsorted(sec, sec - W, side="left")
    B = np.searchsorted(sec, sec, side="right") - 1
    c = x.astype(np.longdouble) - np.longdouble(x.mean())
    P = np.concatenate([[0], np.cumsum(c)])
    Q = np.concatenate([[0], np.cumsum(c * c)])
    cnt = (B - A + 1).astype(np.int64)
    S, SS = P[B + 1] - P[A], Q[B + 1] - Q[A]
    mean = (S / cnt + np.longdouble(x.mean())).astype(np.float64)
    var = (SS - S * S / cnt) / np.maximum(cnt - 1, 1)
    sd = np.sqrt(np.maximum(var, 0)).astype(np.float64)
    lv = np.floor(np.log2(cnt)).astype(np.int64)
    tmin, tmax = [x.copy()], [x.copy()]
    for l in range(1, int(lv.max()) + 1):
        h = 1 << (l - 1)
        tmin.append(np.minimum(tmin[-1][:n - 2 * h + 1], tmin[-1][h:n - h + 1]))
        tmax.append(np.maximum(tmax[-1][:n - 2 * h + 1], tmax[-1][h:n - h + 1]))
    mn, mx = np.empty(n), np.empty(n)
    for l in range(int(lv.max()) + 1):
        m = lv == l
        mn[m] = np.minimum(tmin[l][A[m]], tmin[l][B[m] + 1 - (1 << l)])
        mx[m] = np.maximum(tmax[l][A[m]], tmax[l][B[m] + 1 - (1 << l)])
    return cnt, mean, sd, mn, mx


@pytest.mark.parametrize("trend", [0.0, 0.01])
def test_range_stats_frames_across_superblocks(ops, trend):
    """Frames of ~100 000 rows span several 65 536-row superblocks: pieces with different pivots
    are merged pairwise.  Checked against an extended-precision numpy evaluation."""
    rng = np.random.default_rng(5)
    n = 300_000
    ts = BASE + np.sort(rng.integers(0, 3_000_000_000, n))   # 3000 s, 100 rows per second
    x = 500 + rng.normal(0, 1, n).cumsum() * 0.02 + trend * np.arange(n)
    from tempo_amd import _lib
    _lib.profile_enable(True)
    _lib.profile_reset()
    out = ops.range_stats(None, _dev(ts), _dev(x), None, 1, 1000)
    torch.cuda.synchronize()
    launched = {name for name, _, _ in _lib.profile_read()}
    _lib.profile_reset()
    short = ops.range_stats(None, _dev(ts), _dev(x), None, 1, 5)       # 500-row frames: region kernel only
    torch.cuda.synchronize()
    launched_short = {name for name, _, _ in _lib.profile_read()}
    _lib.profile_enable(False)
    assert any("wide_rows_kernel" in k for k in launched), launched
    assert not any("wide_" in k for k in launched_short), launched_short
    del short
    g = {k: v.cpu().numpy() for k, v in out.items()}
    assert np.array_equal(g["perm"], np.arange(n))
    cnt, mean, sd, mn, mx = _numpy_reference(ts // 1_000_000, x, 1000)
    assert np.array_equal(g["count"], cnt) and (g["valid"] & 1).all()
    assert np.array_equal(g["min"], mn) and np.array_equal(g["max"], mx)
    np.testing.assert_allclose(g["mean"], mean, rtol=RTOL)
    m = cnt > 1
    np.testing.assert_allclose(g["stddev"][m], sd[m], rtol=RTOL, atol=1e-12 * 500)
    np.testing.assert_allclose(g["sum"], mean * cnt, rtol=1e-9)


def _by_row(d, perm, names):
    inv = np.empty(len(perm), np.int64)
    inv[np.asarray(perm, np.int64)] = np.arange(len(perm))
    return {k: np.asarray(d[k])[inv] for k in names}


@pytest.mark.parametrize("case", ["ticks", "sparse_keys", "seconds_ts", "w0", "long_window", "single_series",
                                  "with_nan", "random_order"])
def test_range_stats_any_order_over_second_cells(orc, ops, case, monkeypatch):
    """any_order: a time-ordered feed is aggregated per (key, second) cell and every row's frame
    looked up as the union of at most two runs of cells; the rows come back where they lie.
    Compared row by row with the oracle: counts / min / max / NULL flags exact, fp64 statistics to
    1e-9.  NULLs, a window of 0 s, a window longer than the table, a table without keys,
    second-valued timestamps; NaNs and a feed in random order fall back to the ordered path."""
    monkeypatch.setenv("TEMPO_DENSE_MIN_ROWS", "1")
    rng = np.random.default_rng(131)
    n, nk, W = 250_000, 37, 600
    gap = 0.05                                      # ~20 rows per second over all keys
    if case == "long_window":
        n = 60_000                                  # every frame will reach back to the key's first row
    if case == "sparse_keys":
        nk = 700                                    # ~35 seconds between two rows of a key
    if case == "single_series":
        nk = 1
    key = rng.integers(0, nk, n).astype(np.int32)
    ts = BASE + (rng.exponential(gap, n).cumsum() * 1e6).astype(np.int64)
    secs = case == "seconds_ts"
    if secs:
        ts = ts // 1_000_000
    if case == "w0":
        W = 0
    if case == "long_window":
        W = 10**7
    if case == "random_order":
        p = rng.permutation(n); key, ts = key[p], ts[p]
    x = 100 + rng.normal(0, 1, n).cumsum() * 0.01
    if case == "with_nan":
        x[rng.integers(0, n, 20)] = np.nan
    xv = (rng.random(n) >= 0.1).astype(np.uint8)
    k_np = None if case == "single_series" else key
    out = ops.range_stats(_dev(k_np), _dev(ts), _dev(x), _dev(xv), nk, W, ts_is_seconds=secs, any_order=True)
    torch.cuda.synchronize()
    g = {k: v.cpu().numpy() for k, v in out.items()}
    perm = g["perm"].astype(np.int64)
    assert np.array_equal(np.sort(perm), np.arange(n))
    if case not in ("with_nan", "random_order"):
        assert np.array_equal(perm, np.arange(n)), "the cells path leaves the rows where they lie"
    ref = orc.range_stats(k_np, ts, x, xv, W, ts_is_seconds=secs)
    names = ["count", "mean", "min", "max", "sum", "stddev", "zscore"]
    G = _by_row(g, perm, names + ["valid"])
    R = _by_row(ref, ref["perm"], names + ["stat_valid", "stddev_valid", "zscore_valid"])
    assert np.array_equal(G["count"], R["count"])
    sv, dv, zv = (G["valid"] & 1) != 0, (G["valid"] & 2) != 0, (G["valid"] & 4) != 0
    assert np.array_equal(sv, R["stat_valid"].astype(bool))
    assert np.array_equal(dv, R["stddev_valid"].astype(bool))
    assert np.array_equal(zv, R["zscore_valid"].astype(bool))
    assert np.array_equal(G["min"][sv], R["min"][sv], equal_nan=True)
    assert np.array_equal(G["max"][sv], R["max"][sv], equal_nan=True)
    np.testing.assert_allclose(G["sum"][sv], R["sum"][sv], rtol=RTOL, atol=0, equal_nan=True)
    np.testing.assert_allclose(G["mean"][sv], R["mean"][sv], rtol=RTOL, atol=0, equal_nan=True)
    np.testing.assert_allclose(G["stddev"][dv], R["stddev"][dv], rtol=RTOL, atol=1e-10, equal_nan=True)
    zt = RTOL * np.maximum(1.0, np.abs(R["zscore"][zv]))
    dz = np.abs(G["zscore"][zv] - R["zscore"][zv])
    ok = (dz <= zt * 10) | (np.isnan(G["zscore"][zv]) & np.isnan(R["zscore"][zv]))
    assert ok.all(), f"zscore off: max {np.nanmax(dz)}"


def test_profile_slots_from_several_host_threads(ops):
    """Launch slots of the per-kernel event timing come from an atomic counter: four host threads
    running the operator on streams of their own each get their launches recorded once, the
    results are those of a single-threaded call, and nothing is reported as dropped."""
    import threading
    from tempo_amd import _lib
    rng = np.random.default_rng(11)
    n = 50_000
    key = rng.integers(0, 8, n).astype(np.int32)
    ts = BASE + np.sort(rng.integers(0, 600_000_000, n))
    x = rng.normal(0, 1, n)
    dk, dt, dx = _dev(key), _dev(ts), _dev(x)
    _lib.profile_enable(True)
    _lib.profile_reset()
    want = ops.range_stats(dk, dt, dx, None, 8, 30)
    torch.cuda.synchronize()
    per_call = len(_lib.profile_read())
    assert per_call > 0
    _lib.profile_reset()
    got, errs = [None] * 4, []

    def work(i):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                for _ in range(5):
                    got[i] = ops.range_stats(dk, dt, dx, None, 8, 30)
                torch.cuda.current_stream().synchronize()
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    recs = _lib.profile_read()
    dropped = _lib.profile_dropped()
    _lib.profile_enable(False)
    _lib.profile_reset()
    assert not errs, errs
    assert len(recs) == 4 * 5 * per_call and dropped == 0
    assert all(ms >= 0.0 for _, _, ms in recs)
    for g in got:
        for name in ("perm", "count", "min", "max", "mean", "stddev"):
            assert torch.equal(g[name], want[name]), name


def test_range_stats_long_frames_after_a_long_run_of_nulls(orc, ops):
    """Frames of thousands of rows (the chunk-prefix path) over values far from 0 with a tiny
    spread, where a run of NULLs longer than a chunk opens a 65 536-row superblock: the pivot of
    the sums must still be a value of the data.  Standard deviation against an extended-precision
    evaluation of a few frames, everything else against the oracle."""
    rng = np.random.default_rng(31)
    n = 200_000
    ts = BASE + np.sort(rng.integers(0, 1_000_000_000, n))     # 1000 s, 200 rows per second
    x = 1e6 + np.cumsum(rng.normal(0, 1e-3, n))
    xv = np.ones(n, np.uint8)
    xv[65_536 - 300:65_536 + 4000] = 0     # NULLs across the start of the second superblock
    xv[131_072:131_072 + 1500] = 0         # ... and exactly from the start of the third
    out = ops.range_stats(None, _dev(ts), _dev(x), _dev(xv), 1, 30)
    g = {k: v.cpu().numpy() for k, v in out.items()}
    ref = orc.range_stats(np.zeros(n, np.int32), ts, x, xv, 30)
    assert np.array_equal(g["perm"], ref["perm"]) and np.array_equal(g["count"], ref["count"])
    assert np.array_equal(g["min"], ref["min"]) and np.array_equal(g["max"], ref["max"])
    np.testing.assert_allclose(g["mean"], ref["mean"], rtol=1e-12)
    sec = ts // 1_000_000
    for i in (66_000, 70_000, 75_000, 131_500, 133_000, 140_000, n - 1):
        m = (sec >= sec[i] - 30) & (sec <= sec[i]) & (xv != 0)
        vals = x[m].astype(np.longdouble)
        exact = float(np.sqrt(((vals - vals.mean()) ** 2).sum() / (len(vals) - 1)))
        assert g["count"][i] == m.sum() > 1000
        assert abs(g["stddev"][i] - exact) <= 1e-9 * exact, (i, g["stddev"][i], exact)
    # the oracle's sequential recurrence (Spark's own) adds values of 1e6 to a running mean: with a
    # spread of 1e-2 it keeps about seven digits of the standard deviation
    np.testing.assert_allclose(g["stddev"], ref["stddev"], rtol=1e-6)


@pytest.mark.parametrize("order", ["random", "time", "sorted"])
@pytest.mark.parametrize("n_cols", [1, 3, 6])
def test_range_stats_several_columns_in_one_call(orc, ops, order, n_cols):
    """tmp_range_stats_cols: one ordering, every column travelling with the rows; column by column
    the same results as the single-column call (bit for bit) and the oracle's.  Mixed double and
    int64 columns, with and without NULLs; six columns go in two groups."""
    rng = np.random.default_rng(50 + n_cols)
    n, nk = 120_000, 9
    key = rng.integers(0, nk, n).astype(np.int32)
    ts = BASE + (rng.exponential(0.2, n).cumsum() * 1e6).astype(np.int64)
    if order == "random":
        p = rng.permutation(n); key, ts = key[p], ts[p]
    elif order == "sorted":
        p = np.lexsort((ts, key)); key, ts = key[p], ts[p]
    xs, xv = [], []
    for c in range(n_cols):
        if c % 3 == 1:
            xs.append(rng.integers(1, 2000, n))   # positive: a mean of exactly 0 has no relative tolerance
        else:
            xs.append(100.0 * (c + 1) + rng.normal(0, 1, n).cumsum() * 0.01)
        xv.append(None if c % 2 == 0 else (rng.random(n) >= 0.1 * c).astype(np.uint8))
    perm, outs = ops.range_stats_cols(_dev(key), _dev(ts), [_dev(x) for x in xs], [_dev(v) for v in xv], nk, 30)
    assert len(outs) == n_cols
    for c in range(n_cols):
        g, ref = _compare(orc, ops, key, ts, xs[c], xv[c], nk, 30)   # single-column call vs oracle
        assert np.array_equal(perm.cpu().numpy(), g["perm"])
        for name in ("mean", "count", "min", "max", "sum", "stddev", "zscore", "valid"):
            assert np.array_equal(outs[c][name].cpu().numpy(), g[name], equal_nan=True), (c, name)


def test_range_stats_several_columns_long_frames(orc, ops):
    """Frames longer than the on-chip region with several columns: the wide path completes each."""
    rng = np.random.default_rng(77)
    n, nk = 60_000, 2
    key = rng.integers(0, nk, n).astype(np.int32)
    ts = BASE + (rng.exponential(0.01, n).cumsum() * 1e6).astype(np.int64)   # ~100 rows per second
    xs = [50.0 + rng.normal(0, 1, n), rng.integers(0, 50, n)]
    xv = [(rng.random(n) >= 0.2).astype(np.uint8), None]
    perm, outs = ops.range_stats_cols(_dev(key), _dev(ts), [_dev(x) for x in xs], [_dev(v) for v in xv], nk, 120)
    for c in range(2):
        g, ref = _compare(orc, ops, key, ts, xs[c], xv[c], nk, 120)
        for name in ("mean", "count", "min", "max", "sum", "stddev", "zscore", "valid"):
            assert np.array_equal(outs[c][name].cpu().numpy(), g[name], equal_nan=True), (c, name)


@pytest.mark.parametrize("cap", [1536, 2048, 2560])
@pytest.mark.parametrize("window", [120, 600])
def test_range_stats_every_region_size(orc, ops, monkeypatch, cap, window):
    """The statistics kernel stages a region of 1536, 2048 or 2560 rows around its 1024 (picked from the
    expected rows per frame): each size gives the oracle's numbers, also where frames are longer than the
    size reaches back (those rows take the wide path) -- and so does the size the library picks itself."""
    rng = np.random.default_rng(40 + window)
    key, ts, x, xv = _ticks(rng, 150_000, 7, 1.0, 0.01)
    monkeypatch.setenv("TEMPO_RK_CAP", str(cap))
    forced, _ = _compare(orc, ops, key, ts, x, xv, 7, window)
    monkeypatch.delenv("TEMPO_RK_CAP")
    picked, _ = _compare(orc, ops, key, ts, x, xv, 7, window)
    for name in ("count", "min", "max", "valid"):
        assert np.array_equal(forced[name], picked[name], equal_nan=forced[name].dtype.kind == "f"), name


def test_range_stats_bunched_rows_get_the_large_region(orc, ops):
    """Rows spread thinly over a long period, except for a burst that puts 1000 rows into one frame: the
    estimate of rows per frame picks the smallest region, the burst's frames leave it, and the column is
    run once more with the largest region (which holds them) -- not sent down the wide path."""
    from tempo_amd import _lib
    rng = np.random.default_rng(77)
    n, nk = 120_000, 3
    key = rng.integers(0, nk, n).astype(np.int32)
    ts = BASE + np.sort(rng.integers(0, 400_000, n)) * 1_000_000          # ~0.1 rows per second and key
    burst = slice(50_000, 53_000)
    ts[burst] = ts[50_000] + np.sort(rng.integers(0, 300, 3_000)) * 1_000_000   # 1000 rows per key within 300 s
    ts = np.sort(ts)
    x = 50 + rng.normal(0, 1, n).cumsum() * 0.01
    _lib.profile_enable(True)
    _lib.profile_reset()
    _compare(orc, ops, key, ts, x, None, nk, 600)
    names = [name for name, _, _ in _lib.profile_read()]
    _lib.profile_enable(False)
    # (the feed is time-ordered and its words fit 32 bits: both runs are over the narrow words)
    assert sum(name.startswith("range_stats_kernel<T>") for name in names) == 2, names
    assert not any("wide" in name for name in names), names


@pytest.mark.parametrize("case", ["plain", "nulls", "nonfinite", "long_frames", "one_series", "int_metric"])
def test_range_stats_rows_in_order_need_no_words(orc, ops, monkeypatch, case):
    """A table that arrives in (key, ts) order is aggregated where it lies: the kernel builds the sort
    words from the key / time / NULL-flag columns, out_perm is the identity.  Frames with NaN / inf and
    frames longer than the region send the call back to laid-out words; either way the numbers are the
    oracle's and those of the call that lays the words out first."""
    from tempo_amd import _lib
    rng = np.random.default_rng(len(case))
    n, nk = 90_000, 6
    key, ts, x, xv = _ticks(rng, n, nk, 1.0, 0.05 if case != "plain" else 0.0, sort=True)
    o = np.lexsort((ts, key))
    key, ts, x = key[o], ts[o], x[o]
    xv = None if xv is None else xv[o]
    window = 300
    if case == "nonfinite":
        x[rng.integers(0, n, 40)] = rng.choice([np.nan, np.inf, -np.inf], 40)
    if case == "long_frames":
        window = 60000         # about 1650 rows per frame: beyond every region
    if case == "one_series":
        key = np.zeros(n, np.int32); nk = 1
        ts = np.sort(ts)
    if case == "int_metric":
        x = rng.integers(1, 1000, n)
    _lib.profile_enable(True)
    _lib.profile_reset()
    got, _ = _compare(orc, ops, key, ts, x, xv, nk, window)
    names = {name for name, _, _ in _lib.profile_read()}
    _lib.profile_enable(False)
    assert "range_stats_kernel<T>+words" in names, names
    assert ("make_comp_kernel" in names) == (case in ("nonfinite", "long_frames")), names
    assert np.array_equal(got["perm"], np.arange(n))
    monkeypatch.setenv("TEMPO_NO_STATS_IN_PLACE", "1")
    laid, _ = _compare(orc, ops, key, ts, x, xv, nk, window)
    for name in ("count", "min", "max", "valid", "perm"):
        assert np.array_equal(got[name], laid[name], equal_nan=got[name].dtype.kind == "f"), name


@pytest.mark.parametrize("case", ["plain", "nulls", "nonfinite", "long_frames", "int_metric", "wide_words"])
def test_range_stats_time_ordered_feed_takes_narrow_words(orc, ops, monkeypatch, case):
    """A time-ordered feed whose sort words fit 32 bits: the one key pass writes them as 32-bit words and
    the kernel reads them as such.  Frames with NaN / inf and frames longer than the region send the call
    through the rows once more with 64-bit words; words that do not fit never take the narrow form.
    Either way the numbers are the oracle's and those of the call with the narrow form switched off."""
    from tempo_amd import _lib
    rng = np.random.default_rng(100 + len(case))
    n, nk = (1_200_000 if case == "plain" else 300_000), 50
    key, ts, x, xv = _ticks(rng, n, nk, 1.0, 0.05 if case != "plain" else 0.0, sort=True)
    window = 300
    if case == "nonfinite":
        x[rng.integers(0, n, 40)] = rng.choice([np.nan, np.inf, -np.inf], 40)
    if case == "long_frames":
        window = 5_000_000     # a key ticks every 2500 s: about 2000 rows per frame, beyond every region
    if case == "int_metric":
        x = rng.integers(1, 1000, n)
    if case == "wide_words":
        ts = BASE + (ts - BASE) * 4000   # a span of 4.8e9 seconds: 33 bits of time alone
        window = 300 * 4000
    _lib.profile_enable(True)
    _lib.profile_reset()
    got, _ = _compare(orc, ops, key, ts, x, xv, nk, window)
    passes = [name for name, _, _ in _lib.profile_read() if name.startswith("radix_scatter_kernel+words")]
    _lib.profile_enable(False)
    # one key pass; a second one only where the narrow run had to be repeated with 64-bit words
    assert len(passes) == (2 if case in ("nonfinite", "long_frames") else 1), passes
    monkeypatch.setenv("TEMPO_NO_NARROW_WORDS", "1")
    _lib.profile_enable(True)
    _lib.profile_reset()
    wide, _ = _compare(orc, ops, key, ts, x, xv, nk, window)
    passes = [name for name, _, _ in _lib.profile_read() if name.startswith("radix_scatter_kernel+words")]
    _lib.profile_enable(False)
    assert len(passes) == 1, passes
    for name in got:
        assert np.array_equal(got[name], wide[name], equal_nan=got[name].dtype.kind == "f"), name


@pytest.mark.parametrize("in_order", [False, True], ids=["time_ordered", "key_ts_ordered"])
@pytest.mark.parametrize("window,stretch_rows", [(60, 5000), (3000, 40000)], ids=["region", "wide_path"])
def test_range_stats_nearly_constant_stretches(ops, window, stretch_rows, in_order):
    """A metric that sits on a level for a while (noise of 1e-7) and then steps by 0.37: the standard
    deviation of every frame that lies inside one stretch, against the frame's values added up in
    extended precision (1e-9).  Frames next to a step are the hard ones: the sums they are taken from
    have the far larger values of the next stretch as neighbours.  (Spark's own recurrence, and with
    it the oracle, is good to about 1e-7 on such frames: the comparison is with the exact value.)"""
    rng = np.random.default_rng(5)
    n, nk = 400_000, 8
    key = rng.integers(0, nk, n).astype(np.int32)
    ts = BASE + np.sort(rng.integers(0, 50_000, n)) * 1_000_000      # one row per second and key
    stretch = np.arange(n) // stretch_rows
    level = 100 + stretch * 0.37
    x = level + rng.normal(0, 1e-7, n)
    if in_order:   # the table laid out in (key, ts) order: the kernel builds its sort words from the columns
        o = np.lexsort((ts, key))
        key, ts, x, stretch, level = key[o], ts[o], x[o], stretch[o], level[o]
    out = ops.range_stats(_dev(key), _dev(ts), _dev(x), None, nk, window)
    g = {k: v.cpu().numpy() for k, v in out.items()}
    p = g["perm"].astype(np.int64)
    sec = (ts - BASE) // 1_000_000
    ks, ss, st = key[p], sec[p], stretch[p]
    comp = ks.astype(np.int64) * (1 << 40) + ss
    assert (np.diff(comp) >= 0).all()
    a = np.searchsorted(comp, comp - np.minimum(ss, window), side="left")
    b = np.searchsorted(comp, comp, side="right")
    assert np.array_equal(g["count"], b - a)
    y = x[p].astype(np.longdouble) - level[p].astype(np.longdouble)   # exact: the level is common to the frame
    c1 = np.concatenate([[0], np.cumsum(y)])
    c2 = np.concatenate([[0], np.cumsum(y * y)])
    cnt = (b - a).astype(np.longdouble)
    m2 = (c2[b] - c2[a]) - (c1[b] - c1[a]) ** 2 / cnt
    inside = (st[a] == st[b - 1]) & (cnt > 1)
    assert inside.sum() > n // 4
    sd = np.sqrt(np.where(inside, m2 / np.maximum(cnt - 1, 1), 1))
    err = np.where(inside, np.abs(g["stddev"].astype(np.longdouble) - sd) / sd, 0).astype(np.float64)
    assert err.max() < 1e-9, (float(err.max()), int((err > 1e-9).sum()), np.flatnonzero(err > 1e-9)[:8])
    mean = level[p] + ((c1[b] - c1[a]) / cnt).astype(np.float64)
    assert np.abs(g["mean"] - mean)[inside].max() < 1e-9 * 100


def test_range_stats_nearly_constant_stretches_over_cells(ops):
    """The same through the (key, second) cells that a time-ordered feed may be aggregated over when the
    caller takes the rows in any order: per-second aggregates merged pairwise, the means carried in two
    parts so that the difference of two cells' means is not rounded at the level of the values."""
    from tempo_amd import _lib
    rng = np.random.default_rng(9)
    n, nk, w = 1 << 22, 8, 60
    key = rng.integers(0, nk, n).astype(np.int32)
    ts = BASE + np.sort(rng.integers(0, (n // 400) * 1_000_000, n))     # 50 rows per second and key
    stretch = np.arange(n) // 400_000
    level = 100 + stretch * 0.37
    x = level + rng.normal(0, 1e-7, n)
    _lib.profile_enable(True)
    _lib.profile_reset()
    out = ops.range_stats(_dev(key), _dev(ts), _dev(x), None, nk, w, any_order=True)
    torch.cuda.synchronize()
    names = {name for name, _, _ in _lib.profile_read()}
    _lib.profile_enable(False)
    assert "cell_rows_kernel" in names, names
    g = {k: v.cpu().numpy() for k, v in out.items()}
    p = g["perm"].astype(np.int64)
    sec = (ts - BASE) // 1_000_000
    order = np.lexsort((sec, key))
    ks, ss, st = key[order], sec[order], stretch[order]
    comp = ks.astype(np.int64) * (1 << 40) + ss
    a = np.searchsorted(comp, comp - np.minimum(ss, w), side="left")
    b = np.searchsorted(comp, comp, side="right")
    y = x[order].astype(np.longdouble) - level[order].astype(np.longdouble)
    c1 = np.concatenate([[0], np.cumsum(y)])
    c2 = np.concatenate([[0], np.cumsum(y * y)])
    cnt = (b - a).astype(np.longdouble)
    m2 = (c2[b] - c2[a]) - (c1[b] - c1[a]) ** 2 / cnt
    inside = (st[a] == st[b - 1]) & (cnt > 1)
    sd = np.sqrt(np.where(inside, m2 / np.maximum(cnt - 1, 1), 1)).astype(np.float64)
    inv = np.empty(n, np.int64)
    inv[p] = np.arange(n)
    assert np.array_equal(g["count"][inv[order]], b - a)
    err = np.where(inside, np.abs(g["stddev"][inv[order]] - sd) / sd, 0)
    assert inside.sum() > n // 2 and err.max() < 1e-9, float(err.max())


@pytest.mark.parametrize("window", [30, 5000])
def test_range_stats_values_whose_squares_overflow(orc, ops, window):
    """Values of 1e200: Spark's recurrence overflows to an infinite M2 and reports an infinite standard
    deviation for every frame that holds one; such frames are evaluated by that recurrence here too
    (the short frames by the region kernel's serial path, the long ones by the wide path's)."""
    rng = np.random.default_rng(window)
    n, nk = 60_000, 3
    key, ts, x, xv = _ticks(rng, n, nk, 1.0, 0.02)
    x[rng.integers(0, n, 25)] = rng.choice([1e200, -1e200, 3e160], 25)
    got, ref = _compare(orc, ops, key, ts, x, xv, nk, window)
    assert np.isinf(ref["stddev"]).any() and np.array_equal(np.isinf(got["stddev"]), np.isinf(ref["stddev"]))
