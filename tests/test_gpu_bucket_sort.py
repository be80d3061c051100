"""The bucketed form of the ordering step (partition passes over a bucket number, in-LDS tile
sort, plain passes as device-side stand-by): its order must be the oracle's stable order
(python/tempo/tsdf.py:117-121, :563-572) whether the tiles absorb every bucket or not."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

BASE = 1_596_240_000_000_000


def _kernels_of(fn):
    from tempo_amd import _lib
    _lib.profile_enable(True)
    _lib.profile_reset()
    out = fn()
    torch.cuda.synchronize()
    names = {}
    for name, _, ms in _lib.profile_read():
        names[name] = names.get(name, 0.0) + ms
    _lib.profile_enable(False)
    return out, names


def _check(orc, key, ts, n_keys, **kw):
    from tempo_amd import ops
    d = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()
    perm, names = _kernels_of(lambda: ops.sort_order(d(key), d(ts), n_keys, seq=d(kw.get("seq")),
                                                      seq_valid=d(kw.get("seq_valid")),
                                                      floor_seconds=kw.get("floor_seconds", False)))
    if kw.get("floor_seconds"):   # the oracle's window order: (key, floor(ts / 1e6), seq NULLS FIRST), stable
        want, _ = orc.window_order(key, ts, seq=kw.get("seq"), seq_valid=kw.get("seq_valid"))
    else:
        want = np.lexsort((np.arange(len(ts)), ts, key))
    assert np.array_equal(perm.cpu().numpy().astype(np.int64), np.asarray(want, dtype=np.int64))
    return names


def test_bucketed_uniform_tables_use_the_tile_sort(orc, monkeypatch):
    monkeypatch.setenv("TEMPO_BUCKET_MIN_ROWS", "3000")
    monkeypatch.delenv("TEMPO_PLAIN_SORT", raising=False)
    rng = np.random.default_rng(0)
    for n, n_keys in [(300_000, 40), (1_000_003, 1), (200_000, 3000), (90_000, 7)]:
        w = 1.0 / np.arange(1, n_keys + 1) ** 1.2
        key = rng.choice(n_keys, n, p=w / w.sum()).astype(np.int32)
        ts = BASE + rng.integers(0, 86_400_000_000, n)
        names = _check(orc, key, ts, n_keys)
        assert "tile_sort_kernel" in names, (n, n_keys, sorted(names))


def test_bucketed_bunched_and_duplicated_times_fall_back(orc, monkeypatch):
    """Buckets assume that a key's rows are spread over its time range; rows bunched into a sliver
    of it, or sharing one timestamp, overflow a tile -> the stand-by passes must deliver the order."""
    monkeypatch.setenv("TEMPO_BUCKET_MIN_ROWS", "3000")
    monkeypatch.delenv("TEMPO_PLAIN_SORT", raising=False)
    rng = np.random.default_rng(1)
    n = 400_000
    key = rng.integers(0, 5, n).astype(np.int32)
    # 1: 60 % of the rows inside one second of a day-long range
    ts = BASE + np.where(rng.random(n) < 0.6, 40_000_000_000 + rng.integers(0, 1_000_000, n),
                         rng.integers(0, 86_400_000_000, n))
    _check(orc, key, ts, 5)
    # 2: one timestamp shared by 100 000 rows of one key (ties keep input order)
    ts2 = BASE + rng.integers(0, 86_400_000_000, n)
    ts2[(key == 2) & (rng.random(n) < 0.9)] = BASE + 5_000_000
    _check(orc, key, ts2, 5)
    # 3: every row the same (key, ts)
    _check(orc, np.zeros(n, dtype=np.int32), np.full(n, BASE), 1)
    # 4: two distinct timestamps only
    _check(orc, key, BASE + rng.integers(0, 2, n) * 77, 5)


def test_bucketed_with_sequence_column_and_seconds_cast(orc, monkeypatch):
    monkeypatch.setenv("TEMPO_BUCKET_MIN_ROWS", "3000")
    monkeypatch.delenv("TEMPO_PLAIN_SORT", raising=False)
    rng = np.random.default_rng(2)
    n = 500_000
    key = rng.integers(0, 30, n).astype(np.int32)
    ts = BASE + rng.integers(0, 20_000 * 1_000_000, n)
    seq = rng.integers(0, 1000, n)
    seq_valid = (rng.random(n) >= 0.2).astype(np.uint8)
    names = _check(orc, key, ts, 30, seq=seq, seq_valid=seq_valid, floor_seconds=True)
    assert "tile_sort_kernel" in names


def test_bucketed_default_threshold_at_six_million_rows(monkeypatch):
    """At the library's own size threshold (no override), checked against torch's stable sort."""
    monkeypatch.delenv("TEMPO_BUCKET_MIN_ROWS", raising=False)
    monkeypatch.delenv("TEMPO_PLAIN_SORT", raising=False)
    from tempo_amd import ops
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    n, n_keys = 6_000_000, 1000
    key = torch.randint(0, n_keys, (n,), generator=g, device="cuda", dtype=torch.int32)
    key[: n // 4] = 3                                      # a hot key
    ts = BASE + torch.randint(0, 86_400_000_000, (n,), generator=g, device="cuda", dtype=torch.int64)
    perm, names = _kernels_of(lambda: ops.sort_order(key, ts, n_keys))
    assert "tile_sort_kernel" in names
    word = (key.to(torch.int64) << 40) | (ts - BASE)
    want = torch.sort(word, stable=True).indices.to(torch.int32)
    assert torch.equal(perm, want)


def _runs_table(rng, n, n_runs, n_keys, span_us, step_us):
    """n rows in n_runs time-ordered runs laid end to end (keys interleaved inside a run): what a
    rank holds after the key shuffle of time-ordered feeds."""
    key = rng.integers(0, n_keys, n).astype(np.int32)
    ts = BASE + rng.integers(0, span_us // step_us + 1, n) * step_us
    cuts = np.sort(rng.choice(np.arange(1, n), n_runs - 1, replace=False)) if n_runs > 1 else np.array([], int)
    bounds = np.r_[0, cuts, n]
    for a, b in zip(bounds[:-1], bounds[1:]):
        ts[a:b] = np.sort(ts[a:b], kind="stable")
    return key, ts


@pytest.mark.parametrize("n,n_runs,n_keys,span_s,step_us", [
    (200_000, 8, 100, 3000, 1),             # many rows per second once the seconds cast applies
    (150_001, 2, 7, 50, 1_000_000),         # heavy duplicates: 50 distinct times
    (70_000, 16, 1, 10_000, 1000),          # the most runs merged, a single series
    (9_000, 5, 40, 600, 1),                 # less than one tile
    (120_000, 17, 30, 3000, 1),             # one run too many: the full sort takes over
])
@pytest.mark.parametrize("floor_seconds", [False, True])
def test_time_ordered_runs_are_merged(orc, monkeypatch, n, n_runs, n_keys, span_s, step_us, floor_seconds):
    monkeypatch.delenv("TEMPO_PLAIN_SORT", raising=False)
    rng = np.random.default_rng(n + n_runs)
    key, ts = _runs_table(rng, n, n_runs, n_keys, span_s * 1_000_000, step_us)
    names = _check(orc, key, ts, n_keys, floor_seconds=floor_seconds) if floor_seconds else _check(orc, key, ts, n_keys)
    # rows that happen to be in time order across a run boundary merge two runs into one: fewer, never more
    assert ("tile_merge_kernel" in names) == (n_runs <= 16), sorted(names)


def test_time_ordered_runs_with_sequence_and_operators(orc, monkeypatch):
    """Runs whose ties the sequence column breaks (NULLS FIRST), and the operators that sit on the
    ordering: EMA and withRangeStats on a table of runs equal the oracle's."""
    from tempo_amd import ops
    rng = np.random.default_rng(77)
    n, n_runs, n_keys = 180_000, 6, 25
    key = rng.integers(0, n_keys, n).astype(np.int32)
    ts = BASE + rng.integers(0, 2000, n) * 1_000_000
    seq = rng.integers(0, 50, n)
    seq_valid = (rng.random(n) >= 0.2).astype(np.uint8)
    bounds = np.r_[0, np.sort(rng.choice(np.arange(1, n), n_runs - 1, replace=False)), n]
    for a, b in zip(bounds[:-1], bounds[1:]):
        o = np.lexsort((np.where(seq_valid[a:b] != 0, seq[a:b] + 1, 0), ts[a:b]))
        for arr in (key, ts, seq, seq_valid):
            arr[a:b] = arr[a:b][o]
    names = _check(orc, key, ts, n_keys, seq=seq, seq_valid=seq_valid, floor_seconds=True)
    assert "tile_merge_kernel" in names
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    x = rng.normal(0, 1, n)
    xv = (rng.random(n) >= 0.1).astype(np.uint8)
    perm, ema = ops.ema(d(key), d(ts), d(x), d(xv), n_keys, window=7, exp_factor=0.3, seq=d(seq), seq_valid=d(seq_valid))
    operm, oema = orc.ema(key, ts, x, xv, 7, 0.3, seq=seq, seq_valid=seq_valid)
    assert np.array_equal(perm.cpu().numpy(), operm) and np.array_equal(ema.cpu().numpy(), oema)
    out = ops.range_stats(d(key), d(ts), d(x), d(xv), n_keys, 30)
    ref = orc.range_stats(key, ts, x, xv, 30)
    assert np.array_equal(out["perm"].cpu().numpy(), ref["perm"]) and np.array_equal(out["count"].cpu().numpy(), ref["count"])
    np.testing.assert_allclose(out["mean"].cpu().numpy(), ref["mean"], rtol=1e-9, atol=1e-12)


def _per_key_ordered(rng, n, n_keys, jitter_us):
    """Interleaved devices, each in its own time order, clocks not aligned: rows are not in time
    order, yet every key's rows are."""
    per = n // n_keys
    n = per * n_keys
    slot = np.repeat(np.arange(per, dtype=np.int64), n_keys)
    key = np.tile(np.arange(n_keys, dtype=np.int32), per)
    ts = BASE + slot * 1_000_000 + rng.integers(-jitter_us, jitter_us + 1, n)
    return key, ts


@pytest.mark.parametrize("floor_seconds", [False, True])
def test_rows_ordered_within_every_key(orc, monkeypatch, floor_seconds):
    """Large enough for the sampled look (2^20 rows): the key passes are tried on their own, the
    check passes and no other sort kernel does any work."""
    monkeypatch.delenv("TEMPO_PLAIN_SORT", raising=False)
    rng = np.random.default_rng(21)
    key, ts = _per_key_ordered(rng, 1_300_000, 9, 200_000)
    assert (np.diff(ts) < 0).sum() > 1000            # far more steps back than a handful of runs
    names = _check(orc, key, ts, 9, floor_seconds=floor_seconds)
    assert "ordered_check_kernel" in names and "tile_sort_kernel" not in names


def test_key_order_guess_wrong_falls_back(orc, monkeypatch):
    """The sampled windows see ordered keys, but one key has a single row out of place far from any
    window's reach of its predecessor: the check fails and the stand-by full sort delivers."""
    monkeypatch.delenv("TEMPO_PLAIN_SORT", raising=False)
    rng = np.random.default_rng(22)
    key, ts = _per_key_ordered(rng, 1_300_000, 9, 200_000)
    n = len(ts)
    pos = 256 * (n // 512) + 2000                     # between two sampled windows (1024 rows every n / 512)
    pos += (4 - pos) % 9
    ts[pos] = ts[4]                                   # one row of key 4 jumps back to the start
    names = _check(orc, key, ts, 9)
    assert "ordered_check_kernel" in names
    # ... and EMA / range stats on such a table equal the oracle's
    from tempo_amd import ops
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    x = rng.normal(0, 1, n)
    perm, ema = ops.ema(d(key), d(ts), d(x), None, 9, window=5, exp_factor=0.3)
    operm, oema = orc.ema(key, ts, x, None, 5, 0.3)
    assert np.array_equal(perm.cpu().numpy(), operm) and np.array_equal(ema.cpu().numpy(), oema)


def test_time_ordered_runs_of_a_single_series_with_travelling_columns(orc, monkeypatch):
    """One series (no key bits) in a few time-ordered runs: after the run merge no key pass follows
    that would bring the operators' columns along, so they are fetched through the merged order.
    withRangeStats (double and integer metric), EMA, resample of two columns, lookback features and
    the fused join's left payload against the oracle."""
    from tempo_amd import _lib, ops
    monkeypatch.delenv("TEMPO_PLAIN_SORT", raising=False)
    rng = np.random.default_rng(4242)
    n, n_runs = 90_000, 5
    key, ts = _runs_table(rng, n, n_runs, 1, 40_000 * 1_000_000, 250_000)
    d = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()
    x = 100 + np.cumsum(rng.normal(0, 1, n)) * 0.01
    xi = rng.integers(-10**9, 10**9, n)
    xv = (rng.random(n) >= 0.1).astype(np.uint8)
    _lib.profile_enable(True)
    _lib.profile_reset()
    out = ops.range_stats(None, d(ts), d(x), d(xv), 1, 5)
    torch.cuda.synchronize()
    names = {nm for nm, _, _ in _lib.profile_read()}
    _lib.profile_enable(False)
    assert "tile_merge_kernel" in names, sorted(names)
    ref = orc.range_stats(key, ts, x, xv, 5)
    assert np.array_equal(out["perm"].cpu().numpy(), ref["perm"]) and np.array_equal(out["count"].cpu().numpy(), ref["count"])
    ok = ref["count"] > 0
    assert np.array_equal(out["min"].cpu().numpy()[ok], ref["min"][ok]) and np.array_equal(out["max"].cpu().numpy()[ok], ref["max"][ok])
    np.testing.assert_allclose(out["mean"].cpu().numpy()[ok], ref["mean"][ok], rtol=1e-9)
    out = ops.range_stats(d(key), d(ts), d(xi), None, 1, 5)
    ref = orc.range_stats(key, ts, xi, None, 5)
    for name in ("perm", "count", "min", "max", "sum"):
        assert np.array_equal(out[name].cpu().numpy(), ref[name]), name
    perm, ema = ops.ema(None, d(ts), d(x), d(xv), 1, window=9, exp_factor=0.25)
    operm, oema = orc.ema(key, ts, x, xv, 9, 0.25)
    assert np.array_equal(perm.cpu().numpy(), operm) and np.array_equal(ema.cpu().numpy(), oema)
    r = ops.resample(None, d(ts), 1, 60_000_000, "min", [d(x), d(xi)], [d(xv), None])
    o = orc.resample(key, ts, 60_000_000, "min", [x, xi], [xv, None])
    assert r["n"] == o["n"] and np.array_equal(r["start"].cpu().numpy(), o["start"])
    assert np.array_equal(r["valid"].cpu().numpy(), o["valid"])
    assert np.array_equal(r["f64"].cpu().numpy(), o["f64"]) and np.array_equal(r["i64"].cpu().numpy(), o["i64"])
    rows, lens, feat, fv = ops.lookback(None, d(ts), [d(x), d(xi.astype(np.float64))], [d(xv), None], 1, 4, exact_size=False)
    operm, olen, ofeat, ofv = orc.lookback(key, ts, [x, xi.astype(np.float64)], [xv, None], 4)
    assert np.array_equal(rows.cpu().numpy().astype(np.int64), operm) and np.array_equal(lens.cpu().numpy(), olen)
    assert np.array_equal(feat.cpu().numpy(), ofeat) and np.array_equal(fv.cpu().numpy(), ofv)
    # fused join: the left side is the table of runs, its payload column travels with the ordering
    nr = 30_000
    tr = BASE + rng.integers(0, 40_000 * 4, nr) * 250_000
    bid = rng.normal(0, 1, nr)
    j = ops.asof_join_table(None, d(ts), None, d(tr), 1, [d(x)], [d(bid)], [None])
    idx, _ = orc.asof_join(key, ts, np.zeros(nr, np.int32), tr, valid_r=[None])
    idx = np.asarray(idx).reshape(-1, n)
    order = np.lexsort((np.arange(n), ts))
    assert np.array_equal(j["perm"].cpu().numpy().astype(np.int64), order)
    assert np.array_equal(j["left"][0].cpu().numpy(), x[order])
    hit = idx[0][order] >= 0
    assert np.array_equal(j["right"][0][1].cpu().numpy() != 0, hit)
    assert np.array_equal(j["right"][0][0].cpu().numpy()[hit], bid[idx[0][order][hit]])
