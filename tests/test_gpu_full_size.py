"""The benchmark configurations at their full sizes, through the C ABI, checked by properties and
by independent device-side restatements that stay cheap at 10^8 - 10^9 rows (sorted searches and
prefix sums in torch; the oracle covers the same operators exactly at sizes it finishes in
seconds).  Inputs are the benchmark's own generators."""
import os
import sys

import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def gen():
    import bench
    return bench


def _is_permutation(perm, n):
    seen = torch.zeros(n, dtype=torch.uint8, device=perm.device)
    seen[perm.long()] = 1
    return perm.numel() == n and int(seen.sum()) == n


def _check_range_stats(gen, n, ordered=False, layout="time"):
    """withRangeStats(600 s) on n ticks of the benchmark's stream, 100 symbols: row order, frame counts
    (exact) for every row; sums / means of every row against a prefix-sum restatement (1e-7 at 10^8
    rows, 1e-3 at 10^9: the check's own cancellation over a prefix that long, not the kernel's -- a net
    for gross errors); and all seven statistics of 4096 sampled rows against their frames added up
    directly (1e-9).  Intermediate columns are dropped as soon as they are spent: at 10^9 rows each is 8 GB."""
    from tempo_amd import ops
    dev = torch.device("cuda", 0)
    torch.cuda.empty_cache()
    n_keys, w = 100, 600
    t = gen.gen_ticks(dev, n, n_keys, 2345)
    if layout == "runs":   # eight time-ordered runs end to end: what a rank holds after a key shuffle
        o = torch.arange(n, device=dev).view(-1, 8).t().reshape(-1)
        t = {k: v[o].contiguous() for k, v in t.items()}
        del o
    elif layout == "permuted":   # no order at all: the full (key, second) sort
        o = torch.randperm(n, device=dev)
        t = {k: v[o].contiguous() for k, v in t.items()}
        del o
    if ordered:   # rows laid out in (key, ts) order: the kernel then reads the table's own columns
        o = torch.argsort(t["key"], stable=True)   # (the stream is time-ordered: a stable sort by key does it)
        t = {k: v[o].contiguous() for k, v in t.items()}
        del o
    out = ops.range_stats(t["key"], t["ts"], t["x"], t["x_valid"], n_keys, w)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()   # the call's workspace
    if ordered:
        assert torch.equal(out["perm"], torch.arange(n, dtype=out["perm"].dtype, device=dev))
    assert _is_permutation(out["perm"], n)
    p = out["perm"].long()
    sec = torch.div(t["ts"][p], 1_000_000, rounding_mode="floor")
    comp = (t["key"][p].long() << 40) | (sec - int(sec.min()))
    del sec
    assert bool((comp[1:] >= comp[:-1]).all())                      # window order: (key, second)
    ok = t["x_valid"][p] != 0
    x = torch.where(ok, t["x"][p], torch.zeros((), dtype=torch.float64, device=dev))
    del p, t
    # frame of row i: rows of its key with second in [sec_i - w, sec_i], later same-second rows included
    rel = comp & ((1 << 40) - 1)
    lo_key = comp - rel.clamp_(max=w)
    del rel
    start = torch.searchsorted(comp, lo_key, right=False)
    del lo_key
    end = torch.searchsorted(comp, comp, right=True)
    del comp
    cv = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(ok.long(), 0)])
    count = cv[end] - cv[start]
    del cv
    assert torch.equal(out["count"], count)
    # ---- sampled rows: the frame's values gathered and added up directly
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    idx = torch.cat([torch.randint(0, n, (4094,), generator=g, device=dev), torch.tensor([0, n - 1], device=dev)])
    f0, f1 = start[idx], end[idx]
    width = int((f1 - f0).max())
    assert 0 < width < 4096
    pos = f0[:, None] + torch.arange(width, device=dev)[None, :]
    inside = pos < f1[:, None]
    pos.clamp_(max=n - 1)
    fv = x[pos]
    fok = ok[pos] & inside
    cnt = fok.sum(1)
    assert torch.equal(cnt, out["count"][idx])
    zero = torch.zeros((), dtype=torch.float64, device=dev)
    fsum = torch.where(fok, fv, zero).sum(1)
    fmean = fsum / cnt.clamp_min(1)
    dev2 = torch.where(fok, (fv - fmean[:, None]) ** 2, zero).sum(1)
    fstd = torch.sqrt(dev2 / (cnt - 1).clamp_min(1))
    inf = torch.full((), float("inf"), dtype=torch.float64, device=dev)
    fmin = torch.where(fok, fv, inf).min(1).values
    fmax = torch.where(fok, fv, -inf).max(1).values
    some, many = cnt > 0, cnt > 1

    def near(a, b, sel, what):
        err = ((a - b).abs() / b.abs().clamp_min(1e-300))[sel]
        assert err.numel() == 0 or float(err.max()) < 1e-9, (what, float(err.max()))

    near(out["sum"][idx], fsum, some & (fsum.abs() > 1e-6), "sum")
    near(out["mean"][idx], fmean, some & (fmean.abs() > 1e-6), "mean")
    near(out["stddev"][idx], fstd, many & (fstd > 1e-9), "stddev")
    assert torch.equal(out["min"][idx][some], fmin[some]) and torch.equal(out["max"][idx][some], fmax[some])
    zs = (x[idx] - fmean) / fstd
    zsel = many & ok[idx] & (fstd > 1e-9)
    assert float((out["zscore"][idx] - zs).abs()[zsel].max()) < 1e-7   # (x - mean) / stddev: the division magnifies mean's 1e-9
    del pos, inside, fv, fok, idx, f0, f1
    cs = torch.cat([torch.zeros(1, dtype=torch.float64, device=dev), torch.cumsum(x - 100.0 * ok, 0)])
    total = (cs[end] - cs[start]) + 100.0 * count
    del cs, start, end
    has = count > 0
    assert bool(((out["valid"] & 1) != 0).eq(has).all())
    coarse = 1e-7 if n <= 100_000_000 else 1e-3
    err = ((out["sum"] - total).abs() / total.abs().clamp_min(1.0))[has].max()
    assert float(err) < coarse
    mean = total / count.clamp_min(1)
    del total
    assert float(((out["mean"] - mean).abs() / mean.abs().clamp_min(1.0))[has].max()) < coarse
    del mean
    assert bool((out["min"][has] <= out["mean"][has] + 1e-9).all()) and bool((out["max"][has] >= out["mean"][has] - 1e-9).all())
    sd_ok = (out["valid"] & 2) != 0
    assert bool(sd_ok.eq(count > 1).all()) and bool((out["stddev"][sd_ok] >= 0).all())
    del sd_ok, count, has
    z_ok = (out["valid"] & 4) != 0
    z = (x - out["mean"]) / out["stddev"]
    assert float((out["zscore"] - z).abs()[z_ok].max()) < 1e-9 * 1e3
    del z, z_ok
    # own value lies inside its frame's extremes
    assert bool(((x >= out["min"]) & (x <= out["max"]))[ok].all())
    del out, x, ok
    torch.cuda.empty_cache()


def test_range_stats_100m_ticks_window_600(gen):
    _check_range_stats(gen, 100_000_000)


def test_range_stats_1b_ticks_window_600(gen):
    """The same at the size the benchmark's metric is quoted on: 10^9 ticks in one call."""
    _check_range_stats(gen, 1_000_000_000)


def test_range_stats_1b_ticks_in_key_ts_order(gen):
    """10^9 ticks that arrive in (key, ts) order: nothing is laid out, the rows stay where they are."""
    _check_range_stats(gen, 1_000_000_000, ordered=True)


@pytest.mark.parametrize("layout", ["runs", "permuted"])
def test_range_stats_1b_ticks_other_row_orders(gen, layout):
    """10^9 ticks as a rank holds them after a key shuffle (eight time-ordered runs end to end: the run
    merge) and in no order at all (the full sort)."""
    _check_range_stats(gen, 1_000_000_000, layout=layout)


@pytest.mark.parametrize("ordered", [False, True], ids=["random_order", "key_ts_order"])
def test_skew_join_1b_rows(gen, ordered):
    """asofJoin(tsPartitionVal=10, fraction=0.1) of 5x10^8 trades and 5x10^8 quotes with Zipf keys:
    the joined right timestamp is exactly the latest quote at or before the trade that the
    trade's bracket may see; NULL-skipping columns take a quote that is visible and not NULL.
    ordered: both tables laid out in (key, ts) order first -- the one-sweep join, with the left table
    taken by reference."""
    from tempo_amd import ops
    dev = torch.device("cuda", 0)
    torch.cuda.empty_cache()
    n = 500_000_000
    n_keys = 10_000
    q = gen.gen_trades_quotes(dev, n, n, n_keys, 4567)
    if ordered:
        for side, cols in (("l", ["key_l", "ts_l", "trade_pr"]), ("r", ["key_r", "ts_r", "bid", "ask", "bid_valid", "ask_valid"])):
            o = torch.argsort(q["ts_" + side], stable=True)
            o = o[torch.argsort(q["key_" + side][o], stable=True)]
            for c in cols:
                q[c] = q[c][o].contiguous()
            del o
    j = ops.asof_join_table(q["key_l"], q["ts_l"], q["key_r"], q["ts_r"], n_keys, [q["trade_pr"]],
                            [q["ts_r"], q["bid"]], [None, q["bid_valid"]], skip_nulls=True,
                            ts_partition_val=10, fraction=0.1, left_by_reference=ordered)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    if ordered:
        assert j["in_place"] and j["perm"] is None
        assert j["key"].data_ptr() == q["key_l"].data_ptr() and j["left"][0].data_ptr() == q["trade_pr"].data_ptr()
    else:
        assert _is_permutation(j["perm"], n)
        p = j["perm"].long()
        assert torch.equal(j["key"], q["key_l"][p]) and torch.equal(j["ts"], q["ts_l"][p])
        assert torch.equal(j["left"][0], q["trade_pr"][p])
        del p
    lcomp = (j["key"].long() << 40) | (j["ts"] - gen.BASE_US)
    assert bool((lcomp[1:] >= lcomp[:-1]).all())                     # (key, ts) order
    rcomp, rorder = torch.sort((q["key_r"].long() << 40) | (q["ts_r"] - gen.BASE_US))
    pos = torch.searchsorted(rcomp, lcomp, right=True) - 1           # latest quote <= trade, any key
    cand = rcomp[pos.clamp_min(0)]
    same_key = (pos >= 0) & ((cand >> 40) == (lcomp >> 40))
    cand_ts = (cand & ((1 << 40) - 1)) + gen.BASE_US
    del rcomp, lcomp, cand

    def bracket(ts):   # tsdf.py:178-180, the same IEEE operations
        d = ts.double() / 1000000.0
        part = 10.0 * torch.trunc(d / 10.0)
        return d, part

    d_l, part_l = bracket(j["ts"])
    d_r, part_r = bracket(cand_ts)
    del d_l
    visible = same_key & ((part_r == part_l) | ((part_r + 10.0 == part_l) & ((d_r - part_r) / 10.0 >= 1 - 0.1)))
    del d_r, part_r, part_l, same_key
    got_ts, got_ok = j["right"][0]
    assert torch.equal(got_ok != 0, visible)
    assert torch.equal(got_ts[visible], cand_ts[visible])
    # NULL skipping: where the latest visible quote has a bid, that is the bid taken; a taken bid
    # is never newer than the latest visible quote, and nothing is taken where nothing is visible
    bid, bid_ok = j["right"][1]
    bid_ok = bid_ok != 0
    assert not bool((bid_ok & ~visible).any())
    src = rorder[pos.clamp_min(0)]
    direct = visible & (q["bid_valid"][src] != 0)
    assert bool(bid_ok[direct].all()) and torch.equal(bid[direct], q["bid"][src][direct])
    assert int(j["missing"][0]) == int((~visible).any())


def test_resample_and_ema_500m_accelerometer(gen):
    """resample('min', mean) and EMA(30) on 5x10^8 accelerometer rows (9 users, 5 ms sampling)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_ops
    from tempo_amd import ops
    dev = torch.device("cuda", 0)
    key, ts, cols, n_keys = bench_ops.accelerometer(dev, 500_000_000)
    n = ts.numel()
    r = ops.resample(key, ts, n_keys, 60_000_000, "mean", cols[:1], [None])
    torch.cuda.synchronize()
    bucket = torch.div(ts, 60_000_000, rounding_mode="floor")
    comp = (key.long() << 40) | (bucket - int(bucket.min()))
    uniq, inv = torch.unique(comp, return_inverse=True)
    del comp, bucket
    m = uniq.numel()
    assert int(r["n"]) == m
    sums = torch.zeros(m, dtype=torch.float64, device=dev).index_add_(0, inv, cols[0])
    cnts = torch.zeros(m, dtype=torch.float64, device=dev).index_add_(0, inv, torch.ones(n, dtype=torch.float64, device=dev))
    del inv
    assert torch.equal(r["key"][:m].long(), uniq >> 40)
    want = sums / cnts
    got = r["f64"][0][:m]
    assert float(((got - want).abs() / want.abs().clamp_min(1e-3)).max()) < 1e-9
    del r, sums, cnts, uniq
    # EMA: the reference's sum of lagged terms in its own order, on the rows in window order
    perm, ema = ops.ema(key, ts, cols[0], None, n_keys, window=30, exp_factor=0.2)
    torch.cuda.synchronize()
    assert _is_permutation(perm, n)
    p = perm.long()
    k_s, x_s = key[p], cols[0][p]
    del p
    acc = torch.zeros(n, dtype=torch.float64, device=dev)
    for i, wgt in enumerate(ops.ema_weights(30, 0.2)):
        if i == 0:
            acc += wgt * x_s
        else:
            term = torch.zeros(n, dtype=torch.float64, device=dev)
            same = k_s[i:] == k_s[:-i]
            term[i:] = torch.where(same, wgt * x_s[:-i], torch.zeros((), dtype=torch.float64, device=dev))
            acc += term
            del term, same
    assert float((ema - acc).abs().max()) <= 1e-12


def test_join_then_lookback_chain_200m(gen):
    """The multi-GPU configuration's operator chain on one GPU at 10^8 + 10^8 rows: skew asofJoin,
    then withLookbackFeatures(['trade_pr', 'right_bid', 'right_ask'], 10) on the joined table.
    The joined table is in window order already, so every kept row's features are the ten rows
    before it, oldest first, NULLs of the joined columns carried as NULLs."""
    from tempo_amd import ops
    dev = torch.device("cuda", 0)
    n, n_keys, L = 100_000_000, 10_000, 10
    q = gen.gen_trades_quotes(dev, n, n, n_keys, 4567)
    j = ops.asof_join_table(q["key_l"], q["ts_l"], q["key_r"], q["ts_r"], n_keys, [q["trade_pr"]],
                            [q["bid"], q["ask"]], [q["bid_valid"], q["ask_valid"]], skip_nulls=True,
                            ts_partition_val=10, fraction=0.1)
    del q
    key, ts = j["key"], j["ts"]
    xs = [j["left"][0], j["right"][0][0], j["right"][1][0]]
    valid = [None, j["right"][0][1], j["right"][1][1]]
    rows, _, feat, fv = ops.lookback(key, ts, xs, valid, n_keys, L, exact_size=True)
    torch.cuda.synchronize()
    keep = torch.zeros(n, dtype=torch.bool, device=dev)
    keep[L:] = key[L:] == key[:-L]
    want_rows = torch.nonzero(keep).flatten()
    assert torch.equal(rows.long(), want_rows)
    for l in range(L):
        src = want_rows - (L - l)
        for f in range(3):
            ok = torch.ones_like(src, dtype=torch.bool) if valid[f] is None else valid[f][src] != 0
            assert torch.equal(fv[:, l, f] != 0, ok)
            assert torch.equal(feat[:, l, f][ok], xs[f][src][ok])
