"""HIP AS-OF match against the CPU oracle (bit-exact indices), through the C ABI."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu

BASE = 1_596_240_000_000_000


def _dev(a, dtype=None):
    if a is None:
        return None
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def _make(rng, nL, nR, nkeys, span_us, step_us, n_masks, null_p, with_seq, sorted_input=False):
    key_l = rng.integers(0, nkeys, nL).astype(np.int32)
    key_r = rng.integers(0, nkeys, nR).astype(np.int32)
    ts_l = BASE + rng.integers(0, span_us // step_us + 1, nL) * step_us
    ts_r = BASE + rng.integers(0, span_us // step_us + 1, nR) * step_us
    if sorted_input:
        o = np.lexsort((ts_l, key_l)); key_l, ts_l = key_l[o], ts_l[o]
        o = np.lexsort((ts_r, key_r)); key_r, ts_r = key_r[o], ts_r[o]
    masks = [(rng.random(nR) >= null_p).astype(np.uint8) for _ in range(n_masks)]
    seq = seq_valid = None
    if with_seq:
        seq = rng.integers(-1000, 1000, nR)
        seq_valid = (rng.random(nR) >= 0.1).astype(np.uint8)
    return key_l, ts_l, key_r, ts_r, seq, seq_valid, masks


def _run_both(orc, ops, data, nkeys, **kw):
    key_l, ts_l, key_r, ts_r, seq, seq_valid, masks = data
    row, midx, missing = ops.asof_join(_dev(key_l), _dev(ts_l), _dev(key_r), _dev(ts_r), nkeys,
                                       seq_r=_dev(seq), seq_r_valid=_dev(seq_valid),
                                       masks=[_dev(m) for m in masks], **kw)
    torch.cuda.synchronize()
    # oracle: column 0 has no NULLs (row rule), then one column per mask
    idx, omiss = orc.asof_join(key_l, ts_l, key_r, ts_r, seq_r=seq, seq_r_valid=seq_valid,
                               valid_r=[None] + masks, skip_nulls=True, **kw)
    assert np.array_equal(row.cpu().numpy().astype(np.int64), idx[0])
    got = midx.cpu().numpy().astype(np.int64)
    for m in range(len(masks)):
        assert np.array_equal(got[m], idx[1 + m]), f"mask {m}"
    assert missing.cpu().numpy().tolist() == omiss.tolist()
    # skipNulls=False rule == the row index for every column
    idx2, _ = orc.asof_join(key_l, ts_l, key_r, ts_r, seq_r=seq, seq_r_valid=seq_valid,
                            valid_r=masks[:1] or [None], skip_nulls=False,
                            **{k: v for k, v in kw.items() if k == "ts_is_seconds"}) \
        if "ts_partition_val" not in kw else (None, None)
    if idx2 is not None:
        assert np.array_equal(row.cpu().numpy().astype(np.int64), idx2[0])


@pytest.fixture(scope="module")
def ops():
    from tempo_amd import ops
    return ops


@pytest.mark.parametrize("seed", range(6))
def test_asof_small_random(orc, ops, seed):
    rng = np.random.default_rng(seed)
    nL, nR = int(rng.integers(1, 3000)), int(rng.integers(1, 3000))
    data = _make(rng, nL, nR, 5, 60_000_000, 250_000, seed % 3, 0.3, with_seq=seed % 2 == 1)
    _run_both(orc, ops, data, 5)


@pytest.mark.parametrize("with_seq", [False, True])
def test_asof_medium_unsorted(orc, ops, with_seq):
    rng = np.random.default_rng(42)
    data = _make(rng, 300_000, 200_000, 100, 6 * 3600 * 1_000_000, 1000, 2, 0.01, with_seq)
    _run_both(orc, ops, data, 100)


def test_asof_presorted_inputs_skip_sort(orc, ops):
    rng = np.random.default_rng(7)
    data = _make(rng, 100_000, 150_000, 37, 3600 * 1_000_000, 1000, 1, 0.05, False, sorted_input=True)
    _run_both(orc, ops, data, 37)


@pytest.mark.parametrize("v,fraction", [(10, 0.1), (3, 0.5), (7.5, 0.25), (1, 0.1)])
def test_asof_skew_brackets(orc, ops, v, fraction):
    rng = np.random.default_rng(11)
    data = _make(rng, 50_000, 20_000, 20, 600 * 1_000_000, 50_000, 2, 0.2, False)
    _run_both(orc, ops, data, 20, ts_partition_val=v, fraction=fraction)


def test_asof_zipf_heavy_key_and_long_null_runs(orc, ops):
    rng = np.random.default_rng(5)
    nL, nR, nkeys = 200_000, 200_000, 1000
    w = 1.0 / np.arange(1, nkeys + 1) ** 1.2
    cdf = np.cumsum(w / w.sum())
    key_l = np.searchsorted(cdf, rng.random(nL)).clip(0, nkeys - 1).astype(np.int32)
    key_r = np.searchsorted(cdf, rng.random(nR)).clip(0, nkeys - 1).astype(np.int32)
    ts_l = BASE + rng.integers(0, 86_400_000_000, nL)
    ts_r = BASE + rng.integers(0, 86_400_000_000, nR)
    m0 = (rng.random(nR) >= 0.995).astype(np.uint8)  # almost always NULL
    m1 = np.zeros(nR, np.uint8)                      # always NULL
    _run_both(orc, ops, (key_l, ts_l, key_r, ts_r, None, None, [m0, m1]), nkeys)


def test_asof_empty_and_single_series(orc, ops):
    rng = np.random.default_rng(3)
    key_l, ts_l, key_r, ts_r, _, _, masks = _make(rng, 1000, 0, 3, 10_000_000, 1000, 1, 0.5, False)
    row, midx, missing = ops.asof_join(_dev(key_l), _dev(ts_l), _dev(key_r), _dev(ts_r), 3,
                                       masks=[_dev(masks[0])])
    assert (row.cpu().numpy() == -1).all() and (midx.cpu().numpy() == -1).all()
    assert missing.cpu().numpy().tolist() == [1, 1]
    # no partition columns: one global series
    ts_l = BASE + rng.integers(0, 10_000, 5000) * 1000
    ts_r = BASE + rng.integers(0, 10_000, 4000) * 1000
    row, _, _ = ops.asof_join(None, _dev(ts_l), None, _dev(ts_r), 1)
    idx, _ = orc.asof_join(np.zeros(5000, np.int32), ts_l, np.zeros(4000, np.int32), ts_r,
                           valid_r=[None])
    assert np.array_equal(row.cpu().numpy().astype(np.int64), idx[0])


def test_sort_order_matches_oracle(orc, ops):
    rng = np.random.default_rng(9)
    n = 250_000
    key = rng.integers(0, 50, n).astype(np.int32)
    ts = BASE + rng.integers(-3_000_000, 900 * 1_000_000, n)
    seq = rng.integers(0, 4, n)
    seq_valid = (rng.random(n) >= 0.2).astype(np.uint8)
    perm = ops.sort_order(_dev(key), _dev(ts), 50, floor_seconds=True, seq=_dev(seq),
                          seq_valid=_dev(seq_valid))
    operm, _ = orc.window_order(key, ts, seq=seq, seq_valid=seq_valid)
    assert np.array_equal(perm.cpu().numpy().astype(np.int64), operm)


def test_gather(ops):
    rng = np.random.default_rng(1)
    src = rng.normal(size=1000)
    valid = (rng.random(1000) > 0.3).astype(np.uint8)
    idx = rng.integers(-1, 1000, 5000).astype(np.int32)
    out, ov = ops.gather(_dev(src), _dev(idx), _dev(valid))
    ok = (idx >= 0) & (valid[np.clip(idx, 0, None)] == 1)
    assert np.array_equal(ov.cpu().numpy().astype(bool), ok)
    assert np.array_equal(out.cpu().numpy(), np.where(ok, src[np.clip(idx, 0, None)], 0.0))


@pytest.mark.parametrize("with_seq", [False, True])
def test_asof_time_ordered_feeds(orc, ops, with_seq):
    """Both tables in arrival (time) order with interleaved keys: key-only radix passes."""
    rng = np.random.default_rng(21)
    key_l, ts_l, key_r, ts_r, seq, seq_valid, masks = _make(rng, 150_000, 120_000, 64, 3600 * 1_000_000, 1000, 2,
                                                            0.05, with_seq)
    o = np.argsort(ts_l, kind="stable"); key_l, ts_l = key_l[o], ts_l[o]
    if with_seq:  # arrival order = (ts, seq NULLS FIRST)
        o = np.lexsort((np.where(seq_valid == 1, seq, -10**9), ts_r))
    else:
        o = np.argsort(ts_r, kind="stable")
    key_r, ts_r = key_r[o], ts_r[o]
    masks = [m[o] for m in masks]
    if with_seq:
        seq, seq_valid = seq[o], seq_valid[o]
    _run_both(orc, ops, (key_l, ts_l, key_r, ts_r, seq, seq_valid, masks), 64)


def test_asof_sorted_output_mode(orc, ops):
    """sorted_output: the same matches, delivered in (key, ts) order with the row permutation."""
    rng = np.random.default_rng(33)
    key_l, ts_l, key_r, ts_r, _, _, masks = _make(rng, 70_000, 50_000, 17, 900 * 1_000_000, 1000, 2, 0.1, False)
    row, midx, missing, perm, key_s, ts_s = ops.asof_join(_dev(key_l), _dev(ts_l), _dev(key_r), _dev(ts_r), 17,
                                                          masks=[_dev(m) for m in masks], sorted_output=True)
    idx, omiss = orc.asof_join(key_l, ts_l, key_r, ts_r, valid_r=[None] + masks)
    p = perm.cpu().numpy().astype(np.int64)
    assert np.array_equal(p, np.lexsort((np.arange(len(ts_l)), ts_l, key_l)))  # stable (key, ts) order
    assert np.array_equal(key_s.cpu().numpy(), key_l[p]) and np.array_equal(ts_s.cpu().numpy(), ts_l[p])
    assert np.array_equal(row.cpu().numpy(), idx[0][p])
    for m in range(2):
        assert np.array_equal(midx[m].cpu().numpy(), idx[1 + m][p])
    assert missing.cpu().numpy().tolist() == omiss.tolist()
    # an empty right side still orders the left side
    row, midx, missing, perm, key_s, ts_s = ops.asof_join(_dev(key_l), _dev(ts_l), _dev(key_r[:0]), _dev(ts_r[:0]),
                                                          17, sorted_output=True)
    assert (row.cpu().numpy() == -1).all() and np.array_equal(perm.cpu().numpy().astype(np.int64), p)
    assert missing.cpu().numpy().tolist() == [1]


@pytest.mark.parametrize("skip_nulls,order", [(True, "random"), (False, "random"), (True, "time"), (True, "sorted")])
def test_asof_fused_table(orc, ops, skip_nulls, order):
    """The fused join (payload carried through the sort) equals oracle indices + a plain take."""
    rng = np.random.default_rng(44)
    nL, nR, nk = 90_000, 60_000, 23
    key_l, ts_l, key_r, ts_r, _, _, masks = _make(rng, nL, nR, nk, 1800 * 1_000_000, 1000, 2, 0.15, False)
    if order == "time":
        o = np.argsort(ts_l, kind="stable"); key_l, ts_l = key_l[o], ts_l[o]
        o = np.argsort(ts_r, kind="stable"); key_r, ts_r = key_r[o], ts_r[o]; masks = [m[o] for m in masks]
    if order == "sorted":
        o = np.lexsort((ts_l, key_l)); key_l, ts_l = key_l[o], ts_l[o]
        o = np.lexsort((ts_r, key_r)); key_r, ts_r = key_r[o], ts_r[o]; masks = [m[o] for m in masks]
    pr = rng.normal(100, 1, nL)
    qty = rng.integers(1, 1000, nL)
    bid, ask, vol = rng.normal(99, 1, nR), rng.normal(101, 1, nR), rng.integers(0, 10**12, nR)
    out = ops.asof_join_table(_dev(key_l), _dev(ts_l), _dev(key_r), _dev(ts_r), nk, [_dev(pr), _dev(qty)],
                              [_dev(ts_r), _dev(bid), _dev(ask), _dev(vol)], [None, _dev(masks[0]), _dev(masks[1]), None],
                              skip_nulls=skip_nulls)
    idx, omiss = orc.asof_join(key_l, ts_l, key_r, ts_r, valid_r=[None] + masks, skip_nulls=skip_nulls)
    p = out["perm"].cpu().numpy().astype(np.int64)
    assert np.array_equal(p, np.lexsort((np.arange(nL), ts_l, key_l)))
    assert np.array_equal(out["key"].cpu().numpy(), key_l[p]) and np.array_equal(out["ts"].cpu().numpy(), ts_l[p])
    assert np.array_equal(out["left"][0].cpu().numpy(), pr[p]) and np.array_equal(out["left"][1].cpu().numpy(), qty[p])
    src_cols = [ts_r, bid, ask, vol]
    col_idx = [idx[0], idx[1], idx[2], idx[0]]
    col_mask = [None, masks[0], masks[1], None]
    for c in range(4):
        j = col_idx[c][p]
        ok = j >= 0
        if col_mask[c] is not None and not skip_nulls:
            ok = ok & (col_mask[c][np.clip(j, 0, None)] == 1)
        vals, valid = out["right"][c]
        assert np.array_equal(valid.cpu().numpy().astype(bool), ok), c
        want = np.where(ok, src_cols[c][np.clip(j, 0, None)], 0)
        assert np.array_equal(vals.cpu().numpy(), want), c
    if skip_nulls:
        assert out["missing"].cpu().numpy().tolist() == omiss.tolist()


@pytest.mark.parametrize("step_l,step_r", [(1000, 1000), (1_000_000, 1000), (1_000_000, 1_000_000), (1000, 1), (7, 1000)])
def test_asof_timestamp_granularity(orc, ops, step_l, step_r):
    """Whole-millisecond / whole-second microsecond columns sort on a shorter word (the time field
    is stored in units of the common granularity); both signs of the epoch, sides that differ."""
    rng = np.random.default_rng(step_l % 97 + step_r % 89)
    nL, nR, nk = 30_000, 40_000, 11
    key_l = rng.integers(0, nk, nL).astype(np.int32)
    key_r = rng.integers(0, nk, nR).astype(np.int32)
    ts_l = rng.integers(-3_000_000_000 // step_l, 3_000_000_000 // step_l, nL) * step_l
    ts_r = rng.integers(-3_000_000_000 // step_r, 3_000_000_000 // step_r, nR) * step_r
    masks = [(rng.random(nR) >= 0.2).astype(np.uint8)]
    _run_both(orc, ops, (key_l, ts_l, key_r, ts_r, None, None, masks), nk)
    _run_both(orc, ops, (key_l, ts_l, key_r, ts_r, None, None, masks), nk, ts_partition_val=30, fraction=0.3)


YEAR_US = 366 * 86_400_000_000


def _wide_tables(rng, nL, nR, nk, n_masks, seq_span):
    key_l = rng.integers(0, nk, nL).astype(np.int32)
    key_r = rng.integers(0, nk, nR).astype(np.int32)
    # few distinct times per key, so that equal (key, ts) pairs -- where the sequence decides -- occur
    ts_l = BASE + rng.integers(0, 2000, nL) * (YEAR_US // 2000) + rng.integers(0, 3, nL)
    ts_r = BASE + rng.integers(0, 2000, nR) * (YEAR_US // 2000) + rng.integers(0, 3, nR)
    ts_l[0], ts_r[0] = BASE, BASE + YEAR_US          # the whole year, to the microsecond: 45 bits
    masks = [(rng.random(nR) >= 0.3).astype(np.uint8) for _ in range(n_masks)]
    seq = seq_valid = None
    if seq_span:
        seq = rng.integers(-seq_span, seq_span, nR)
        seq_valid = (rng.random(nR) >= 0.15).astype(np.uint8)
    return key_l, ts_l, key_r, ts_r, seq, seq_valid, masks


@pytest.mark.parametrize("nk,n_masks,seq_span,what", [
    (900, 2, 2**30, "sequence values sorted in passes of their own"),          # 10 + 45 + 32 + 2 bits
    (12_000, 8, 0, "validity bits beside the sort words"),                     # 14 + 45 + 8 bits
    (3_000, 9, 2**20, "both"),                                                 # 12 + 45 + 22 + 9 bits
    (1_000_000, 8, 2**12, "key dictionary joined in halves as well"),          # 20 + 45 alone exceed a word
])
def test_asof_tables_wider_than_one_sort_word(orc, ops, nk, n_masks, seq_span, what):
    """(key, time, sequence, validity bits) beyond 64 bits: the reference joins such tables
    (python/tempo/tsdf.py:117-121 has no width limit), so must the device path -- record mode and
    fused table mode, with and without skipNulls."""
    rng = np.random.default_rng(nk)
    nL, nR = 60_000, 80_000
    data = _wide_tables(rng, nL, nR, nk, n_masks, seq_span)
    _run_both(orc, ops, data, nk)
    key_l, ts_l, key_r, ts_r, seq, seq_valid, masks = data
    vals = [rng.normal(0, 1, nR) for _ in masks]
    for skip_nulls in (True, False):
        out = ops.asof_join_table(_dev(key_l), _dev(ts_l), _dev(key_r), _dev(ts_r), nk, [],
                                  [_dev(ts_r)] + [_dev(v) for v in vals], [None] + [_dev(m) for m in masks],
                                  skip_nulls=skip_nulls, seq_r=_dev(seq), seq_r_valid=_dev(seq_valid))
        idx, omiss = orc.asof_join(key_l, ts_l, key_r, ts_r, seq_r=seq, seq_r_valid=seq_valid,
                                   valid_r=[None] + masks, skip_nulls=skip_nulls)
        p = out["perm"].cpu().numpy().astype(np.int64)
        assert np.array_equal(p, np.lexsort((np.arange(nL), ts_l, key_l)))
        for c, src in enumerate([ts_r] + vals):
            j = (idx[c] if skip_nulls else idx[0])[p]
            ok = j >= 0
            if c > 0 and not skip_nulls:
                ok = ok & (masks[c - 1][np.clip(j, 0, None)] == 1)
            got, valid = out["right"][c]
            assert np.array_equal(valid.cpu().numpy().astype(bool), ok), (what, c)
            assert np.array_equal(got.cpu().numpy(), np.where(ok, src[np.clip(j, 0, None)], 0)), (what, c)


@pytest.mark.parametrize("order", ["per_key", "runs"])
def test_asof_fused_table_ordered_feeds(orc, ops, order):
    """Sides above 2^20 rows whose order lets the ordering step take a shortcut -- every key's rows
    in time order although the table is not (interleaved feeds with jitter), or a few time-ordered
    runs end to end (what a rank holds after a shuffle) -- with the left payload travelling along."""
    rng = np.random.default_rng(61)
    nk = 12
    per = 95_000
    n = per * nk

    def side(jitter):
        slot = np.repeat(np.arange(per, dtype=np.int64), nk)
        key = np.tile(np.arange(nk, dtype=np.int32), per)
        ts = BASE + slot * 1_000_000 + rng.integers(-jitter, jitter + 1, n)
        return key, ts

    key_l, ts_l = side(300_000)
    key_r, ts_r = side(300_000)
    if order == "runs":
        cuts = np.r_[0, np.sort(rng.choice(np.arange(1, n), 4, replace=False)), n]
        for a, b in zip(cuts[:-1], cuts[1:]):
            o = np.argsort(ts_l[a:b], kind="stable"); key_l[a:b], ts_l[a:b] = key_l[a:b][o], ts_l[a:b][o]
            o = np.argsort(ts_r[a:b], kind="stable"); key_r[a:b], ts_r[a:b] = key_r[a:b][o], ts_r[a:b][o]
        # the runs overlap in time: shuffle whole runs so that time steps back between them
        ts_l = np.concatenate([ts_l[a:b] for a, b in list(zip(cuts[:-1], cuts[1:]))[::-1]])
        key_l = np.concatenate([key_l[a:b] for a, b in list(zip(cuts[:-1], cuts[1:]))[::-1]])
    mask = (rng.random(n) >= 0.3).astype(np.uint8)
    pr = rng.normal(100, 1, n)
    bid = rng.normal(99, 1, n)
    out = ops.asof_join_table(_dev(key_l), _dev(ts_l), _dev(key_r), _dev(ts_r), nk, [_dev(pr)],
                              [_dev(ts_r), _dev(bid)], [None, _dev(mask)], skip_nulls=True)
    idx, _ = orc.asof_join(key_l, ts_l, key_r, ts_r, valid_r=[None, mask], skip_nulls=True)
    p = out["perm"].cpu().numpy().astype(np.int64)
    assert np.array_equal(p, np.lexsort((np.arange(n), ts_l, key_l)))
    assert np.array_equal(out["left"][0].cpu().numpy(), pr[p])
    for c, (src, j) in enumerate([(ts_r, idx[0][p]), (bid, idx[1][p])]):
        vals, valid = out["right"][c]
        assert np.array_equal(valid.cpu().numpy().astype(bool), j >= 0), c
        assert np.array_equal(vals.cpu().numpy(), np.where(j >= 0, src[np.clip(j, 0, None)], 0)), c


def test_asof_right_side_on_its_own_stream(orc, ops, monkeypatch):
    """Sides of 2^20 rows or more: the right side is ordered on the library's own stream, forked from
    and joined back into the caller's (TEMPO_JOIN_STREAMS=0 keeps everything on the caller's): the
    joined table is the oracle's, twice in a row (the stream and its events are reused), and equal to
    what the single-stream call returns."""
    rng = np.random.default_rng(67)
    nk, n = 40, 1_200_000
    key_l = rng.integers(0, nk, n).astype(np.int32)
    key_r = rng.integers(0, nk, n).astype(np.int32)
    ts_l = BASE + rng.integers(0, 3_600_000_000, n)
    ts_r = BASE + rng.integers(0, 3_600_000_000, n)
    mask = (rng.random(n) >= 0.2).astype(np.uint8)
    pr, bid, ask = rng.normal(100, 1, n), rng.normal(99, 1, n), rng.normal(101, 1, n)

    def run():
        return ops.asof_join_table(_dev(key_l), _dev(ts_l), _dev(key_r), _dev(ts_r), nk, [_dev(pr)],
                                   [_dev(ts_r), _dev(bid), _dev(ask)], [None, _dev(mask), None],
                                   skip_nulls=True, ts_partition_val=10, fraction=0.1)

    monkeypatch.setenv("TEMPO_JOIN_STREAMS", "0")
    one = run()
    monkeypatch.delenv("TEMPO_JOIN_STREAMS", raising=False)
    idx, _ = orc.asof_join(key_l, ts_l, key_r, ts_r, valid_r=[None, mask], skip_nulls=True,
                           ts_partition_val=10, fraction=0.1)
    for _ in range(2):
        two = run()
        p = two["perm"].cpu().numpy().astype(np.int64)
        assert np.array_equal(p, one["perm"].cpu().numpy())
        assert np.array_equal(two["left"][0].cpu().numpy(), pr[p])
        for c, (src, j) in enumerate([(ts_r, idx[0][p]), (bid, idx[1][p]), (ask, idx[0][p])]):
            vals, valid = two["right"][c]
            assert np.array_equal(valid.cpu().numpy().astype(bool), j >= 0), c
            assert np.array_equal(vals.cpu().numpy(), np.where(j >= 0, src[np.clip(j, 0, None)], 0)), c
            assert np.array_equal(vals.cpu().numpy(), one["right"][c][0].cpu().numpy()), c


@pytest.mark.parametrize("skip_nulls", [True, False])
def test_asof_fused_table_many_columns(orc, ops, skip_nulls):
    """More joined columns and masks than the merge's unrolled instance takes (six columns, five
    of them with NULLs): the general instance writes the same table."""
    rng = np.random.default_rng(73)
    nL, nR, nk, nm = 50_000, 40_000, 11, 5
    key_l, ts_l, key_r, ts_r, _, _, masks = _make(rng, nL, nR, nk, 900 * 1_000_000, 1000, nm, 0.25, False)
    pr = rng.normal(100, 1, nL)
    cols = [rng.normal(50 + c, 1, nR) for c in range(nm)]
    out = ops.asof_join_table(_dev(key_l), _dev(ts_l), _dev(key_r), _dev(ts_r), nk, [_dev(pr)],
                              [_dev(ts_r)] + [_dev(c) for c in cols], [None] + [_dev(m) for m in masks],
                              skip_nulls=skip_nulls)
    idx, omiss = orc.asof_join(key_l, ts_l, key_r, ts_r, valid_r=[None] + masks, skip_nulls=skip_nulls)
    p = out["perm"].cpu().numpy().astype(np.int64)
    assert np.array_equal(p, np.lexsort((np.arange(nL), ts_l, key_l)))
    assert np.array_equal(out["left"][0].cpu().numpy(), pr[p])
    for c, (src, j, mask) in enumerate([(ts_r, idx[0], None)] + [(cols[m], idx[1 + m], masks[m]) for m in range(nm)]):
        j = j[p]
        ok = j >= 0
        if mask is not None and not skip_nulls:
            ok = ok & (mask[np.clip(j, 0, None)] == 1)
        vals, valid = out["right"][c]
        assert np.array_equal(valid.cpu().numpy().astype(bool), ok), c
        assert np.array_equal(vals.cpu().numpy(), np.where(ok, src[np.clip(j, 0, None)], 0)), c
    if skip_nulls:
        assert out["missing"].cpu().numpy().tolist() == omiss.tolist()
