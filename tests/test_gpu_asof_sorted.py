"""The one-sweep AS-OF join that tables in (key, ts[, seq]) order take (csrc/asof_sorted.hip) against
the CPU oracle, through the C ABI: record mode and fused table mode, sequence column, skew brackets,
slices longer than the on-chip buffer, more columns than the unrolled instance takes, a single
series, and equality with the general path on the same tables."""
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


def _sorted_tables(rng, nL, nR, nk, span_us, step_us, n_masks, null_p, with_seq, zipf=False):
    if zipf:
        w = 1.0 / np.arange(1, nk + 1) ** 1.2
        cdf = np.cumsum(w / w.sum())
        key_l = np.searchsorted(cdf, rng.random(nL)).clip(0, nk - 1).astype(np.int32)
        key_r = np.searchsorted(cdf, rng.random(nR)).clip(0, nk - 1).astype(np.int32)
    else:
        key_l = rng.integers(0, nk, nL).astype(np.int32)
        key_r = rng.integers(0, nk, nR).astype(np.int32)
    ts_l = BASE + rng.integers(0, span_us // step_us + 1, nL) * step_us
    ts_r = BASE + rng.integers(0, span_us // step_us + 1, nR) * step_us
    seq = seq_valid = None
    if with_seq:
        seq = rng.integers(-50, 50, nR)
        seq_valid = (rng.random(nR) >= 0.2).astype(np.uint8)
        o = np.lexsort((np.where(seq_valid == 1, seq, -10**9), ts_r, key_r))  # NULLS FIRST
        seq, seq_valid = seq[o], seq_valid[o]
    else:
        o = np.lexsort((ts_r, key_r))
    key_r, ts_r = key_r[o], ts_r[o]
    o = np.lexsort((ts_l, key_l))
    key_l, ts_l = key_l[o], ts_l[o]
    masks = [(rng.random(nR) >= null_p).astype(np.uint8) for _ in range(n_masks)]
    return key_l, ts_l, key_r, ts_r, seq, seq_valid, masks


def _check_records(orc, ops, data, nk, **kw):
    key_l, ts_l, key_r, ts_r, seq, seq_valid, masks = data
    row, midx, missing = ops.asof_join(_dev(key_l), _dev(ts_l), _dev(key_r), _dev(ts_r), nk, seq_r=_dev(seq),
                                       seq_r_valid=_dev(seq_valid), masks=[_dev(m) for m in masks], **kw)
    idx, omiss = orc.asof_join(key_l, ts_l, key_r, ts_r, seq_r=seq, seq_r_valid=seq_valid,
                               valid_r=[None] + masks, skip_nulls=True, **kw)
    assert np.array_equal(row.cpu().numpy().astype(np.int64), idx[0])
    got = midx.cpu().numpy().astype(np.int64)
    for m in range(len(masks)):
        assert np.array_equal(got[m], idx[1 + m]), f"mask {m}"
    assert missing.cpu().numpy().tolist() == omiss.tolist()


@pytest.mark.parametrize("seed", range(8))
def test_sorted_join_records_small(orc, ops, seed):
    """Small tables with many ties (coarse time grid), 0 to 3 masks, every other seed with a
    sequence column, every third in skew mode."""
    rng = np.random.default_rng(100 + seed)
    nL, nR = int(rng.integers(1, 4000)), int(rng.integers(1, 4000))
    data = _sorted_tables(rng, nL, nR, 1 + seed % 5, 60_000_000, 250_000, seed % 4, 0.4, with_seq=seed % 2 == 1)
    kw = dict(ts_partition_val=10, fraction=0.1) if seed % 3 == 0 else {}
    _check_records(orc, ops, data, 1 + seed % 5, **kw)


@pytest.mark.parametrize("nL,nR", [(3_000, 400_000), (400_000, 3_000), (250_000, 250_000)])
@pytest.mark.parametrize("with_seq", [False, True])
def test_sorted_join_records_uneven_sides(orc, ops, nL, nR, with_seq):
    """Few left rows against many right rows: a block's slice of the right table is far longer than
    the on-chip buffer and is worked through chunk by chunk, the last valid row of an almost always
    NULL column lying several chunks back; and the other way round (slices of a row or two)."""
    rng = np.random.default_rng(nL % 97 + with_seq)
    data = _sorted_tables(rng, nL, nR, 7, 3600 * 1_000_000, 1000, 3, 0.3, with_seq, zipf=True)
    data[6][1][:] = (rng.random(nR) >= 0.999).astype(np.uint8)   # almost always NULL
    data[6][2][:] = 0                                            # always NULL
    _check_records(orc, ops, data, 7)
    _check_records(orc, ops, data, 7, ts_partition_val=30, fraction=0.3)


def test_sorted_join_single_series(orc, ops):
    rng = np.random.default_rng(5)
    ts_l = np.sort(BASE + rng.integers(0, 10_000, 50_000) * 1000)
    ts_r = np.sort(BASE + rng.integers(0, 10_000, 40_000) * 1000)
    mask = (rng.random(40_000) >= 0.5).astype(np.uint8)
    row, midx, missing = ops.asof_join(None, _dev(ts_l), None, _dev(ts_r), 1, masks=[_dev(mask)])
    idx, omiss = orc.asof_join(np.zeros(50_000, np.int32), ts_l, np.zeros(40_000, np.int32), ts_r,
                               valid_r=[None, mask])
    assert np.array_equal(row.cpu().numpy().astype(np.int64), idx[0])
    assert np.array_equal(midx[0].cpu().numpy().astype(np.int64), idx[1])
    assert missing.cpu().numpy().tolist() == omiss.tolist()


@pytest.mark.parametrize("skip_nulls", [True, False])
@pytest.mark.parametrize("n_cols", [3, 7])
def test_sorted_join_fused_table(orc, ops, monkeypatch, skip_nulls, n_cols):
    """Fused table mode (3 nullable columns: the unrolled instance; 7: the general one), skew mode,
    against the oracle and against the general path on the very same tables."""
    rng = np.random.default_rng(44 + n_cols)
    nL, nR, nk = 120_000, 90_000, 23
    key_l, ts_l, key_r, ts_r, _, _, masks = _sorted_tables(rng, nL, nR, nk, 1800 * 1_000_000, 1000, n_cols, 0.15, False)
    pr, qty = rng.normal(100, 1, nL), rng.integers(1, 1000, nL)
    cols = [rng.normal(50 + c, 1, nR) for c in range(n_cols)]
    vol = rng.integers(0, 10**12, nR)

    def run():
        return ops.asof_join_table(_dev(key_l), _dev(ts_l), _dev(key_r), _dev(ts_r), nk, [_dev(pr), _dev(qty)],
                                   [_dev(ts_r)] + [_dev(c) for c in cols] + [_dev(vol)],
                                   [None] + [_dev(m) for m in masks] + [None], skip_nulls=skip_nulls,
                                   ts_partition_val=10, fraction=0.1)

    out = run()
    idx, omiss = orc.asof_join(key_l, ts_l, key_r, ts_r, valid_r=[None] + masks, skip_nulls=True,
                               ts_partition_val=10, fraction=0.1)
    assert np.array_equal(out["perm"].cpu().numpy(), np.arange(nL))
    assert np.array_equal(out["key"].cpu().numpy(), key_l) and np.array_equal(out["ts"].cpu().numpy(), ts_l)
    assert np.array_equal(out["left"][0].cpu().numpy(), pr) and np.array_equal(out["left"][1].cpu().numpy(), qty)
    srcs = [ts_r] + cols + [vol]
    col_mask = [None] + masks + [None]
    for c, src in enumerate(srcs):
        m = col_mask[c]
        j = idx[0] if (m is None or not skip_nulls) else idx[c]   # column c = 1 + mask (c - 1)
        ok = j >= 0
        if m is not None and not skip_nulls:
            ok = ok & (m[np.clip(j, 0, None)] == 1)
        vals, valid = out["right"][c]
        assert np.array_equal(valid.cpu().numpy().astype(bool), ok), c
        assert np.array_equal(vals.cpu().numpy(), np.where(ok, src[np.clip(j, 0, None)], 0)), c
    if skip_nulls:
        assert out["missing"].cpu().numpy().tolist() == omiss.tolist()
    monkeypatch.setenv("TEMPO_NO_SORTED_JOIN", "1")
    general = run()
    for c in range(len(srcs)):
        assert torch.equal(general["right"][c][0], out["right"][c][0]) and torch.equal(general["right"][c][1], out["right"][c][1])
    assert torch.equal(general["perm"], out["perm"]) and torch.equal(general["missing"], out["missing"])


def test_sorted_join_wider_than_one_sort_word(orc, ops):
    """10^6 keys x a year of microseconds x a sequence column do not fit one sort word; tables in
    order need no sort word at all."""
    rng = np.random.default_rng(9)
    nL, nR, nk = 60_000, 80_000, 1_000_000
    year = 366 * 86_400_000_000
    data = list(_sorted_tables(rng, nL, nR, nk, year, 1, 2, 0.3, True))
    data[0][:] = np.sort(rng.integers(0, 300, nL) * 3000).astype(np.int32)   # few distinct keys, so that rows meet
    data[2][:] = np.sort(rng.integers(0, 300, nR) * 3000).astype(np.int32)
    # restore (key, ts[, seq]) order after the key change
    key_l, ts_l, key_r, ts_r, seq, seq_valid, masks = data
    o = np.lexsort((ts_l, key_l)); key_l, ts_l = key_l[o], ts_l[o]
    o = np.lexsort((np.where(seq_valid == 1, seq, -10**9), ts_r, key_r))
    key_r, ts_r, seq, seq_valid = key_r[o], ts_r[o], seq[o], seq_valid[o]
    masks = [m[o] for m in masks]
    _check_records(orc, ops, (key_l, ts_l, key_r, ts_r, seq, seq_valid, masks), nk)


def test_sorted_join_columns_that_start_inside_an_allocation(orc, ops):
    """Views that begin one element into their allocations are not aligned for 16-byte accesses:
    the kernel then moves the left table's columns row by row; same joined table."""
    rng = np.random.default_rng(12)
    nL, nR, nk = 70_001, 50_001, 9
    key_l, ts_l, key_r, ts_r, _, _, masks = _sorted_tables(rng, nL, nR, nk, 900 * 1_000_000, 1000, 1, 0.2, False)
    pr, bid = rng.normal(100, 1, nL), rng.normal(99, 1, nR)
    d = lambda a: _dev(np.concatenate([a[:1], a]))[1:]
    out = ops.asof_join_table(d(key_l), d(ts_l), d(key_r), d(ts_r), nk, [d(pr)], [d(ts_r), d(bid)],
                              [None, d(masks[0])], skip_nulls=True)
    idx, _ = orc.asof_join(key_l, ts_l, key_r, ts_r, valid_r=[None] + masks, skip_nulls=True)
    assert np.array_equal(out["perm"].cpu().numpy(), np.arange(nL))
    assert np.array_equal(out["key"].cpu().numpy(), key_l) and np.array_equal(out["ts"].cpu().numpy(), ts_l)
    assert np.array_equal(out["left"][0].cpu().numpy(), pr)
    for c, (src, j) in enumerate([(ts_r, idx[0]), (bid, idx[1])]):
        vals, valid = out["right"][c]
        assert np.array_equal(valid.cpu().numpy().astype(bool), j >= 0), c
        assert np.array_equal(vals.cpu().numpy(), np.where(j >= 0, src[np.clip(j, 0, None)], 0)), c


def test_sorted_join_left_table_by_reference(orc, ops):
    """left_by_reference: a left table in order is not copied -- the result names the caller's own
    key / ts / payload tensors and the same joined columns; a left table out of order is ordered
    and copied as ever."""
    rng = np.random.default_rng(21)
    nL, nR, nk = 80_000, 60_000, 13
    key_l, ts_l, key_r, ts_r, _, _, masks = _sorted_tables(rng, nL, nR, nk, 900 * 1_000_000, 1000, 1, 0.2, False)
    pr, bid = rng.normal(100, 1, nL), rng.normal(99, 1, nR)
    dk, dt, dp = _dev(key_l), _dev(ts_l), _dev(pr)
    args = (_dev(key_r), _dev(ts_r), nk, [dp], [_dev(ts_r), _dev(bid)], [None, _dev(masks[0])])
    ref = ops.asof_join_table(dk, dt, *args)
    out = ops.asof_join_table(dk, dt, *args, left_by_reference=True)
    assert out["in_place"] and out["perm"] is None and not ref["in_place"]
    assert out["key"] is dk and out["ts"] is dt and out["left"][0] is dp
    for c in range(2):
        assert torch.equal(out["right"][c][0], ref["right"][c][0]) and torch.equal(out["right"][c][1], ref["right"][c][1])
    assert torch.equal(out["missing"], ref["missing"])
    o = rng.permutation(nL)
    out = ops.asof_join_table(_dev(key_l[o]), _dev(ts_l[o]), *args[:3], [_dev(pr[o])], *args[4:], left_by_reference=True)
    assert not out["in_place"]
    assert torch.equal(out["key"], ref["key"]) and torch.equal(out["ts"], ref["ts"])
    for c in range(2):
        assert torch.equal(out["right"][c][1], ref["right"][c][1])
