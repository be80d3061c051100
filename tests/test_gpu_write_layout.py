"""tmp_write_layout (event_dt / event_time / date-partitioned clustering order of
python/tempo/io.py:29-39) against the CPU oracle through the C ABI: every output bit-exact."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu

BASE = 1_596_240_000_000_000
DAY = 86_400_000_000


def _dev(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope="module")
def ops():
    from tempo_amd import ops
    return ops


def _check(orc, ops, key, ts, n_keys):
    dt, tm, perm = orc.write_layout(key, ts)
    g_dt, g_tm, g_perm, g_rng = ops.write_layout(_dev(key), _dev(ts), n_keys)
    torch.cuda.synchronize()
    assert np.array_equal(g_dt.cpu().numpy(), dt)
    assert np.array_equal(g_tm.cpu().numpy().view(np.int64), tm.view(np.int64))
    assert np.array_equal(g_perm.cpu().numpy().astype(np.int64), perm)
    if len(ts):
        assert g_rng.cpu().tolist() == [int(dt.min()), int(dt.max())]


@pytest.mark.parametrize("seed,days,nkeys,unit", [(0, 1, 5, 1), (1, 3, 40, 1), (2, 400, 7, 1000),
                                                   (3, 70_000, 3, 1), (4, 12, 0, 1_000_000), (5, 2, 1, 1)])
def test_write_layout_bit_exact(orc, ops, seed, days, nkeys, unit):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 60_000))
    ts = BASE + rng.integers(0, days * DAY // unit, n) * unit
    if seed == 3:
        ts -= 40_000 * DAY  # dates on both sides of the epoch
    key = rng.integers(0, nkeys, n).astype(np.int32) if nkeys else None
    _check(orc, ops, key, ts, max(nkeys, 1))


def test_write_layout_ordered_and_tiny_inputs(orc, ops):
    rng = np.random.default_rng(9)
    n = 30_000
    key = np.sort(rng.integers(0, 9, n).astype(np.int32))
    ts = BASE + np.sort(rng.integers(0, 5 * DAY, n))
    o = np.lexsort((ts, key))
    _check(orc, ops, key[o], ts[o], 9)          # rows already in (key, ts) order, several dates
    _check(orc, ops, np.sort(key), ts, 9)       # a time-ordered feed
    _check(orc, ops, key[:1], ts[:1], 9)
    dt, tm, perm, _ = ops.write_layout(_dev(key[:0]), _dev(ts[:0]), 9)
    assert dt.numel() == 0 and perm.numel() == 0
    # equal (date, key, ts) rows keep their input order
    same = np.full(5000, BASE + 5, dtype=np.int64)
    _check(orc, ops, np.zeros(5000, np.int32), same, 1)


def test_write_layout_large_properties(ops):
    """2x10^8 rows: dates non-decreasing along the order, (key, ts) non-decreasing within a date,
    the order is a permutation, event_time stays inside [0, 235960)."""
    n = 200_000_000
    g = torch.Generator(device="cuda").manual_seed(11)
    key = torch.randint(0, 1000, (n,), generator=g, device="cuda", dtype=torch.int32)
    ts = BASE + torch.randint(0, 30 * DAY, (n,), generator=g, device="cuda", dtype=torch.int64)
    dt, tm, perm, rng = ops.write_layout(key, ts, 1000)
    p = perm.long()
    d, k, t = dt[p], key[p], ts[p]
    assert bool((d[1:] >= d[:-1]).all())
    same_d = d[1:] == d[:-1]
    assert bool(((k[1:] >= k[:-1]) | ~same_d).all())
    assert bool(((t[1:] >= t[:-1]) | ~same_d | (k[1:] != k[:-1])).all())
    del d, k, t, same_d
    seen = torch.zeros(n, dtype=torch.uint8, device="cuda")
    seen[p] = 1
    assert int(seen.sum()) == n
    assert float(tm.min()) >= 0.0 and float(tm.max()) < 235960.0
    assert rng.cpu().tolist() == [int(dt.min()), int(dt.max())]
    assert bool((dt.long() == torch.div(ts, DAY, rounding_mode="floor")).all())


def test_tsdf_write_round_trip(tmp_path):
    """TSDF.write on the device path: the reference's test input (tsdf_tests.py:746-787)."""
    import tsdf_golden_cases as cases
    from tempo_amd import TSDF
    cases.case_write_to_delta(TSDF)
