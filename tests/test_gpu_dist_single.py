"""The key shuffle on the GPU with a one-rank RCCL group: device bucketing kernel, per-column
all_to_all_single and the operator on the shuffled rows (multi-rank exchange logic is covered on
CPU by test_dist_gloo.py; 2+ GPUs are not available to this test)."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def test_shuffle_by_key_single_rank_nccl(orc):
    import torch.distributed as dist
    from tempo_amd import ops
    from tempo_amd.dist import shuffle_by_key
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        rng = np.random.default_rng(0)
        n = 200_000
        key = rng.integers(0, 300, n).astype(np.int32)
        ts = 1_596_240_000_000_000 + rng.integers(0, 3000, n) * 1_000_000
        x = rng.normal(10, 2, n)
        xv = (rng.random(n) > 0.1).astype(np.uint8)
        d = lambda a: torch.from_numpy(a).cuda()
        k2, (t2, x2, v2), send, recv = shuffle_by_key(d(key), [d(ts), d(x), d(xv)])
        assert send == [n] and recv == [n]
        # one owner: the bucketed order is the identity, every column arrives intact
        assert np.array_equal(k2.cpu().numpy(), key) and np.array_equal(t2.cpu().numpy(), ts)
        assert np.array_equal(x2.cpu().numpy(), x) and np.array_equal(v2.cpu().numpy(), xv)
        out = ops.range_stats(k2, t2, x2, v2, 300, 60)
        ref = orc.range_stats(key, ts, x, xv, 60)
        assert np.array_equal(out["count"].cpu().numpy(), ref["count"])
        np.testing.assert_allclose(out["mean"].cpu().numpy(), ref["mean"], rtol=1e-9)
    finally:
        if created:
            dist.destroy_process_group()


def test_hash_partition_many_parts(orc):
    from tempo_amd import _lib, ops
    rng = np.random.default_rng(1)
    n = 300_000
    key = rng.integers(0, 10_000, n).astype(np.int32)
    for parts in (2, 4, 8):
        perm, counts = ops.hash_partition(torch.from_numpy(key).cuda(), parts)
        owner = np.array([_lib.lib().tmp_key_owner(int(k), parts) for k in range(10_000)])[key]
        assert np.array_equal(counts.cpu().numpy(), np.bincount(owner, minlength=parts))
        assert np.array_equal(perm.cpu().numpy(), np.argsort(owner, kind="stable"))
        # shares are balanced to a few percent for 10k keys
        assert counts.max().item() < 1.2 * n / parts


def test_key_counts(orc):
    from tempo_amd import ops
    rng = np.random.default_rng(2)
    for n, n_keys, stride in [(250_000, 100, 1), (250_001, 10_000, 16), (70_000, 50_000, 3), (5, 7, 1), (0, 4, 1)]:
        key = rng.integers(0, n_keys, n).astype(np.int32)
        key[: n // 3] = 0  # a hot key
        got = ops.key_counts(torch.from_numpy(key).cuda(), n_keys, stride).cpu().numpy()
        assert np.array_equal(got, np.bincount(key[::stride], minlength=n_keys))


def _skew_tables(rng, n, n_keys, hot_share):
    base = 1_596_240_000_000_000
    mk = lambda: np.where(rng.random(n) < hot_share, 0, rng.integers(1, n_keys, n)).astype(np.int32)
    kl, kr = mk(), mk()
    tl = base + rng.integers(0, 3_600_000, n) * 1000
    tr = base + rng.integers(0, 3_600_000, n) * 1000
    return kl, tl, kr, tr


@pytest.mark.parametrize("v", [10, 2.5])
def test_partition_with_owner_table_and_time_slices(orc, v):
    """Device routing == the host statement of the rules, for plain tables, sliced keys and the
    halo copies, at 8 parts."""
    from host_partition import host_partition
    from tempo_amd import ops
    from tempo_amd.dist import plan_owners
    rng = np.random.default_rng(5)
    n, n_keys, world = 400_000, 300, 8
    kl, tl, kr, tr = _skew_tables(rng, n, n_keys, 0.3)
    counts = np.bincount(kl, minlength=n_keys) + np.bincount(kr, minlength=n_keys)
    dev = torch.device("cuda")
    plain = plan_owners(counts, world, dev)
    perm, cnt = ops.hash_partition(torch.from_numpy(kl).cuda(), world, plain.table)
    e_perm, e_cnt = host_partition(kl, world, plain.table)
    assert np.array_equal(perm.cpu().numpy(), e_perm.numpy()) and np.array_equal(cnt.cpu().numpy(), e_cnt.numpy())
    assert cnt.max().item() > 1.5 * n / world          # the hot key overloads its owner ...
    plan = plan_owners(counts, world, dev, skew=dict(ts_partition_val=v, fraction=0.1),
                       ts_range=(int(min(tl.min(), tr.min())), int(max(tl.max(), tr.max()))))
    assert plan.slices is not None
    for key, ts, halo in [(kl, tl, False), (kr, tr, False), (kr, tr, True)]:
        perm, cnt = ops.hash_partition(torch.from_numpy(key).cuda(), world, plan.table, plan.slices,
                                       torch.from_numpy(ts).cuda(), halo)
        e_perm, e_cnt = host_partition(key, world, plan.table, plan.slices, ts, halo)
        assert np.array_equal(cnt.cpu().numpy(), e_cnt.numpy())
        m = int(e_cnt.sum())
        assert np.array_equal(perm.cpu().numpy()[:m], e_perm.numpy()[:m])
        if not halo:
            assert cnt.max().item() < 1.1 * n / world   # ... unless it is cut along the time axis
        else:
            assert 0 < m < n // 20


def test_sliced_join_equals_whole_join(orc):
    """Eight simulated ranks on one GPU: route both sides (and the halo copies) with the device
    kernels, join every part on its own, and the union must be the single join row for row."""
    from tempo_amd import ops
    from tempo_amd.dist import plan_owners
    rng = np.random.default_rng(6)
    n, n_keys, world = 300_000, 50, 8
    kl, tl, kr, tr = _skew_tables(rng, n, n_keys, 0.4)
    bid = rng.normal(100, 1, n)
    bv = (rng.random(n) > 0.05).astype(np.uint8)
    pr = rng.normal(100, 1, n)
    d = lambda a: torch.from_numpy(a).cuda()
    counts = np.bincount(kl, minlength=n_keys) + np.bincount(kr, minlength=n_keys)
    plan = plan_owners(counts, world, torch.device("cuda"), skew=dict(ts_partition_val=10, fraction=0.1),
                       ts_range=(int(min(tl.min(), tr.min())), int(max(tl.max(), tr.max()))))
    assert plan.slices is not None

    def join(kl_, tl_, pr_, kr_, tr_, bid_, bv_):
        j = ops.asof_join_table(kl_, tl_, kr_, tr_, n_keys, [pr_], [tr_, bid_], [None, bv_], skip_nulls=True,
                                ts_partition_val=10, fraction=0.1)
        torch.cuda.synchronize()
        rts, rts_ok = j["right"][0]
        rb, rb_ok = j["right"][1]
        rows = np.stack([j["key"].cpu().numpy().astype(np.float64), j["ts"].cpu().numpy().astype(np.float64),
                         j["left"][0].cpu().numpy(),
                         np.where(rts_ok.cpu().numpy() != 0, rts.cpu().numpy().astype(np.float64), -1.0),
                         np.where(rb_ok.cpu().numpy() != 0, rb.cpu().numpy(), -1.0)], axis=1)
        return rows

    whole = join(d(kl), d(tl), d(pr), d(kr), d(tr), d(bid), d(bv))
    pl, cl = ops.hash_partition(d(kl), world, plan.table, plan.slices, d(tl))
    pr_, cr = ops.hash_partition(d(kr), world, plan.table, plan.slices, d(tr))
    ph, ch = ops.hash_partition(d(kr), world, plan.table, plan.slices, d(tr), True)
    pl, pr_, ph = pl.cpu().numpy(), pr_.cpu().numpy(), ph.cpu().numpy()
    ol = np.r_[0, np.cumsum(cl.cpu().numpy())]
    orr = np.r_[0, np.cumsum(cr.cpu().numpy())]
    oh = np.r_[0, np.cumsum(ch.cpu().numpy())]
    assert oh[-1] > 0
    parts = []
    for r in range(world):
        li = pl[ol[r]:ol[r + 1]]
        ri = np.r_[pr_[orr[r]:orr[r + 1]], ph[oh[r]:oh[r + 1]]]
        if len(li) == 0:
            continue
        parts.append(join(d(kl[li]), d(tl[li]), d(pr[li]), d(kr[ri]), d(tr[ri]), d(bid[ri]), d(bv[ri])))
    got = np.concatenate(parts)
    assert got.shape == whole.shape
    order = lambda a: a[np.lexsort(a.T[::-1])]
    assert np.array_equal(order(got), order(whole))


def test_shuffle_with_sliced_plan_single_rank_nccl():
    """The device shuffle with a time-sliced owner plan and halo routing on a one-rank RCCL
    group: every slice has the same owner, so nothing is copied and every row arrives once."""
    import torch.distributed as dist
    from tempo_amd.dist import OwnerPlan, shuffle_by_key
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29578")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        rng = np.random.default_rng(8)
        kl, tl, kr, tr = _skew_tables(rng, 100_000, 20, 0.5)
        d = lambda a: torch.from_numpy(a).cuda()
        i32 = lambda v: torch.tensor(v, dtype=torch.int32, device="cuda")
        table = torch.zeros(20, dtype=torch.int32, device="cuda")
        table[0] = -1
        plan = OwnerPlan(table, dict(ts_partition_val=10.0, v_is_int=True, ts_is_seconds=False,
                                     one_minus_fraction=0.9, bracket_min=int(tr.min() // 10_000_000),
                                     bracket_count=361, split_slices=i32([4]), split_first=i32([0]),
                                     slice_owner=i32([0, 0, 0, 0])))
        k2, (t2,), send, recv = shuffle_by_key(d(kr), [d(tr)], owners=plan, ts=d(tr), halo=True)
        assert send == [len(kr)] and recv == [len(kr)]
        assert np.array_equal(k2.cpu().numpy(), kr) and np.array_equal(t2.cpu().numpy(), tr)
    finally:
        if created:
            dist.destroy_process_group()


def test_chunked_exchange_equals_single_message(monkeypatch):
    """Columns larger than MAX_MESSAGE_BYTES go out as a sequence of grouped send/recv rounds over
    views (what every 10^9-row column does under RCCL).  With the limit at 64 KiB a 200k-row
    table needs up to 25 rounds; every column (1-, 4- and 8-byte elements, plus the empty halo
    route of a sliced plan) must arrive exactly as through one all_to_all_single."""
    import torch.distributed as dist
    import tempo_amd.dist as tdist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29579")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        rng = np.random.default_rng(9)
        n = 200_003
        kl, tl, kr, tr = _skew_tables(rng, n, 20, 0.5)
        x = rng.normal(0, 1, n)
        flag = (rng.random(n) > 0.5).astype(np.uint8)
        small = rng.integers(-9, 9, n).astype(np.int32)
        d = lambda a: torch.from_numpy(a).cuda()
        i32 = lambda v: torch.tensor(v, dtype=torch.int32, device="cuda")
        table = torch.zeros(20, dtype=torch.int32, device="cuda")
        table[0] = -1
        plan = tdist.OwnerPlan(table, dict(ts_partition_val=10.0, v_is_int=True, ts_is_seconds=False,
                                           one_minus_fraction=0.9, bracket_min=int(tr.min() // 10_000_000),
                                           bracket_count=361, split_slices=i32([4]), split_first=i32([0]),
                                           slice_owner=i32([0, 0, 0, 0])))
        cols = [d(tr), d(x), d(flag), d(small)]
        whole = tdist.shuffle_by_key(d(kr), cols, owners=plan, ts=cols[0], halo=True)
        monkeypatch.setattr(tdist, "MAX_MESSAGE_BYTES", 1 << 16)
        calls = []
        real = dist.all_to_all
        monkeypatch.setattr(dist, "all_to_all", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
        parts = tdist.shuffle_by_key(d(kr), cols, owners=plan, ts=cols[0], halo=True)
        torch.cuda.synchronize()
        assert len(calls) >= 25 + 13 + 4 + 13           # rounds of the 8-, 8-, 1- and 4-byte columns (+ key)
        assert parts[2] == whole[2] == [n] and parts[3] == whole[3] == [n]
        assert torch.equal(parts[0], whole[0]) and np.array_equal(parts[0].cpu().numpy(), kr)
        for a, b, src in zip(parts[1], whole[1], [tr, x, flag, small]):
            assert torch.equal(a, b) and np.array_equal(a.cpu().numpy(), src)
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("n,world", [(400_000, 8), (70_001, 3), (50, 2), (1_300_000, 64), (0, 4)])
def test_partition_move_equals_partition_plus_gather(n, world):
    """The fused route-and-move call delivers what tmp_hash_partition + one gather per column
    do: hashed owners, an owner table, time slices and the halo copies; 1-, 4- and 8-byte columns."""
    from tempo_amd import ops
    from tempo_amd.dist import plan_owners
    rng = np.random.default_rng(n + world)
    n_keys = 40
    kl, tl, kr, tr = _skew_tables(rng, max(n, 1), n_keys, 0.45)
    key, ts = kr[:n], tr[:n]
    cols = [key, ts, rng.normal(0, 1, n), (rng.random(n) > 0.5).astype(np.uint8), rng.integers(0, 9, n).astype(np.int32)]
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    dcols = [d(c) for c in cols]
    counts = np.bincount(key, minlength=n_keys) * 2
    plans = [(None, None)]
    if n:
        plain = plan_owners(counts, world, torch.device("cuda"))
        sliced = plan_owners(counts, world, torch.device("cuda"), skew=dict(ts_partition_val=10, fraction=0.1),
                             ts_range=(int(ts.min()), int(ts.max())))
        plans += [(plain.table, None), (sliced.table, sliced.slices)]
    for table, slices in plans:
        for halo in ([False, True] if slices is not None else [False]):
            dts = dcols[1] if slices is not None else None
            perm, cnt = ops.hash_partition(dcols[0], world, table, slices, dts, halo)
            outs, cnt2 = ops.partition_move(dcols[0], dcols, world, table, slices, dts, halo)
            torch.cuda.synchronize()
            assert np.array_equal(cnt.cpu().numpy(), cnt2.cpu().numpy())
            m = int(cnt.sum())
            assert m == n or halo
            p = perm.cpu().numpy()[:m]
            for c, o in zip(cols, outs):
                assert np.array_equal(o.cpu().numpy()[:m], c[p])


@pytest.mark.parametrize("n_keys", [1, 3])
def test_time_ranges_equal_single_gpu(orc, n_keys):
    """A table without partition columns (one global series; and one with three hot keys) over 8
    simulated ranks that own consecutive time ranges: routing kernels, halo / carried rows and the
    HIP operators per rank; the union of the ranks' results is the operator on the whole table."""
    import timesplit_cases as tc
    from tempo_amd import ops, timesplit
    comm = timesplit.SimComm(8)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    have, want = tc.check_range_stats(comm, ops, T, n=120_000, n_keys=n_keys, window=30)
    assert have.keys() == want.keys()
    tc.compare_range_stats(have, want)
    have, want = tc.check_ema(comm, ops, T, n=100_000, n_keys=n_keys, window=12, with_seq=n_keys > 1)
    assert have == want
    have, want = tc.check_lookback(comm, ops, T, n=60_000, n_keys=n_keys, size=6, exact=n_keys == 1)
    tc.compare_lookback(have, want)
    for skip_nulls in (True, False):
        have, want = tc.check_join(comm, ops, T, n=100_000, n_keys=n_keys, skip_nulls=skip_nulls)
        assert have == want


def test_time_ranges_single_rank_nccl(orc):
    """The same path through a one-rank RCCL group: exchange and all-gather on the device."""
    import torch.distributed as dist
    import timesplit_cases as tc
    from tempo_amd import ops, timesplit
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29580")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        comm = timesplit.GroupComm()
        T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
        have, want = tc.check_ema(comm, ops, T, n=50_000, n_keys=1, window=5)
        assert have == want
        have, want = tc.check_join(comm, ops, T, n=50_000, n_keys=2)
        assert have == want
    finally:
        if created:
            dist.destroy_process_group()


def test_hot_key_by_time_rest_by_hash_on_the_device(orc, monkeypatch):
    """One key holds half the rows and is cut along the time axis over 8 simulated ranks, the other
    keys go to their hash owners in the same exchange; HIP routing and operators throughout."""
    import timesplit_cases as tc
    from tempo_amd import ops, timesplit
    monkeypatch.setattr(tc, "HOT_SHARE", 0.5)
    comm = timesplit.SimComm(8)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    kw = dict(hot_keys=[0])
    have, want = tc.check_range_stats(comm, ops, T, n=90_000, n_keys=40, window=30, **kw)
    assert have.keys() == want.keys()
    tc.compare_range_stats(have, want)
    have, want = tc.check_ema(comm, ops, T, n=80_000, n_keys=40, window=12, **kw)
    assert have == want
    have, want = tc.check_join(comm, ops, T, n=80_000, n_keys=40, **kw)
    assert have == want
