"""Time-range sharding of series that cannot be hashed, with simulated ranks on the CPU: the
product's routing / halo / carry logic (tempo_amd/timesplit.py) around host stand-ins for the
device kernels (the oracle for the operators, numpy for the routing).  The union over ranks must
be the operator on the whole table."""
import numpy as np
import pytest
import torch

import oracle_ops
import timesplit_cases as tc
from host_partition import host_partition, host_take
from tempo_amd import timesplit

T = lambda a: torch.from_numpy(np.ascontiguousarray(a))


def _comm(world):
    return timesplit.SimComm(world, partition_fn=host_partition, take_fn=host_take)


@pytest.mark.parametrize("world,n_keys,window", [(3, 1, 20), (5, 3, 400), (4, 1, 0)])
def test_range_stats_by_time(orc, world, n_keys, window):
    have, want = tc.check_range_stats(_comm(world), oracle_ops, T, n=3000, n_keys=n_keys, window=window)
    assert have.keys() == want.keys()
    tc.compare_range_stats(have, want)


@pytest.mark.parametrize("world,n_keys,window,with_seq", [(3, 1, 7, False), (4, 3, 30, True), (2, 1, 1, False)])
def test_ema_by_time(orc, world, n_keys, window, with_seq):
    have, want = tc.check_ema(_comm(world), oracle_ops, T, n=2500, n_keys=n_keys, window=window, with_seq=with_seq)
    assert have == want        # bit for bit: the same lags meet the same taps


@pytest.mark.parametrize("world,n_keys,size,exact", [(3, 1, 5, True), (4, 2, 12, False)])
def test_lookback_by_time(orc, world, n_keys, size, exact):
    have, want = tc.check_lookback(_comm(world), oracle_ops, T, n=2000, n_keys=n_keys, size=size, exact=exact)
    tc.compare_lookback(have, want)


@pytest.mark.parametrize("world,n_keys,skip_nulls,with_seq", [(3, 1, True, False), (4, 3, False, False), (5, 2, True, True)])
def test_asof_join_by_time(orc, world, n_keys, skip_nulls, with_seq):
    have, want = tc.check_join(_comm(world), oracle_ops, T, n=2500, n_keys=n_keys, skip_nulls=skip_nulls,
                               with_seq=with_seq)
    assert have == want


def test_cuts_are_whole_seconds_and_shared():
    comm = _comm(4)
    rng = np.random.default_rng(0)
    ts = [T(tc.BASE + rng.integers(0, 10**9, 1000)) for _ in range(4)]

    def body(rank):
        cuts = yield from timesplit._cuts(4, [ts[rank]], 1_000_000)
        yield ("done", cuts)

    cuts = timesplit._drive(comm, body)
    assert all(torch.equal(c, cuts[0]) for c in cuts)
    assert (cuts[0] % 1_000_000 == 0).all() and (cuts[0][1:] >= cuts[0][:-1]).all()


def test_hot_key_by_time_and_the_rest_by_hash(orc, monkeypatch):
    """A key with 60 % of the rows is cut along the time axis, the other 19 keys hash to their
    owners in the same exchange: every operator's union over the ranks is still the whole result,
    and the hot key's rows really are spread (no rank holds more than half of them)."""
    monkeypatch.setattr(tc, "HOT_SHARE", 0.6)
    world, n_keys, kw = 4, 20, dict(hot_keys=[0])
    have, want = tc.check_range_stats(_comm(world), oracle_ops, T, n=4000, n_keys=n_keys, window=25, **kw)
    assert have.keys() == want.keys()
    tc.compare_range_stats(have, want)
    have, want = tc.check_ema(_comm(world), oracle_ops, T, n=3000, n_keys=n_keys, window=9, **kw)
    assert have == want
    have, want = tc.check_lookback(_comm(world), oracle_ops, T, n=2500, n_keys=n_keys, size=4, exact=False, **kw)
    tc.compare_lookback(have, want)
    have, want = tc.check_join(_comm(world), oracle_ops, T, n=3000, n_keys=n_keys, **kw)
    assert have == want
    # where the rows went
    s = tc.make_series(2, 3000, n_keys)
    tables = [dict(key=T(s["key"][sl]), ts=T(s["ts"][sl]), x=T(s["x"][sl]), x_valid=T(s["xv"][sl]), extra=[])
              for sl in tc.chunks(3000, world)]
    got = timesplit.ema_by_time(_comm(world), tables, n_keys, 9, 0.3, local_ops=oracle_ops, hot_keys=[0])
    hot_rows = [int((g["key"] == 0).sum()) for g in got]
    assert max(hot_rows) < 0.5 * sum(hot_rows) and min(hot_rows) > 0
    for k in range(1, n_keys):
        assert sum(int((g["key"] == k).any()) for g in got) <= 1, "a key that hashes landed on two ranks"
