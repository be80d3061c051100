"""World-size-2 test of the key shuffle on CPU (gloo): shard by key owner, run the operator per
rank, and the union of the ranks' results must equal the single-process result row for row."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


import pytest


@pytest.mark.parametrize("max_message_bytes", [None, 1000])
def test_key_shuffle_two_ranks(tmp_path, orc, max_message_bytes):
    """max_message_bytes=1000: every column goes out in many rounds (125 rows of an 8-byte
    column per peer and round), the path a 10^9-row column takes at the default 1 GiB limit."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    if max_message_bytes:
        env["TEMPO_TEST_MAX_MESSAGE_BYTES"] = str(max_message_bytes)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path)]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    rng = np.random.default_rng(1234)
    n, nkeys = 4000, 37
    base = 1_596_240_000_000_000
    key = rng.integers(0, nkeys, n).astype(np.int32)
    ts = base + rng.integers(0, 600, n) * 1_000_000
    x = rng.normal(100, 3, n)
    ref = orc.range_stats(key, ts, x, None, 30)
    want = {int(r): (ref["mean"][p_], ref["count"][p_], ref["stddev"][p_], ref["max"][p_])
            for p_, r in enumerate(ref["perm"])}
    seen = 0
    for rank in range(2):
        d = np.load(tmp_path / f"rank{rank}.npz")
        for i, r in enumerate(d["rid"]):
            m, c, s, mx = want[int(r)]
            assert d["count"][i] == c and d["mx"][i] == mx
            assert abs(d["mean"][i] - m) <= 1e-9 * abs(m) and abs(d["stddev"][i] - s) <= 1e-9 * max(abs(s), 1e-12)
            seen += 1
    assert seen == n
    # sliced hot key: union of the ranks' skew joins == the single-process join, by global row id
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from dist_worker import join_tables
    kl, tl, kr, tr, va = join_tables()
    ref, _ = orc.asof_join(kl, tl, kr, tr, valid_r=[None, va], ts_partition_val=10, fraction=0.1)
    got = np.full_like(ref, -2)
    for rank in range(2):
        d = np.load(tmp_path / f"join{rank}.npz")
        assert (got[:, d["left"]] == -2).all(), "a left row was joined on two ranks"
        got[:, d["left"]] = d["right"]
    assert np.array_equal(got, ref)
    # one global series cut into time ranges (halo rows, carried tails, carried last right rows)
    import json
    parts = [json.load(open(tmp_path / f"timesplit{rank}.json")) for rank in range(2)]
    for op in ("ema", "join", "range_stats"):
        want = parts[0][op][1]
        have = {}
        for p_ in parts:
            assert not (have.keys() & p_[op][0].keys()), "a row came out of two ranks"
            have.update(p_[op][0])
        assert len(want) == 1500 and all(len(p_[op][0]) > 100 for p_ in parts), "every rank owns a time range"
        if op == "range_stats":
            import timesplit_cases as tc
            assert have.keys() == want.keys()
            tc.compare_range_stats({k: tuple(v) for k, v in have.items()}, {k: tuple(v) for k, v in want.items()})
        else:
            assert have == want, op


def test_assign_owners_balances_and_is_deterministic():
    from tempo_amd.dist import assign_owners, plan_owners
    rng = np.random.default_rng(3)
    w = 1.0 / np.arange(1, 10_001) ** 1.2
    counts = rng.multinomial(30_000_000, w / w.sum())
    for world in (2, 4):
        own = assign_owners(counts, world)
        load = np.bincount(own, weights=counts, minlength=world)
        assert load.max() <= 1.01 * load.mean()
        assert np.array_equal(own, assign_owners(counts.copy(), world))
    # 100 equal keys over 8 ranks: 12 or 13 keys each
    own = assign_owners(np.full(100, 1000), 8)
    assert sorted(np.bincount(own, minlength=8).tolist()) == [12] * 4 + [13] * 4
    # a key heavier than a rank's share: whole without skew, sliced with it
    import torch
    plain = plan_owners(counts, 8, torch.device("cpu"))
    assert plain.slices is None and (plain.table.numpy() >= 0).all()
    base = 1_596_240_000_000_000
    plan = plan_owners(counts, 8, torch.device("cpu"), skew=dict(ts_partition_val=10, fraction=0.1),
                       ts_range=(base, base + 86_400_000_000))
    assert plan.slices is not None and plan.table[0].item() == -1
    sl = plan.slices
    n_sl = sl["split_slices"].numpy()
    items = counts.astype(np.float64)
    load = np.zeros(8)
    tab = plan.table.numpy()
    np.add.at(load, tab[tab >= 0], items[tab >= 0])
    for s_, k in enumerate(np.flatnonzero(tab < 0)):
        assert tab[k] == -(s_ + 1)
        owners = sl["slice_owner"].numpy()[sl["split_first"].numpy()[s_]:][:n_sl[s_]]
        np.add.at(load, owners, items[k] / n_sl[s_])
    assert load.max() <= 1.02 * load.mean()
    # wrap-around of the int32 product v * q switches slicing off
    far = plan_owners(counts, 8, torch.device("cpu"), skew=dict(ts_partition_val=10, fraction=0.1),
                      ts_range=(base, 3_000_000_000 * 1_000_000))
    assert far.slices is None
