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
