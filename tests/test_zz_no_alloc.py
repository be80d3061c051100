"""tmp_hash_partition_static (the entry point that allocates nothing: its scratch is a device
array inside the library) against tmp_hash_partition, through the C ABI: two calls in a row, the
second with a smaller input, so stale scratch contents of the first would show; bit-exact
(row permutation and per-part counts are integer results)."""
import ctypes as C

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _both(key_np, n_parts, table=None):
    from tempo_amd import ops
    key = torch.from_numpy(key_np).cuda()
    tab = None if table is None else torch.from_numpy(table).cuda()
    before = torch.cuda.memory_allocated()
    perm_s, cnt_s = ops.hash_partition(key, n_parts, tab, static_scratch=True)
    grown = torch.cuda.memory_allocated() - before
    perm_w, cnt_w = ops.hash_partition(key, n_parts, tab)
    torch.cuda.synchronize()
    # the static form allocated its two outputs and nothing else (no workspace)
    assert grown <= perm_s.numel() * 4 + cnt_s.numel() * 8 + 1024
    return perm_s.cpu().numpy(), cnt_s.cpu().numpy(), perm_w.cpu().numpy(), cnt_w.cpu().numpy()


def test_static_scratch_partition_matches_workspace_form():
    from tempo_amd import _lib
    rng = np.random.default_rng(11)
    n1, n2 = 900_001, 70_003          # second call smaller: it must not see the first one's leftovers
    for n, parts, n_keys in [(n1, 8, 5000), (n2, 3, 40)]:
        key = rng.integers(0, n_keys, n).astype(np.int32)
        ps, cs, pw, cw = _both(key, parts)
        assert np.array_equal(cs, cw) and int(cs.sum()) == n
        assert np.array_equal(ps, pw)
        owner = np.array([_lib.lib().tmp_key_owner(int(k), parts) for k in range(n_keys)])[key]
        assert np.array_equal(ps, np.argsort(owner, kind="stable"))
    # with an owner table, and an empty input
    key = rng.integers(0, 100, 50_000).astype(np.int32)
    table = rng.integers(0, 5, 100).astype(np.int32)
    ps, cs, pw, cw = _both(key, 5, table)
    assert np.array_equal(ps, pw) and np.array_equal(cs, cw)
    assert np.array_equal(ps, np.argsort(table[key], kind="stable"))
    ps, cs, pw, cw = _both(np.zeros(0, dtype=np.int32), 4)
    assert cs.tolist() == [0, 0, 0, 0] and len(ps) == 0


def test_static_scratch_partition_rejects_oversize():
    from tempo_amd import _lib, ops
    n = _lib.TMP_STATIC_PARTITION_MAX_ROWS + 1
    key = torch.zeros(n, dtype=torch.int32, device="cuda")
    with pytest.raises(_lib.TempoHipError, match="exceed the built-in scratch"):
        ops.hash_partition(key, 4, static_scratch=True)
    # exactly the maximum still works
    key = torch.arange(n - 1, dtype=torch.int32, device="cuda") % 7
    perm, cnt = ops.hash_partition(key, 4, static_scratch=True)
    perm_w, cnt_w = ops.hash_partition(key, 4)
    assert torch.equal(perm, perm_w) and torch.equal(cnt, cnt_w)
