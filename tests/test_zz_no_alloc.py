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


def test_static_scratch_partition_takes_no_device_memory():
    """Device memory as the driver counts it (hipMemGetInfo, which sees every hipMalloc, not only
    torch's) is the same before and after a call made with outputs that exist already; the first
    call on a device may resolve the scratch array's address, so one warm-up call comes first."""
    import ctypes as C
    from tempo_amd import _lib
    n, parts = 400_000, 8
    key = (torch.arange(n, dtype=torch.int32, device="cuda") * 7919) % 1000
    perm = torch.empty(n, dtype=torch.int32, device="cuda")
    counts = torch.zeros(parts, dtype=torch.int64, device="cuda")
    a = _lib.PartitionArgs()
    a.n, a.key, a.n_parts, a.out_perm, a.out_counts = n, key.data_ptr(), parts, perm.data_ptr(), counts.data_ptr()
    stream = torch.cuda.current_stream().cuda_stream
    _lib.check(_lib.lib().tmp_hash_partition_static(C.byref(a), stream), "warm-up")
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    for _ in range(3):
        _lib.check(_lib.lib().tmp_hash_partition_static(C.byref(a), stream), "tmp_hash_partition_static")
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free_before
    assert int(counts.sum()) == n
    # argument errors come under the code tmp_hash_partition gives them, before any capacity test
    a.n = -1
    assert _lib.lib().tmp_hash_partition_static(C.byref(a), stream) == _lib.TMP_EINVAL
    ws = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
    assert _lib.lib().tmp_hash_partition(C.byref(a), ws.data_ptr(), ws.numel(), stream) == _lib.TMP_EINVAL
