"""tmp_wire_encode / tmp_wire_decode against the torch expressions of dist.WirePlan: the same wire
bytes column for column, and the table back bit for bit (the Exchange of partitionBy(partitionCols),
python/tempo/tsdf.py:117-121, moves these bytes)."""
import os

import pytest
import torch
import torch.distributed as dist

from tempo_amd import dist as tdist
from tempo_amd import ops

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def one_rank_group():
    """plan_wire reduces the column ranges over the group: a group of this one process."""
    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29641")
        dist.init_process_group("gloo", rank=0, world_size=1)
        created = True
    yield
    if created:
        dist.destroy_process_group()


def _table(n, n_keys, spans, n_masks, seed, dev):
    g = torch.Generator().manual_seed(seed)
    key = torch.randint(0, n_keys, (n,), generator=g, dtype=torch.int32)
    cols = []
    for s in spans:
        lo = -(1 << 40) + 12345
        cols.append(torch.randint(0, s, (n,), generator=g, dtype=torch.int64) + lo)
        if n:
            cols[-1][0], cols[-1][-1] = lo, lo + s - 1
    for _ in range(n_masks):
        cols.append(torch.randint(0, 2, (n,), generator=g, dtype=torch.uint8))
    cols.append(torch.rand(n, generator=g, dtype=torch.float64))
    cols.append(torch.randint(-5, 5, (n,), generator=g, dtype=torch.int64) * (1 << 60))   # stays 8 bytes
    return key.to(dev), [c.to(dev) for c in cols]


def _both(plan, key, cols):
    plan.device_kernels = True
    dev_wire = plan.encode(key, cols)
    plan.device_kernels = False
    ref_wire = plan.encode(key, cols)
    assert len(dev_wire) == len(ref_wire)
    for a, b in zip(dev_wire, ref_wire):
        assert a.dtype == b.dtype and torch.equal(a, b)
    dtypes = [c.dtype for c in cols]
    ref_key, ref_cols = plan.decode(ref_wire, dtypes)
    plan.device_kernels = True
    got_key, got_cols = plan.decode(ref_wire, dtypes)
    assert torch.equal(got_key, ref_key) and torch.equal(got_key, key)
    for a, b, c in zip(got_cols, ref_cols, cols):
        assert a.dtype == c.dtype and torch.equal(a.view(torch.uint8), b.view(torch.uint8))
        assert torch.equal(a.view(torch.uint8), c.view(torch.uint8))


@pytest.mark.parametrize("n", [0, 1, 3, 4, 1027, 1 << 20])
@pytest.mark.parametrize("n_keys,n_masks,spans", [
    (100, 1, [1 << 20]),                       # one-byte key word, 4-byte time
    (100, 2, [(1 << 36) + 5, 1 << 31]),        # two-byte word (7 + 2 bits), 5- and 4-byte columns
    (40000, 3, [(1 << 47) - 1]),               # 16 + 3 bits: four-byte word, 6-byte column
    (3, 0, [1 << 33, 1 << 45, 7]),             # no flags
    (1 << 20, 8, []),                          # flags only
])
def test_wire_kernels_match_the_torch_expressions(n, n_keys, n_masks, spans):
    key, cols = _table(n, n_keys, spans, n_masks, 17 * n + n_keys, "cuda")
    plan = tdist.plan_wire(key, cols, n_keys)
    if n > 4:
        assert len(plan.mask_cols) == n_masks
        assert sorted(nb for _, nb in plan.int_cols.values())[:len(spans)] == sorted(
            4 if s <= 1 << 32 else (5 if s <= 1 << 40 else 6) for s in spans)
    _both(plan, key, cols)


def test_wire_kernels_take_columns_at_any_alignment():
    n = 5001
    key, cols = _table(n + 3, 1000, [1 << 34, 1 << 10], 2, 5, "cuda")
    plan = tdist.plan_wire(key, cols, 1000)
    for off in (1, 2, 3):
        _both(plan, key[off:off + n], [c[off:off + n] for c in cols])


def test_wire_call_rejects_what_it_cannot_hold():
    key = torch.zeros(8, dtype=torch.int32, device="cuda")
    word = torch.zeros(8, dtype=torch.uint8, device="cuda")
    masks = [torch.zeros(8, dtype=torch.uint8, device="cuda") for _ in range(4)]
    with pytest.raises(ValueError, match="too narrow"):
        ops.wire_code(True, key, 6, word, masks, [], [], [], [], [])
    ints = [torch.zeros(8, dtype=torch.int64, device="cuda")]
    lo = [torch.zeros(8, dtype=torch.int32, device="cuda")]
    with pytest.raises(ValueError, match="4, 5 or 6 bytes"):
        ops.wire_code(True, key, 6, word, [], ints, [0], [7], lo, [None])
    with pytest.raises(ValueError, match="upper part missing"):
        ops.wire_code(True, key, 6, word, [], ints, [0], [5], lo, [None])


def test_tables_wider_than_one_call_take_the_torch_expressions():
    """More than eight flag columns or eight narrowed integer columns: the plan falls back to the
    elementwise form and the table still comes back bit for bit."""
    n = 3000
    g = torch.Generator().manual_seed(3)
    key = torch.randint(0, 50, (n,), generator=g, dtype=torch.int32).cuda()
    cols = [torch.randint(0, 1 << 20, (n,), generator=g, dtype=torch.int64).cuda() for _ in range(9)]
    cols += [torch.randint(0, 2, (n,), generator=g, dtype=torch.uint8).cuda() for _ in range(10)]
    plan = tdist.plan_wire(key, cols, 50)
    assert len(plan.mask_cols) == 10 and len(plan.int_cols) == 9
    assert not plan._on_device(key)
    wire = plan.encode(key, cols)
    back_key, back = plan.decode(wire, [c.dtype for c in cols])
    assert torch.equal(back_key, key)
    for a, b in zip(back, cols):
        assert a.dtype == b.dtype and torch.equal(a, b)
