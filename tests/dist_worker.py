"""Rank body of the 2-process gloo test: shuffle rows by key owner, run the local operator,
write what this rank produced.  The device kernels are replaced by host equivalents that use
the library's own owner function (tmp_key_owner), so the exchange logic is the product's."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import oracle  # noqa: E402
from tempo_amd import _lib  # noqa: E402
from tempo_amd.dist import shuffle_by_key  # noqa: E402


from host_partition import host_counts, host_partition, host_take  # noqa: E402


def join_tables():
    """Global trades / quotes of the sliced-key scenario: key 0 holds ~70 % of the rows."""
    rng = np.random.default_rng(77)
    n = 3000
    base = 1_596_240_000_000_000
    hot = lambda: np.where(rng.random(n) < 0.7, 0, rng.integers(1, 6, n)).astype(np.int32)
    kl, kr = hot(), hot()
    tl = base + rng.integers(0, 600_000, n) * 1000
    tr = base + rng.integers(0, 600_000, n) * 1000
    va = (rng.random(n) > 0.2).astype(np.uint8)
    return kl, tl, kr, tr, va


def join_scenario(out_dir, rank, world):
    from tempo_amd.dist import balanced_owners
    kl, tl, kr, tr, va = join_tables()
    n = len(kl)
    sl = slice(rank * n // world, (rank + 1) * n // world)
    T = torch.from_numpy
    rid = np.arange(n, dtype=np.int64)
    skew = dict(ts_partition_val=10, fraction=0.1, ts_is_seconds=False)
    plan = balanced_owners([T(kl[sl]), T(kr[sl])], 6, stride=1, counts_fn=host_counts, skew=skew,
                           ts=[T(tl[sl]), T(tr[sl])])
    assert plan.slices is not None and int(plan.table[0]) == -1, "the hot key was not sliced"
    k1, (t1, r1), _, _ = shuffle_by_key(T(kl[sl]), [T(tl[sl]), T(rid[sl])], partition_fn=host_partition,
                                        take_fn=host_take, owners=plan, ts=T(tl[sl]))
    k2, (t2, v2, r2), _, _ = shuffle_by_key(T(kr[sl]), [T(tr[sl]), T(va[sl]), T(rid[sl])],
                                            partition_fn=host_partition, take_fn=host_take, owners=plan,
                                            ts=T(tr[sl]), halo=True)
    assert (k1 == 0).any(), "every rank should own a slice of the hot key"
    # narrowed columns (the NULL mask folded into the key byte) through the sliced, halo-carrying route
    from tempo_amd.dist import plan_wire
    qcols = [T(tr[sl]), T(va[sl]), T(rid[sl])]
    wire = plan_wire(T(kr[sl]), qcols, 6)
    assert wire.mask_cols == [1] and wire.bytes_per_row(qcols) == 1 + 4 + 4
    k2w, (t2w, v2w, r2w), _, _ = shuffle_by_key(T(kr[sl]), qcols, partition_fn=host_partition, take_fn=host_take,
                                                owners=plan, ts=T(tr[sl]), halo=True, wire=wire, verify=True)
    assert torch.equal(k2w, k2) and torch.equal(t2w, t2) and torch.equal(v2w, v2) and torch.equal(r2w, r2)
    idx, _ = oracle.asof_join(k1.numpy(), t1.numpy(), k2.numpy(), t2.numpy(), valid_r=[None, v2.numpy()],
                              ts_partition_val=10, fraction=0.1)
    r2n = r2.numpy()
    np.savez(os.path.join(out_dir, f"join{rank}.npz"), left=r1.numpy(),
             right=np.where(idx >= 0, r2n[np.maximum(idx, 0)], -1))


def time_split_scenario(out_dir, rank, world):
    """A table without partition columns (one global series) over time ranges, through the real
    exchange and all-gather: every rank writes the rows it produced, rank 0 also what the
    operator gives on the whole table."""
    import json

    import oracle_ops
    import timesplit_cases as tc
    from tempo_amd import timesplit
    comm = timesplit.GroupComm(partition_fn=host_partition, take_fn=host_take)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    out = {}
    have, want = tc.check_ema(comm, oracle_ops, T, n=1500, n_keys=1, window=9)
    out["ema"] = [have, want]
    have, want = tc.check_join(comm, oracle_ops, T, n=1500, n_keys=2, skip_nulls=True)
    out["join"] = [have, want]
    have, want = tc.check_range_stats(comm, oracle_ops, T, n=1500, n_keys=1, window=45)
    out["range_stats"] = [have, want]
    if rank != 0:
        for v in out.values():
            v[1] = {}
    with open(os.path.join(out_dir, f"timesplit{rank}.json"), "w") as f:
        json.dump(out, f)


def main():
    out_dir = sys.argv[1]
    if os.environ.get("TEMPO_TEST_MAX_MESSAGE_BYTES"):  # force the exchange into many small rounds
        import tempo_amd.dist as tdist
        tdist.MAX_MESSAGE_BYTES = int(os.environ["TEMPO_TEST_MAX_MESSAGE_BYTES"])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    rng = np.random.default_rng(1234)  # every rank draws the same global table and keeps its slice
    n, nkeys = 4000, 37
    base = 1_596_240_000_000_000
    key = rng.integers(0, nkeys, n).astype(np.int32)
    ts = base + rng.integers(0, 600, n) * 1_000_000
    x = rng.normal(100, 3, n)
    rid = np.arange(n, dtype=np.int64)
    sl = slice(rank * n // world, (rank + 1) * n // world)
    k_loc, (ts_loc, x_loc, rid_loc), send, recv = shuffle_by_key(
        torch.from_numpy(key[sl]), [torch.from_numpy(ts[sl]), torch.from_numpy(x[sl]), torch.from_numpy(rid[sl])],
        partition_fn=host_partition, take_fn=host_take)
    assert sum(recv) == k_loc.numel()
    # the same shuffle with the columns narrowed for the trip (key in one byte, the timestamps and row
    # ids as 4-byte offsets) and every exchange checked by checksum: the very same tensors arrive
    from tempo_amd.dist import plan_wire
    cols = [torch.from_numpy(ts[sl]), torch.from_numpy(x[sl]), torch.from_numpy(rid[sl])]
    wire = plan_wire(torch.from_numpy(key[sl]), cols, nkeys)
    assert wire.bytes_per_row(cols) == 1 + 4 + 8 + 4, wire.bytes_per_row(cols)
    k_w, (ts_w, x_w, rid_w), send_w, recv_w = shuffle_by_key(torch.from_numpy(key[sl]), cols, partition_fn=host_partition,
                                                             take_fn=host_take, wire=wire, verify=True)
    assert torch.equal(k_w, k_loc) and torch.equal(ts_w, ts_loc) and torch.equal(x_w, x_loc) and torch.equal(rid_w, rid_loc)
    assert send_w == send and recv_w == recv and ts_w.dtype == torch.int64 and k_w.dtype == torch.int32
    # a key field wider than a byte travels as a 2-byte word (sent as raw bytes: the collectives have
    # no 16-bit integer type)
    wire16 = plan_wire(torch.from_numpy(key[sl]), cols, 5000)
    assert wire16.key_dtype == torch.int16
    k_h, (ts_h, x_h, rid_h), _, _ = shuffle_by_key(torch.from_numpy(key[sl]), cols, partition_fn=host_partition,
                                                   take_fn=host_take, wire=wire16, verify=True)
    assert torch.equal(k_h, k_loc) and torch.equal(ts_h, ts_loc) and torch.equal(x_h, x_loc) and torch.equal(rid_h, rid_loc)
    owners = {_lib.lib().tmp_key_owner(int(k), world) for k in k_loc.numpy()}
    assert owners <= {rank}, "a rank received a key it does not own"
    r = oracle.range_stats(k_loc.numpy(), ts_loc.numpy(), x_loc.numpy(), None, 30)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), rid=rid_loc.numpy()[r["perm"]], mean=r["mean"],
             count=r["count"], stddev=r["stddev"], mx=r["max"])
    join_scenario(out_dir, rank, world)
    time_split_scenario(out_dir, rank, world)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
