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


def host_partition(key, n_parts):
    owner = np.array([_lib.lib().tmp_key_owner(int(k), n_parts) for k in key.numpy()], dtype=np.int64)
    perm = np.argsort(owner, kind="stable")
    return torch.from_numpy(perm.astype(np.int32)), torch.from_numpy(np.bincount(owner, minlength=n_parts))


def host_take(col, perm):
    return col[perm.to(torch.int64)]


def main():
    out_dir = sys.argv[1]
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
    owners = {_lib.lib().tmp_key_owner(int(k), world) for k in k_loc.numpy()}
    assert owners <= {rank}, "a rank received a key it does not own"
    r = oracle.range_stats(k_loc.numpy(), ts_loc.numpy(), x_loc.numpy(), None, 30)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), rid=rid_loc.numpy()[r["perm"]], mean=r["mean"],
             count=r["count"], stddev=r["stddev"], mx=r["max"])
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
