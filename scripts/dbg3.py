import sys, numpy as np, torch
sys.path.insert(0, '.')
from tempo_amd import ops
from oracle import oracle as orc
BASE = 1_596_240_000_000_000
rng = np.random.default_rng(0)
for n in (1500, 2000, 3000, 4096, 4097, 10000):
    key = rng.integers(0, 5, n).astype(np.int32)
    ts = BASE + rng.integers(0, 240, n) * 250_000
    perm = ops.sort_order(torch.from_numpy(key).cuda(), torch.from_numpy(ts).cuda(), 5).cpu().numpy()
    want = np.lexsort((np.arange(n), ts, key))
    print("sort n", n, "ok", np.array_equal(perm, want), "isperm", np.array_equal(np.sort(perm), np.arange(n)))
for srt in (True, False):
  for (nL, nR, nk) in ((1000, 2000, 5), (1100, 500, 5), (3000, 500, 5)):
    key_l = rng.integers(0, nk, nL).astype(np.int32); key_r = rng.integers(0, nk, nR).astype(np.int32)
    ts_l = BASE + rng.integers(0, 240, nL) * 250_000; ts_r = BASE + rng.integers(0, 240, nR) * 250_000
    if srt:
        o = np.lexsort((ts_l, key_l)); key_l, ts_l = key_l[o], ts_l[o]
        o = np.lexsort((ts_r, key_r)); key_r, ts_r = key_r[o], ts_r[o]
    row, midx, miss = ops.asof_join(torch.from_numpy(key_l).cuda(), torch.from_numpy(ts_l).cuda(), torch.from_numpy(key_r).cuda(), torch.from_numpy(ts_r).cuda(), nk)
    idx, om = orc.asof_join(key_l, ts_l, key_r, ts_r, valid_r=[None])
    got = row.cpu().numpy()
    bad = np.nonzero(got != idx[0])[0]
    print("sorted" if srt else "unsorted", nL, nR, nk, "nbad", len(bad), bad[:8], got[bad[:8]], idx[0][bad[:8]])
