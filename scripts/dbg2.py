import sys, numpy as np, torch
sys.path.insert(0, '.')
from tempo_amd import ops
from oracle import oracle as orc
BASE = 1_596_240_000_000_000
rng = np.random.default_rng(0)
for (nL, nR, nk) in ((8, 8, 1), (8, 8, 2), (300, 200, 3), (3000, 2000, 5)):
    key_l = rng.integers(0, nk, nL).astype(np.int32); key_r = rng.integers(0, nk, nR).astype(np.int32)
    ts_l = BASE + rng.integers(0, 240, nL) * 250_000; ts_r = BASE + rng.integers(0, 240, nR) * 250_000
    row, midx, miss = ops.asof_join(torch.from_numpy(key_l).cuda(), torch.from_numpy(ts_l).cuda(), torch.from_numpy(key_r).cuda(), torch.from_numpy(ts_r).cuda(), nk)
    idx, om = orc.asof_join(key_l, ts_l, key_r, ts_r, valid_r=[None])
    got = row.cpu().numpy()
    print(nL, nR, nk, "equal", np.array_equal(got, idx[0]), "nbad", (got != idx[0]).sum())
    if nL <= 8:
        print(" keyL", key_l, "\n tsL", (ts_l-BASE)//250000, "\n keyR", key_r, "\n tsR", (ts_r-BASE)//250000, "\n got", got, "\n want", idx[0])
