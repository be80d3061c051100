import sys, numpy as np, torch
sys.path.insert(0, '.')
from tempo_amd import ops
from oracle import oracle as orc
BASE = 1_596_240_000_000_000
rng = np.random.default_rng(0)
for n in (10, 100, 1000, 5000, 70000):
    key = rng.integers(0, 5, n).astype(np.int32)
    ts = BASE + rng.integers(0, 240, n) * 250_000
    perm = ops.sort_order(torch.from_numpy(key).cuda(), torch.from_numpy(ts).cuda(), 5).cpu().numpy()
    want = np.lexsort((np.arange(n), ts, key))
    ok = np.array_equal(perm, want)
    print("sort n", n, "ok", ok)
    if not ok:
        bad = np.nonzero(perm != want)[0]
        print(" first bad", bad[:10], perm[bad[:10]], want[bad[:10]])
        print(" is perm a permutation:", np.array_equal(np.sort(perm), np.arange(n)))
        sk = key[perm]; st = ts[perm]
        print(" sorted by key?", (np.diff(sk) >= 0).all(), " sorted overall?", (np.diff(sk.astype(np.int64) * (1 << 40) + (st - BASE)) >= 0).all())
