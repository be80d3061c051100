import sys, numpy as np, torch
sys.path.insert(0, '.')
from tempo_amd import ops
from oracle import oracle as orc
BASE = 1_596_240_000_000_000
rng = np.random.default_rng(0)
def runs(b):
    out=[]; s=None; p=None
    for x in b:
        if s is None: s=p=x
        elif x==p+1: p=x
        else: out.append((s,p)); s=p=x
    if s is not None: out.append((s,p))
    return out
for (nL, nR, nk) in ((1100, 500, 5), (1100, 500, 1), (2048, 3000, 2)):
    key_l = rng.integers(0, nk, nL).astype(np.int32); key_r = rng.integers(0, nk, nR).astype(np.int32)
    ts_l = BASE + rng.integers(0, 240, nL) * 250_000; ts_r = BASE + rng.integers(0, 240, nR) * 250_000
    o = np.lexsort((ts_l, key_l)); key_l, ts_l = key_l[o], ts_l[o]
    o = np.lexsort((ts_r, key_r)); key_r, ts_r = key_r[o], ts_r[o]
    row, midx, miss = ops.asof_join(torch.from_numpy(key_l).cuda(), torch.from_numpy(ts_l).cuda(), torch.from_numpy(key_r).cuda(), torch.from_numpy(ts_r).cuda(), nk)
    idx, om = orc.asof_join(key_l, ts_l, key_r, ts_r, valid_r=[None])
    got = row.cpu().numpy()
    bad = np.nonzero(got != idx[0])[0]
    print(nL, nR, nk, "nbad", len(bad), "runs", runs(bad.tolist())[:20])
    for b in bad[:3]:
        print("   i", b, "keyL", key_l[b], "ts", (ts_l[b]-BASE)//250000, "got", got[b], "want", idx[0][b], "wantkey", key_r[idx[0][b]], (ts_r[idx[0][b]]-BASE)//250000)
