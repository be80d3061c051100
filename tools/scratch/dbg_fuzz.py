"""Replay of two fuzz cases with diagnostics (scratch)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import fuzz_ops as fz
from tempo_amd import _lib, ops

def regen(seed):
    rng = np.random.default_rng(seed)
    rng.choice(len(fz.CASES), p=np.array([c[2] for c in fz.CASES], float) / sum(c[2] for c in fz.CASES))
    n_keys = int(rng.choice([1, 3, 40, 1000]))
    n = fz.draw_size(rng)
    key, ts, shape = fz.draw_table(rng, n, n_keys)
    w = int(rng.choice([0, 1, 10, 600, 5000]))
    is_int = rng.random() < 0.25
    x = rng.integers(-10**9, 10**9, n) if is_int else fz.draw_metric(rng, n)
    xv = fz.draw_mask(rng, n)
    return n_keys, n, key, ts, shape, w, is_int, x, xv

def run(seed):
    n_keys, n, key, ts, shape, w, is_int, x, xv = regen(seed)
    print("case", seed, n, shape, n_keys, w, is_int, "mask" if xv is not None else "nomask",
          "nulls", None if xv is None else int((xv == 0).sum()), "nonfinite", 0 if is_int else int((~np.isfinite(x)).sum()),
          "ts span s", (ts.max() - ts.min()) / 1e6, "distinct secs", len(np.unique(ts // 1_000_000)))
    _lib.profile_enable(True); _lib.profile_reset()
    r = ops.range_stats(fz.dev(key), fz.dev(ts), fz.dev(x), fz.dev(xv), n_keys, w)
    torch.cuda.synchronize()
    names = {}
    for nm, _, ms in _lib.profile_read():
        names[nm] = names.get(nm, 0) + 1
    print("kernels", names)
    _lib.profile_enable(False)
    # same table handed over in (key, ts) order: stable sort on the host first
    o = np.lexsort((np.arange(n), ts // 1_000_000, key))
    r2 = ops.range_stats(fz.dev(key[o]), fz.dev(ts[o]), fz.dev(x[o]), fz.dev(None if xv is None else xv[o]), n_keys, w)
    e = fz.orc.range_stats(fz.host(key), fz.host(ts), fz.host(x), fz.host(xv), n_keys, w)
    perm = r["perm"].cpu().numpy()
    print("perm equal oracle", np.array_equal(perm, e["perm"].numpy()), "perm equal host order", np.array_equal(perm, o))
    for name in ("count", "min", "max", "mean", "sum", "stddev"):
        g, g2, want = r[name].cpu().numpy(), r2[name].cpu().numpy(), e[name].numpy()
        v = e["valid"].numpy()
        ok = (v & (2 if name == "stddev" else 1)) != 0
        with np.errstate(all="ignore"):
            bad = ok & ~((g == want) | (np.abs(g - want) <= 1e-9 * np.abs(want) + 1e-6) | (np.isnan(g) & np.isnan(want)))
            bad2 = ok & ~((g2 == want) | (np.abs(g2 - want) <= 1e-9 * np.abs(want) + 1e-6) | (np.isnan(g2) & np.isnan(want)))
        print(name, "bad", int(bad.sum()), "bad when presorted", int(bad2.sum()))
        if bad.any():
            at = np.flatnonzero(bad)
            print("  first bad sorted positions", at[:8], "last", at[-3:])
            for i in at[:5]:
                print("   pos", i, "row", perm[i], "key", key[perm[i]], "ts", ts[perm[i]], "count", e["count"].numpy()[i], "got", g[i], "presorted", g2[i], "want", want[i])
    if name_bad := [nm for nm in ("stddev",) if True]:
        g, want = r["stddev"].cpu().numpy(), e["stddev"].numpy()
        v = e["valid"].numpy()
        with np.errstate(all="ignore"):
            bad = ((v & 2) != 0) & ~(np.abs(g - want) <= 1e-9 * np.abs(want) + 1e-12)
        sec = ts // 1_000_000
        for i in np.flatnonzero(bad)[[0, len(np.flatnonzero(bad)) // 2, -1]] if bad.any() else []:
            row = perm[i]
            m = (key == key[row]) & (sec >= sec[row] - w) & (sec <= sec[row])
            if xv is not None:
                m &= xv != 0
            vals = x[m].astype(np.longdouble)
            exact = float(np.sqrt(((vals - vals.mean()) ** 2).sum() / (len(vals) - 1)))
            print("   exact check pos", i, "frame valid rows", int(m.sum()), "exact", exact, "gpu", g[i], "oracle", want[i],
                  "gpu rel err", abs(g[i] - exact) / exact, "oracle rel err", abs(want[i] - exact) / exact,
                  "x level", float(vals.mean()), "x kind sample", x[:3])
    cnt = e["count"].numpy()
    print("frame rows: max", cnt.max(), "median", np.median(cnt))

for s in sys.argv[1:]:
    run(int(s))
