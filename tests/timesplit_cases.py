"""The four operators over time ranges against the same operator on the whole table, for any
`comm` (simulated ranks or a process group) and any `lops` (the HIP ops on a GPU, the oracle
stand-in on the CPU).  TEST INFRASTRUCTURE shared by the CPU and GPU suites."""
import numpy as np
import torch

from tempo_amd import timesplit

BASE = 1_596_240_000_000_000


HOT_SHARE = 0.0   # > 0: key 0 holds that share of the rows (a hot key next to a tail of small ones)


def make_series(seed, n, n_keys, span_s=900):
    rng = np.random.default_rng(seed)
    key = rng.integers(0, n_keys, n).astype(np.int32)
    if HOT_SHARE:
        key = np.where(rng.random(n) < HOT_SHARE, 0, key).astype(np.int32)
    ts = BASE + rng.integers(0, span_s * 1000, n) * 1000           # milliseconds: many rows per second
    x = rng.normal(50, 5, n)
    xv = (rng.random(n) > 0.15).astype(np.uint8)
    seq = rng.integers(0, 5, n)
    sv = (rng.random(n) > 0.2).astype(np.uint8)
    return dict(key=key, ts=ts, x=x, xv=xv, seq=seq, sv=sv, rid=np.arange(n, dtype=np.int64))


def chunks(n, world):
    return [slice(r * n // world, (r + 1) * n // world) for r in range(world)]


def _by_rid(rid, *cols):
    o = np.argsort(rid, kind="stable")
    return [np.asarray(c)[o] for c in cols]


def check_range_stats(comm, lops, T, n=6000, n_keys=1, window=20, seed=1, **kw):
    """T(array) -> tensor where the operator wants it.  comm.local names the ranks played here;
    returns {rid: (count, mean, max, valid)} for them and the expected dict for all rows."""
    s = make_series(seed, n, n_keys)
    use_key = n_keys > 1
    tables = [dict(key=T(s["key"][sl]) if use_key else None, ts=T(s["ts"][sl]), x=T(s["x"][sl]), x_valid=T(s["xv"][sl]),
                   extra=[T(s["rid"][sl])]) for sl in [chunks(n, comm.world)[r] for r in comm.local]]
    got = timesplit.range_stats_by_time(comm, tables, n_keys, window, local_ops=lops, **kw)
    ref = lops.range_stats(T(s["key"]) if use_key else T(np.zeros(n, np.int32)), T(s["ts"]), T(s["x"]), T(s["xv"]),
                           n_keys, window)
    rp = ref["perm"].cpu().numpy()
    want = {int(r): (int(c), float(m), float(mx), int(v)) for r, c, m, mx, v in
            zip(rp, ref["count"].cpu().numpy(), ref["mean"].cpu().numpy(), ref["max"].cpu().numpy(),
                ref["valid"].cpu().numpy())}
    have = {}
    for g in got:
        for r, c, m, mx, v in zip(g["extra"][0].cpu().numpy(), g["count"].cpu().numpy(), g["mean"].cpu().numpy(),
                                  g["max"].cpu().numpy(), g["valid"].cpu().numpy()):
            assert int(r) not in have, "a row came out of two ranks"
            have[int(r)] = (int(c), float(m), float(mx), int(v))
    return have, want


def compare_range_stats(have, want):
    for r, (c, m, mx, v) in have.items():
        wc, wm, wmx, wv = want[r]
        assert (c, v) == (wc, wv), (r, c, wc, v, wv)
        if v & 1:
            assert mx == wmx and abs(m - wm) <= 1e-9 * max(1.0, abs(wm)), (r, m, wm)


def check_ema(comm, lops, T, n=5000, n_keys=1, window=7, seed=2, with_seq=False, **kw):
    s = make_series(seed, n, n_keys)
    use_key = n_keys > 1
    sq = lambda sl: dict(seq=T(s["seq"][sl]), seq_valid=T(s["sv"][sl])) if with_seq else {}
    tables = [dict(key=T(s["key"][sl]) if use_key else None, ts=T(s["ts"][sl]), x=T(s["x"][sl]), x_valid=T(s["xv"][sl]),
                   extra=[T(s["rid"][sl])], **sq(sl)) for sl in [chunks(n, comm.world)[r] for r in comm.local]]
    got = timesplit.ema_by_time(comm, tables, n_keys, window, 0.3, local_ops=lops, **kw)
    full = slice(0, n)
    perm, ref = lops.ema(T(s["key"]) if use_key else T(np.zeros(n, np.int32)), T(s["ts"]), T(s["x"]), T(s["xv"]), n_keys,
                         window, 0.3, **sq(full))
    want = dict(zip(perm.cpu().numpy().tolist(), ref.cpu().numpy().tolist()))
    have = {}
    for g in got:
        for r, e in zip(g["extra"][0].cpu().numpy().tolist(), g["ema"].cpu().numpy().tolist()):
            assert r not in have
            have[r] = e
    return have, want


def check_lookback(comm, lops, T, n=4000, n_keys=1, size=5, seed=3, exact=True, **kw):
    s = make_series(seed, n, n_keys)
    use_key = n_keys > 1
    y = s["x"] * 2 + 1
    tables = [dict(key=T(s["key"][sl]) if use_key else None, ts=T(s["ts"][sl]), xs=[T(s["x"][sl]), T(y[sl])],
                   xs_valid=[T(s["xv"][sl]), None], extra=[T(s["rid"][sl])])
              for sl in [chunks(n, comm.world)[r] for r in comm.local]]
    got = timesplit.lookback_by_time(comm, tables, n_keys, size, exact, local_ops=lops, **kw)
    rows, lens, feat, fv = lops.lookback(T(s["key"]) if use_key else T(np.zeros(n, np.int32)), T(s["ts"]),
                                         [T(s["x"]), T(y)], [T(s["xv"]), None], n_keys, size, exact)
    fvn = None if fv is None else fv.cpu().numpy()
    want = {int(r): (feat[i].cpu().numpy(), None if fvn is None else fvn[i], None if lens is None else int(lens[i]))
            for i, r in enumerate(rows.cpu().numpy())}
    have = {}
    for g in got:
        f = g["feat"].cpu().numpy()
        v = None if g["feat_valid"] is None else g["feat_valid"].cpu().numpy()
        for i, r in enumerate(g["extra"][0].cpu().numpy()):
            assert int(r) not in have
            have[int(r)] = (f[i], None if v is None else v[i], None if g["lengths"] is None else int(g["lengths"][i]))
    return have, want


def compare_lookback(have, want):
    assert have.keys() == want.keys()
    for r, (f, v, ln) in have.items():
        wf, wv, wl = want[r]
        assert ln == wl
        if v is None:
            assert np.array_equal(f, wf)
        else:
            assert np.array_equal(v, wv) and np.array_equal(np.where(v != 0, f, 0), np.where(wv != 0, wf, 0))


def check_join(comm, lops, T, n=5000, n_keys=1, seed=4, skip_nulls=True, with_seq=False, **kw):
    rng = np.random.default_rng(seed)
    L, R = make_series(seed, n, n_keys, 600), make_series(seed + 100, n, n_keys, 600)
    # long NULL stretches, so that the last non-NULL value of a column lies ranges back
    bid = rng.normal(10, 1, n)
    bv = (rng.random(n) > 0.97).astype(np.uint8)
    ask = rng.integers(0, 1000, n)
    av = (rng.random(n) > 0.5).astype(np.uint8)
    use_key = n_keys > 1
    K = lambda a, sl: T(a[sl]) if use_key else None
    sq = lambda sl: dict(seq=T(R["seq"][sl]), seq_valid=T(R["sv"][sl])) if with_seq else {}
    lefts, rights = [], []
    for sl in [chunks(n, comm.world)[r] for r in comm.local]:
        lefts.append(dict(key=K(L["key"], sl), ts=T(L["ts"][sl]), cols=[T(L["rid"][sl])]))
        m_b, m_a = T(bv[sl]), T(av[sl])
        rights.append(dict(key=K(R["key"], sl), ts=T(R["ts"][sl]), cols=[T(R["ts"][sl]), T(bid[sl]), T(ask[sl]), T(R["rid"][sl])],
                           masks=[None, m_b, m_a, m_a], **sq(sl)))
    got = timesplit.asof_join_by_time(comm, lefts, rights, n_keys, skip_nulls, local_ops=lops, **kw)
    zk = T(np.zeros(n, np.int32))
    full = slice(0, n)
    m_b, m_a = T(bv), T(av)
    ref = lops.asof_join_table(T(L["key"]) if use_key else zk, T(L["ts"]), T(R["key"]) if use_key else zk, T(R["ts"]),
                               n_keys, [T(L["rid"])], [T(R["ts"]), T(bid), T(ask), T(R["rid"])], [None, m_b, m_a, m_a],
                               skip_nulls, **{("seq_r" if k == "seq" else "seq_r_valid"): v for k, v in sq(full).items()})

    def rows(j):
        out = {}
        rid = j["left"][0].cpu().numpy()
        cols = [(v.cpu().numpy(), ok.cpu().numpy()) for v, ok in j["right"]]
        for i, r in enumerate(rid):
            out[int(r)] = tuple((float(v[i]) if ok[i] else None) for v, ok in cols)
        return out

    want = rows(ref)
    have = {}
    for j in got:
        part = rows(j)
        assert not (part.keys() & have.keys()), "a left row was joined on two ranks"
        have.update(part)
    return have, want
