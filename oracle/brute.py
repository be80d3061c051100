"""Definition-level pure-Python restatement of the TSDF hot path -- TEST INFRASTRUCTURE ONLY.

Quadratic loops that apply the window/visibility definitions row by row; used on tiny inputs to
cross-check the C oracle (which uses the reference's sort-and-scan formulation).  Sections of
the reference restated: python/tempo/tsdf.py:111-190 (asofJoin), :563-580 (window order),
:615-635 (EMA), :637-671 (lookback), :673-721 (range stats); python/tempo/resample.py:25-117.
"""
import math


def _trunc_i32(x):
    if x != x:
        return 0
    if x >= 2147483647.0:
        return 2147483647
    if x <= -2147483648.0:
        return -2147483648
    return int(x)


def _wrap_i32(v):
    return ((v + 2 ** 31) % 2 ** 32) - 2 ** 31


def time_partition(ts_us, v, ts_is_seconds=False):
    """(ts_partition, partition_remainder) as tsdf.py:178-180 computes them in Spark."""
    d = float(ts_us) if ts_is_seconds else ts_us / 1000000.0
    q = _trunc_i32(d / v)
    p = _wrap_i32(v * q) if isinstance(v, int) else v * q
    return p, (d - p) / v


def asof_join(key_l, ts_l, key_r, ts_r, seq_r=None, seq_r_valid=None, valid_r=(), skip_nulls=True,
              ts_is_seconds=False, ts_partition_val=None, fraction=0.5):
    nL, nR, C = len(key_l), len(key_r), len(valid_r)
    has_seq = seq_r is not None
    v = ts_partition_val

    def seq_key(j):
        if not has_seq or (seq_r_valid is not None and not seq_r_valid[j]):
            return (0, 0)  # NULLS FIRST
        return (1, int(seq_r[j]))

    out = [[-1] * nL for _ in range(C)]
    missing = [0] * C
    for i in range(nL):
        left_order = (int(ts_l[i]), (0, 0), 1)
        vis = []
        for j in range(nR):
            if key_r[j] != key_l[i]:
                continue
            order = (int(ts_r[j]), seq_key(j), -1)
            if not order < left_order:
                continue
            if v is not None:
                p_l, _ = time_partition(int(ts_l[i]), v, ts_is_seconds)
                p_r, r_r = time_partition(int(ts_r[j]), v, ts_is_seconds)
                nxt = _wrap_i32(p_r + v) if isinstance(v, int) else p_r + v
                if not (p_r == p_l or (nxt == p_l and r_r >= 1 - fraction)):
                    continue
            vis.append((order, j))
        vis.sort()
        for c in range(C):
            nn = [j for _, j in vis if valid_r[c] is None or valid_r[c][j]]
            if not nn:
                missing[c] = 1
            if skip_nulls:
                out[c][i] = nn[-1] if nn else -1
            else:
                out[c][i] = vis[-1][1] if vis else -1
    return out, missing


def _sec(ts, ts_is_seconds):
    return int(ts) if ts_is_seconds else int(ts) // 1000000


def window_order(key, ts, ts_is_seconds=False, seq=None, seq_valid=None):
    def k(i):
        s = (0, 0)
        if seq is not None:
            s = (0, 0) if (seq_valid is not None and not seq_valid[i]) else (1, int(seq[i]))
        return (int(key[i]) if key is not None else 0, _sec(ts[i], ts_is_seconds), s, i)
    return sorted(range(len(ts)), key=k)


def _lt(a, b):
    if a != a:
        return False
    if b != b:
        return True
    return a < b


def range_stats(key, ts, x, x_valid, W, ts_is_seconds=False):
    perm = window_order(key, ts, ts_is_seconds)
    rows = []
    for r in perm:
        s = _sec(ts[r], ts_is_seconds)
        frame = [q for q in perm
                 if (key is None or key[q] == key[r]) and s - W <= _sec(ts[q], ts_is_seconds) <= s]
        vals = [x[q] for q in frame if x_valid is None or x_valid[q]]
        n, avg, m2, tot = 0, 0.0, 0.0, 0 if isinstance(x[0] if len(x) else 0, int) else 0.0
        mn = mx = None
        for v in vals:
            n += 1
            tot = tot + v
            d = float(v) - avg
            dn = d / n
            avg += dn
            m2 += d * (d - dn)
            if mn is None:
                mn = mx = v
            else:
                if _lt(v, mn):
                    mn = v
                if _lt(mx, v):
                    mx = v
        fsum = math.fsum(float(v) for v in vals) if all(v == v and abs(v) != math.inf for v in vals) else None
        sd = math.sqrt(m2 / (n - 1)) if n > 1 else None
        mean = (sum(float(v) for v in vals) / n) if n else None
        xv = x_valid is None or x_valid[r]
        z = (float(x[r]) - mean) / sd if (xv and sd is not None and sd != 0.0) else None
        rows.append(dict(row=r, count=n, sum=tot if n else None, fsum=fsum, mean=mean, min=mn,
                         max=mx, stddev=sd, zscore=z))
    return rows


def ema(key, ts, x, x_valid, window, exp_factor, ts_is_seconds=False, seq=None, seq_valid=None):
    perm = window_order(key, ts, ts_is_seconds, seq, seq_valid)
    pos_in_series = {}
    series = {}
    for r in perm:
        k = int(key[r]) if key is not None else 0
        series.setdefault(k, []).append(r)
        pos_in_series[r] = len(series[k]) - 1
    out = []
    for r in perm:
        k = int(key[r]) if key is not None else 0
        acc = 0.0
        for i in range(window):
            j = pos_in_series[r] - i
            if j < 0:
                continue
            q = series[k][j]
            if x_valid is not None and not x_valid[q]:
                continue
            acc = acc + (exp_factor * (1 - exp_factor) ** i) * float(x[q])
        out.append(acc)
    return perm, out


def lookback(key, ts, xs, xs_valid, L, ts_is_seconds=False, seq=None, seq_valid=None):
    perm = window_order(key, ts, ts_is_seconds, seq, seq_valid)
    series = {}
    out = []
    for r in perm:
        k = int(key[r]) if key is not None else 0
        hist = series.setdefault(k, [])
        window = hist[-L:] if L > 0 else []
        out.append([[None if (xs_valid[f] is not None and not xs_valid[f][q]) else float(xs[f][q])
                     for f in range(len(xs))] for q in window])
        hist.append(r)
    return perm, out


def resample_buckets(key, ts, dur_us):
    """{(key, bucket_start): [rows in input order]} with Spark's window() bucket start."""
    b = {}
    for i in range(len(ts)):
        t = int(ts[i])
        start = t - ((t % dur_us) + dur_us) % dur_us
        b.setdefault((int(key[i]) if key is not None else 0, start), []).append(i)
    return dict(sorted(b.items()))
