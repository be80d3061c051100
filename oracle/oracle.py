"""numpy front end of the C oracle (oracle/tempo_oracle.c) -- TEST INFRASTRUCTURE ONLY.

Every function takes and returns host numpy arrays.  Timestamps are int64 microseconds
(Spark TimestampType) unless ts_is_seconds=True; keys are int32 dictionary ids; validity masks
are uint8 arrays (1 = not NULL) or None for "no NULLs".
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

FUNC_CODES = {"floor": 0, "min": 1, "max": 2, "mean": 3, "ceil": 4}


def build():
    """Compile liboracle.so next to its source (gcc, a second or two)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.orc_resample.restype = C.c_int64
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _i32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.int32)


def _i64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.int64)


def _u8(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.uint8)


def _f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def _ptr_array(arrs):
    n = max(len(arrs), 1)
    out = (C.c_void_p * n)()
    for i, a in enumerate(arrs):
        out[i] = None if a is None else a.ctypes.data
    return out


def asof_join(key_l, ts_l, key_r, ts_r, seq_r=None, seq_r_valid=None, valid_r=(), skip_nulls=True,
              ts_is_seconds=False, ts_partition_val=None, fraction=0.5):
    """Returns (idx[C, nL] int64 right-row index or -1, any_missing[C] int32)."""
    key_l, ts_l, key_r, ts_r = _i32(key_l), _i64(ts_l), _i32(key_r), _i64(ts_r)
    seq_r, seq_r_valid = _i64(seq_r), _u8(seq_r_valid)
    valid_r = [_u8(v) for v in valid_r]
    n_cols = len(valid_r)
    nL, nR = len(key_l), len(key_r)
    idx = np.full((max(n_cols, 1), nL), -2, dtype=np.int64)
    missing = np.zeros(max(n_cols, 1), dtype=np.int32)
    skew = ts_partition_val is not None
    v = float(ts_partition_val) if skew else 0.0
    v_is_int = int(skew and isinstance(ts_partition_val, (int, np.integer)))
    rc = lib().orc_asof_join(
        C.c_int64(nL), _p(key_l), _p(ts_l), C.c_int64(nR), _p(key_r), _p(ts_r), _p(seq_r),
        _p(seq_r_valid), C.c_int(n_cols), _ptr_array(valid_r), C.c_int(int(skip_nulls)),
        C.c_int(int(ts_is_seconds)), C.c_int(int(skew)), C.c_double(v), C.c_int(v_is_int),
        C.c_double(1 - fraction), _p(idx), _p(missing))
    assert rc == 0
    return idx[:n_cols], missing[:n_cols]


def window_order(key, ts, ts_is_seconds=False, seq=None, seq_valid=None):
    key, ts, seq, seq_valid = _i32(key), _i64(ts), _i64(seq), _u8(seq_valid)
    n = len(ts)
    perm = np.zeros(n, dtype=np.int64)
    sec = np.zeros(n, dtype=np.int64)
    rc = lib().orc_window_order(C.c_int64(n), _p(key), _p(ts), C.c_int(int(ts_is_seconds)),
                                _p(seq), _p(seq_valid), _p(perm), _p(sec))
    assert rc == 0
    return perm, sec


def range_stats(key, ts, x, x_valid, range_back_secs, ts_is_seconds=False):
    """Rolling stats in (key, sec) order.  x may be float64 or int64."""
    key, ts, x_valid = _i32(key), _i64(ts), _u8(x_valid)
    n = len(ts)
    is_int = np.issubdtype(np.asarray(x).dtype, np.integer)
    x = _i64(x) if is_int else _f64(x)
    vt = np.int64 if is_int else np.float64
    out = dict(perm=np.zeros(n, np.int64), mean=np.zeros(n), count=np.zeros(n, np.int64),
               min=np.zeros(n, vt), max=np.zeros(n, vt), sum=np.zeros(n, vt), stddev=np.zeros(n),
               zscore=np.zeros(n), stat_valid=np.zeros(n, np.uint8),
               stddev_valid=np.zeros(n, np.uint8), zscore_valid=np.zeros(n, np.uint8))
    fn = lib().orc_range_stats_i64 if is_int else lib().orc_range_stats_f64
    rc = fn(C.c_int64(n), _p(key), _p(ts), C.c_int(int(ts_is_seconds)), _p(x), _p(x_valid),
            C.c_int64(int(range_back_secs)), _p(out["perm"]), _p(out["mean"]), _p(out["count"]),
            _p(out["min"]), _p(out["max"]), _p(out["sum"]), _p(out["stddev"]), _p(out["zscore"]),
            _p(out["stat_valid"]), _p(out["stddev_valid"]), _p(out["zscore_valid"]))
    assert rc == 0
    return out


def ema_weights(window, exp_factor):
    """weight_i exactly as python/tempo/tsdf.py:629 evaluates it (Python float arithmetic)."""
    return np.array([exp_factor * (1 - exp_factor) ** i for i in range(window)], dtype=np.float64)


def ema(key, ts, x, x_valid, window=30, exp_factor=0.2, ts_is_seconds=False, seq=None,
        seq_valid=None):
    key, ts, x, x_valid = _i32(key), _i64(ts), _f64(x), _u8(x_valid)
    seq, seq_valid = _i64(seq), _u8(seq_valid)
    n = len(ts)
    w = ema_weights(window, exp_factor)
    perm = np.zeros(n, np.int64)
    out = np.zeros(n)
    rc = lib().orc_ema_f64(C.c_int64(n), _p(key), _p(ts), C.c_int(int(ts_is_seconds)), _p(seq),
                           _p(seq_valid), _p(x), _p(x_valid), C.c_int(window), _p(w), _p(perm),
                           _p(out))
    assert rc == 0
    return perm, out


def lookback(key, ts, xs, xs_valid, lookback_size, ts_is_seconds=False, seq=None, seq_valid=None):
    key, ts = _i32(key), _i64(ts)
    seq, seq_valid = _i64(seq), _u8(seq_valid)
    xs = [_f64(x) for x in xs]
    xs_valid = [_u8(v) for v in xs_valid]
    n, F, L = len(ts), len(xs), int(lookback_size)
    perm = np.zeros(n, np.int64)
    ln = np.zeros(n, np.int32)
    feat = np.zeros((n, L, F))
    fv = np.zeros((n, L, F), np.uint8)
    rc = lib().orc_lookback_f64(C.c_int64(n), _p(key), _p(ts), C.c_int(int(ts_is_seconds)),
                                _p(seq), _p(seq_valid), C.c_int(F), _ptr_array(xs),
                                _ptr_array(xs_valid), C.c_int(L), _p(perm), _p(ln), _p(feat),
                                _p(fv))
    assert rc == 0
    return perm, ln, feat, fv


def resample(key, ts, dur_us, func, cols, cols_valid):
    """cols: list of float64 or int64 arrays.  Returns dict with one entry per bucket."""
    key, ts = _i32(key), _i64(ts)
    n = len(ts)
    cc, is_i64 = [], []
    for c in cols:
        if np.issubdtype(np.asarray(c).dtype, np.integer):
            cc.append(_i64(c)); is_i64.append(1)
        else:
            cc.append(_f64(c)); is_i64.append(0)
    vv = [_u8(v) for v in cols_valid]
    k = len(cc)
    cap = max(n, 1)
    out_key = np.zeros(cap, np.int32)
    out_start = np.zeros(cap, np.int64)
    out_f64 = np.zeros((max(k, 1), cap))
    out_i64 = np.zeros((max(k, 1), cap), np.int64)
    out_valid = np.zeros((max(k, 1), cap), np.uint8)
    out_row = np.zeros(cap, np.int64)
    flags = (C.c_int * max(k, 1))(*is_i64)
    nb = lib().orc_resample(C.c_int64(n), _p(key), _p(ts), C.c_int64(int(dur_us)),
                            C.c_int(FUNC_CODES[func]), C.c_int(k), _ptr_array(cc), flags,
                            _ptr_array(vv), _p(out_key), _p(out_start), _p(out_f64), _p(out_i64),
                            _p(out_valid), _p(out_row))
    assert nb >= 0
    return dict(n=int(nb), key=out_key[:nb], start=out_start[:nb], f64=out_f64[:k, :nb],
                i64=out_i64[:k, :nb], valid=out_valid[:k, :nb], row=out_row[:nb], is_i64=is_i64)


def grouped_stats(key, ts, dur_us, cols, cols_valid):
    """withGroupedStats core.  cols: list of float64 or int64 arrays.  Returns dict with arrays of
    shape [n_cols][n_buckets]: mean, count, min, max, sum (min/max/sum as raw 8-byte cells viewed
    per column type via cell()), stddev."""
    key, ts = _i32(key), _i64(ts)
    n = len(ts)
    cc, is_i64 = [], []
    for c in cols:
        if np.issubdtype(np.asarray(c).dtype, np.integer):
            cc.append(_i64(c)); is_i64.append(1)
        else:
            cc.append(_f64(c)); is_i64.append(0)
    vv = [_u8(v) for v in cols_valid]
    k = len(cc)
    cap = max(n, 1)
    out_key = np.zeros(cap, np.int32)
    out_start = np.zeros(cap, np.int64)
    mean = np.zeros((max(k, 1), cap))
    cnt = np.zeros((max(k, 1), cap), np.int64)
    mn = np.zeros((max(k, 1), cap), np.int64)
    mx = np.zeros((max(k, 1), cap), np.int64)
    sm = np.zeros((max(k, 1), cap), np.int64)
    sd = np.zeros((max(k, 1), cap))
    flags = (C.c_int * max(k, 1))(*is_i64)
    fn = lib().orc_grouped_stats
    fn.restype = C.c_int64
    nb = fn(C.c_int64(n), _p(key), _p(ts), C.c_int64(int(dur_us)), C.c_int(k), _ptr_array(cc), flags,
            _ptr_array(vv), _p(out_key), _p(out_start), _p(mean), _p(cnt), _p(mn), _p(mx), _p(sm), _p(sd))
    assert nb >= 0

    def cell(a, c):
        return a[c, :nb] if is_i64[c] else a[c, :nb].view(np.float64)

    return dict(n=int(nb), key=out_key[:nb], start=out_start[:nb], mean=mean[:k, :nb], count=cnt[:k, :nb],
                min=[cell(mn, c) for c in range(k)], max=[cell(mx, c) for c in range(k)],
                sum=[cell(sm, c) for c in range(k)], stddev=sd[:k, :nb], is_i64=is_i64)


FILL_CODES = {"zero": 0, "null": 1, "bfill": 2, "ffill": 3, "linear": 4}


def interpolate(key, ts, freq_us, method, cols, cols_valid):
    """Interpolation.interpolate on a sampled table.  cols: list of float64 arrays.  Returns dict
    (n, src, key, ts, ts_interp, val[c], valid[c], interp[c]) in (key, ts) order."""
    key, ts = _i32(key), _i64(ts)
    n = len(ts)
    cc = [_f64(c) for c in cols]
    vv = [_u8(v) for v in cols_valid]
    k = len(cc)
    fn = lib().orc_interpolate
    fn.restype = C.c_int64
    cap = max(n, 1)
    while True:
        src = np.zeros(cap, np.int64)
        okey = np.zeros(cap, np.int32)
        ots = np.zeros(cap, np.int64)
        tsi = np.zeros(cap, np.uint8)
        val = np.zeros((max(k, 1), cap))
        valid = np.zeros((max(k, 1), cap), np.uint8)
        interp = np.zeros((max(k, 1), cap), np.uint8)
        m = fn(C.c_int64(n), _p(key), _p(ts), C.c_int64(int(freq_us)), C.c_int(FILL_CODES[method]), C.c_int(k),
               _ptr_array(cc), _ptr_array(vv), C.c_int64(cap), _p(src), _p(okey), _p(ots), _p(tsi), _p(val),
               _p(valid), _p(interp))
        if m == -3:
            cap *= 4
            continue
        if m == -2:
            raise ValueError("a gap between two rows of one key is shorter than freq")
        assert m >= 0
        return dict(n=int(m), src=src[:m], key=okey[:m], ts=ots[:m], ts_interp=tsi[:m], val=val[:k, :m],
                    valid=valid[:k, :m], interp=interp[:k, :m])


def write_layout(key, ts):
    """(event_dt int32[n], event_time float64[n], perm int64[n]) of python/tempo/io.py:29-39:
    derived columns in input row order, perm = rows in (event_dt, key, ts) order."""
    key, ts = _i32(key), _i64(ts)
    n = len(ts)
    dt = np.zeros(n, dtype=np.int32)
    tm = np.zeros(n, dtype=np.float64)
    perm = np.zeros(n, dtype=np.int64)
    rc = lib().orc_write_layout(C.c_int64(n), _p(key), _p(ts), _p(dt), _p(tm), _p(perm))
    assert rc == 0
    return dt, tm, perm
