"""Test-only stand-in for tempo_amd.ops that answers with the CPU oracle on host tensors.

Lets the CPU suite exercise the host logic of tempo_amd (column encoding, naming, errors, result
assembly) without a GPU.  Never imported by the product."""
import numpy as np
import torch

from oracle import oracle


def _np(t):
    return None if t is None else t.cpu().numpy()


def _t(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t if dtype is None else t.to(dtype)


def sort_order(key, ts, n_keys, floor_seconds=False, seq=None, seq_valid=None):
    perm, _ = oracle.window_order(_np(key), _np(ts), ts_is_seconds=not floor_seconds, seq=_np(seq),
                                  seq_valid=_np(seq_valid))
    return _t(perm, torch.int32)


def asof_join(key_l, ts_l, key_r, ts_r, n_keys, seq_r=None, seq_r_valid=None, masks=(), ts_is_seconds=False,
              ts_partition_val=None, fraction=0.5):
    nL, nR = ts_l.numel(), ts_r.numel()
    kl = _np(key_l) if key_l is not None else np.zeros(nL, np.int32)
    kr = _np(key_r) if key_r is not None else np.zeros(nR, np.int32)
    idx, missing = oracle.asof_join(kl, _np(ts_l), kr, _np(ts_r), seq_r=_np(seq_r), seq_r_valid=_np(seq_r_valid),
                                    valid_r=[None] + [_np(m) for m in masks], skip_nulls=True,
                                    ts_is_seconds=ts_is_seconds, ts_partition_val=ts_partition_val, fraction=fraction)
    return _t(idx[0], torch.int32), _t(idx[1:].reshape(len(masks), nL), torch.int32), _t(missing, torch.int32)


def asof_join_table(key_l, ts_l, key_r, ts_r, n_keys, left_cols, right_cols, right_col_masks, skip_nulls=True,
                    seq_r=None, seq_r_valid=None, ts_is_seconds=False, ts_partition_val=None, fraction=0.5):
    """The fused join's result from the oracle's match indices: rows in (key, ts) order (ties in
    input order), left / right columns gathered on the host."""
    nL, nR = ts_l.numel(), ts_r.numel()
    kl = _np(key_l) if key_l is not None else np.zeros(nL, np.int32)
    kr = _np(key_r) if key_r is not None else np.zeros(nR, np.int32)
    masks, mask_of = [], []
    for m in right_col_masks:
        if m is None:
            mask_of.append(-1)
            continue
        slot = next((k for k, mm in enumerate(masks) if mm is m), None)
        if slot is None:
            slot = len(masks)
            masks.append(m)
        mask_of.append(slot)
    idx, missing = oracle.asof_join(kl, _np(ts_l), kr, _np(ts_r), seq_r=_np(seq_r), seq_r_valid=_np(seq_r_valid),
                                    valid_r=[None] + [_np(m) for m in masks], skip_nulls=True,
                                    ts_is_seconds=ts_is_seconds, ts_partition_val=ts_partition_val, fraction=fraction)
    tl = _np(ts_l)
    perm = np.lexsort((np.arange(nL), tl, kl))
    right = []
    for c, colv in enumerate(right_cols):
        v = _np(colv)
        slot = mask_of[c]
        src = idx[1 + slot] if (skip_nulls and slot >= 0) else idx[0]
        ok = src >= 0
        if not skip_nulls and slot >= 0 and nR:
            ok = ok & (_np(masks[slot])[np.maximum(src, 0)] != 0)
        vals = np.where(ok, v[np.maximum(src, 0)], 0) if nR else np.zeros(nL, v.dtype)
        right.append((_t(vals[perm].astype(v.dtype)), _t(ok[perm].astype(np.uint8))))
    return dict(perm=_t(perm, torch.int32), key=_t(kl[perm]), ts=_t(tl[perm]),
                left=[_t(_np(c)[perm]) for c in left_cols], right=right, missing=_t(missing, torch.int32),
                mask_of=mask_of)


def gather(src, idx, src_valid=None, want_valid=True):
    i = idx.to(torch.int64)
    ok = i >= 0
    safe = i.clamp(min=0)
    if src.numel() == 0:
        out = torch.zeros(idx.numel(), dtype=src.dtype)
    else:
        out = src[safe]
        if src_valid is not None:
            ok = ok & (src_valid[safe] != 0)
    out = torch.where(ok, out, torch.zeros_like(out))
    return out, (ok.to(torch.uint8) if want_valid else None)


def range_stats(key, ts, x, x_valid, n_keys, range_back_secs, ts_is_seconds=False):
    r = oracle.range_stats(_np(key), _np(ts), _np(x), _np(x_valid), range_back_secs, ts_is_seconds=ts_is_seconds)
    valid = r["stat_valid"] | (r["stddev_valid"] << 1) | (r["zscore_valid"] << 2)
    out = {k: _t(r[k]) for k in ("mean", "count", "min", "max", "sum", "stddev", "zscore")}
    out["perm"] = _t(r["perm"], torch.int32)
    out["valid"] = _t(valid.astype(np.uint8))
    return out


def range_stats_cols(key, ts, xs, xs_valid, n_keys, range_back_secs, ts_is_seconds=False):
    outs = [range_stats(key, ts, x, v, n_keys, range_back_secs, ts_is_seconds) for x, v in zip(xs, xs_valid)]
    perm = outs[0].pop("perm")
    for o in outs[1:]:
        assert torch.equal(o.pop("perm"), perm)
    return perm, outs


def ema(key, ts, x, x_valid, n_keys, window=30, exp_factor=0.2, ts_is_seconds=False, seq=None, seq_valid=None):
    perm, out = oracle.ema(_np(key), _np(ts), _np(x), _np(x_valid), window, exp_factor, ts_is_seconds=ts_is_seconds,
                           seq=_np(seq), seq_valid=_np(seq_valid))
    return _t(perm, torch.int32), _t(out)


def lookback(key, ts, xs, xs_valid, n_keys, lookback_size, exact_size=True, ts_is_seconds=False, seq=None,
             seq_valid=None):
    perm, ln, feat, fv = oracle.lookback(_np(key), _np(ts), [_np(x) for x in xs], [_np(v) for v in xs_valid],
                                         lookback_size, ts_is_seconds=ts_is_seconds, seq=_np(seq),
                                         seq_valid=_np(seq_valid))
    if exact_size:
        keep = ln == lookback_size
        return _t(perm[keep], torch.int32), None, _t(feat[keep]), _t(fv[keep])
    return _t(perm, torch.int32), _t(ln), _t(feat), _t(fv)


def resample(key, ts, n_keys, dur_us, func, cols, cols_valid):
    r = oracle.resample(_np(key), _np(ts), dur_us, func, [_np(c) for c in cols], [_np(v) for v in cols_valid])
    return dict(n=r["n"], key=_t(r["key"]), start=_t(r["start"]), f64=_t(r["f64"]), i64=_t(r["i64"]),
                valid=_t(r["valid"]), row=_t(r["row"], torch.int32), is_i64=r["is_i64"])


def grouped_stats(key, ts, n_keys, dur_us, cols, cols_valid):
    r = oracle.grouped_stats(_np(key), _np(ts), dur_us, [_np(c) for c in cols], [_np(v) for v in cols_valid])
    k = len(cols)
    return dict(n=r["n"], key=_t(r["key"]), start=_t(r["start"]), mean=[_t(r["mean"][c]) for c in range(k)],
                count=[_t(r["count"][c]) for c in range(k)], min=[_t(np.ascontiguousarray(r["min"][c])) for c in range(k)],
                max=[_t(np.ascontiguousarray(r["max"][c])) for c in range(k)],
                sum=[_t(np.ascontiguousarray(r["sum"][c])) for c in range(k)],
                stddev=[_t(r["stddev"][c]) for c in range(k)], is_i64=r["is_i64"])


def interpolate(key, ts, n_keys, freq_us, method, cols, cols_valid):
    r = oracle.interpolate(_np(key), _np(ts), freq_us, method, [_np(c) for c in cols], [_np(v) for v in cols_valid])
    k = len(cols)
    return dict(n=r["n"], src=_t(r["src"], torch.int32), key=_t(r["key"]), ts=_t(r["ts"]), ts_interp=_t(r["ts_interp"]),
                val=[_t(np.ascontiguousarray(r["val"][c])) for c in range(k)],
                valid=[_t(np.ascontiguousarray(r["valid"][c])) for c in range(k)],
                interp=[_t(np.ascontiguousarray(r["interp"][c])) for c in range(k)])


def write_layout(key, ts, n_keys):
    dt, tm, perm = oracle.write_layout(_np(key), _np(ts))
    rng = np.array([dt.min(), dt.max()] if len(dt) else [0, 0], dtype=np.int32)
    return _t(dt), _t(tm), _t(perm, torch.int32), _t(rng)
