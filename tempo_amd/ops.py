"""Device-level operators: torch (ROCm) tensors in HBM -> libtempo_hip.so -> torch tensors.

torch is plumbing here (device memory, streams); every computation is a HIP kernel behind the
C ABI of include/tempo_hip.h.  All functions require CUDA (ROCm) tensors and raise otherwise.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import (AsofArgs, EmaArgs, LookbackArgs, PartitionArgs, RangeStatsArgs, ResampleArgs,
                   SortArgs, check, lib, ptr_array)


def _dev(t, dtype, name, n=None):
    if t is None:
        return None
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TypeError(f"{name} must be a CUDA (ROCm) tensor; tempo_amd has no CPU path")
    if t.dtype != dtype:
        raise TypeError(f"{name} must have dtype {dtype}, got {t.dtype}")
    if t.dim() != 1 or not t.is_contiguous():
        raise ValueError(f"{name} must be a contiguous 1-D tensor")
    if n is not None and t.numel() != n:
        raise ValueError(f"{name} has {t.numel()} rows, expected {n}")
    return t


def _p(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _workspace(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def sort_order(key, ts, n_keys, floor_seconds=False, seq=None, seq_valid=None):
    """perm[p] = input row at sorted position p by (key, ts or floor(ts/1e6)[, seq NULLS FIRST]),
    ties in input order (python/tempo/tsdf.py:563-572)."""
    n = ts.numel()
    _dev(ts, torch.int64, "ts")
    _dev(key, torch.int32, "key", n)
    _dev(seq, torch.int64, "seq", n)
    _dev(seq_valid, torch.uint8, "seq_valid", n)
    perm = torch.empty(n, dtype=torch.int32, device=ts.device)
    a = SortArgs(n, _p(key), int(n_keys), _p(ts), int(bool(floor_seconds)), _p(seq), _p(seq_valid),
                 _p(perm))
    ws = _workspace(lib().tmp_sort_order_workspace_bytes(C.byref(a)), ts.device)
    check(lib().tmp_sort_order(C.byref(a), ws.data_ptr(), ws.numel(), _stream()), "tmp_sort_order")
    return perm


def _too_wide(err):
    """The library's refusal of a table whose (key, time[, sequence flag]) does not fit one 64-bit
    sort word even with the sequence values and validity bits moved out of it."""
    return "sort word needs" in str(err)


def _key_halves(key_l, key_r, n_keys):
    """Row selections of the two halves of the key dictionary, for joining a table whose sort word
    is too wide half by half (series are independent per key, and key order is row order)."""
    mid = n_keys // 2
    for lo, hi in ((0, mid), (mid, n_keys)):
        sel_l = torch.nonzero((key_l >= lo) & (key_l < hi)).squeeze(1)
        sel_r = torch.nonzero((key_r >= lo) & (key_r < hi)).squeeze(1)
        yield lo, hi, sel_l, sel_r


def _take(t, sel):
    return None if t is None else t[sel].contiguous()


def asof_join(key_l, ts_l, key_r, ts_r, n_keys, seq_r=None, seq_r_valid=None, masks=(),
              ts_is_seconds=False, ts_partition_val=None, fraction=0.5, sorted_output=False):
    """AS-OF match; a table too wide for one sort word is joined one half of the key dictionary at
    a time (left row order only)."""
    try:
        return _asof_join_once(key_l, ts_l, key_r, ts_r, n_keys, seq_r, seq_r_valid, masks, ts_is_seconds,
                               ts_partition_val, fraction, sorted_output)
    except NotImplementedError as err:
        if not _too_wide(err) or key_l is None or n_keys < 2 or sorted_output:
            raise
    masks = list(masks)
    nL = ts_l.numel()
    idx = torch.full((nL, 1 + len(masks)), -1, dtype=torch.int32, device=ts_l.device)
    missing = torch.zeros(1 + len(masks), dtype=torch.int32, device=ts_l.device)
    for lo, hi, sel_l, sel_r in _key_halves(key_l, key_r, n_keys):
        if sel_l.numel() == 0:
            continue
        row, midx, miss = asof_join(_take(key_l, sel_l) - lo, _take(ts_l, sel_l), _take(key_r, sel_r) - lo,
                                    _take(ts_r, sel_r), hi - lo, _take(seq_r, sel_r), _take(seq_r_valid, sel_r),
                                    [_take(m, sel_r) for m in masks], ts_is_seconds, ts_partition_val, fraction)
        sub = torch.cat([row.unsqueeze(1), midx.t()], dim=1).long()
        back = torch.where(sub >= 0, sel_r[sub.clamp(min=0)] if sel_r.numel() else sub, sub)
        idx[sel_l] = back.to(torch.int32)
        missing |= miss
    return idx[:, 0], idx[:, 1:].t(), missing


def _asof_join_once(key_l, ts_l, key_r, ts_r, n_keys, seq_r=None, seq_r_valid=None, masks=(),
                    ts_is_seconds=False, ts_partition_val=None, fraction=0.5, sorted_output=False):
    """AS-OF match (python/tempo/tsdf.py:463-560).

    Returns (row_idx[nL], mask_idx[n_masks, nL], missing[1 + n_masks]) as int32 device tensors:
    the right row each left row takes its values from (last visible row / last visible row
    whose mask is valid), -1 for none.  row_idx and mask_idx are strided views of one
    [nL, 1 + n_masks] record array (gather() accepts them as they are).

    sorted_output=True returns the records in (key, ts) order instead of the left table's row
    order, followed by (perm, key_sorted, ts_sorted): record p belongs to left row perm[p].
    """
    nL, nR = ts_l.numel(), ts_r.numel()
    dev = ts_l.device
    _dev(ts_l, torch.int64, "ts_l")
    _dev(ts_r, torch.int64, "ts_r")
    _dev(key_l, torch.int32, "key_l", nL)
    _dev(key_r, torch.int32, "key_r", nR)
    _dev(seq_r, torch.int64, "seq_r", nR)
    _dev(seq_r_valid, torch.uint8, "seq_r_valid", nR)
    masks = list(masks)
    if len(masks) > _lib.TMP_MAX_MASKS:
        raise NotImplementedError(f"at most {_lib.TMP_MAX_MASKS} nullable right columns per call")
    for i, m in enumerate(masks):
        _dev(m, torch.uint8, f"masks[{i}]", nR)
    idx = torch.empty((nL, 1 + len(masks)), dtype=torch.int32, device=dev)
    missing = torch.empty(1 + len(masks), dtype=torch.int32, device=dev)
    skew = ts_partition_val is not None
    a = AsofArgs()
    a.n_left, a.n_right = nL, nR
    a.key_left, a.ts_left, a.key_right, a.ts_right = _p(key_l), _p(ts_l), _p(key_r), _p(ts_r)
    a.seq_right, a.seq_right_valid = _p(seq_r), _p(seq_r_valid)
    a.n_keys, a.ts_is_seconds, a.n_masks = int(n_keys), int(bool(ts_is_seconds)), len(masks)
    mask_ptrs = ptr_array([m.data_ptr() for m in masks])
    a.masks = mask_ptrs
    a.skew = int(skew)
    a.v_is_int = int(skew and isinstance(ts_partition_val, int))
    a.ts_partition_val = float(ts_partition_val) if skew else 0.0
    a.one_minus_fraction = float(1 - fraction)  # evaluated in Python float as tsdf.py:185
    a.out_idx, a.out_missing = _p(idx), _p(missing)
    perm = key_s = ts_s = None
    if sorted_output:
        perm = torch.empty(nL, dtype=torch.int32, device=dev)
        key_s = torch.empty(nL, dtype=torch.int32, device=dev)
        ts_s = torch.empty(nL, dtype=torch.int64, device=dev)
        a.sorted_output, a.out_perm, a.out_key, a.out_ts = 1, _p(perm), _p(key_s), _p(ts_s)
    ws = _workspace(lib().tmp_asof_join_workspace_bytes(C.byref(a)), dev)
    check(lib().tmp_asof_join(C.byref(a), ws.data_ptr(), ws.numel(), _stream()), "tmp_asof_join")
    if sorted_output:
        return idx[:, 0], idx[:, 1:].t(), missing, perm, key_s, ts_s
    return idx[:, 0], idx[:, 1:].t(), missing


def _same_mask_key(m):
    """Two right columns share a validity mask when they were given the same memory.  Empty
    tensors all report the same (null) address, so those are told apart by object identity."""
    return ("ptr", m.data_ptr()) if m.numel() else ("obj", id(m))


def asof_join_table(key_l, ts_l, key_r, ts_r, n_keys, left_cols, right_cols, right_col_masks,
                    skip_nulls=True, seq_r=None, seq_r_valid=None, ts_is_seconds=False,
                    ts_partition_val=None, fraction=0.5, left_by_reference=False):
    """Fused AS-OF join; a table too wide for one sort word is joined one half of the key
    dictionary at a time and the halves are concatenated (key order is row order).

    left_by_reference=True: when the left table turns out to be in (key, ts) order already, row p of
    the result is left row p and nothing of the left table is copied -- the result then has
    in_place=True, perm=None, and key / ts / left are the caller's own tensors."""
    try:
        return _asof_join_table_once(key_l, ts_l, key_r, ts_r, n_keys, left_cols, right_cols, right_col_masks,
                                     skip_nulls, seq_r, seq_r_valid, ts_is_seconds, ts_partition_val, fraction,
                                     left_by_reference)
    except NotImplementedError as err:
        if not _too_wide(err) or key_l is None or n_keys < 2:
            raise
    parts = []
    for lo, hi, sel_l, sel_r in _key_halves(key_l, key_r, n_keys):
        taken = {}  # one selection per distinct mask, so that shared masks stay shared
        sub_masks = []
        for m in right_col_masks:
            if m is not None and _same_mask_key(m) not in taken:
                taken[_same_mask_key(m)] = _take(m, sel_r)
            sub_masks.append(None if m is None else taken[_same_mask_key(m)])
        j = asof_join_table(_take(key_l, sel_l) - lo, _take(ts_l, sel_l), _take(key_r, sel_r) - lo, _take(ts_r, sel_r),
                            hi - lo, [_take(c, sel_l) for c in left_cols], [_take(c, sel_r) for c in right_cols],
                            sub_masks, skip_nulls, _take(seq_r, sel_r), _take(seq_r_valid, sel_r), ts_is_seconds,
                            ts_partition_val, fraction)
        j["perm"] = sel_l[j["perm"].long()].to(torch.int32)
        j["key"] = j["key"] + lo
        parts.append(j)
    a, b = parts
    return dict(perm=torch.cat([a["perm"], b["perm"]]), key=torch.cat([a["key"], b["key"]]),
                ts=torch.cat([a["ts"], b["ts"]]), left=[torch.cat(p) for p in zip(a["left"], b["left"])],
                right=[(torch.cat([x[0], y[0]]), torch.cat([x[1], y[1]])) for x, y in zip(a["right"], b["right"])],
                missing=a["missing"] | b["missing"], mask_of=a["mask_of"], in_place=False)


def _asof_join_table_once(key_l, ts_l, key_r, ts_r, n_keys, left_cols, right_cols, right_col_masks,
                          skip_nulls=True, seq_r=None, seq_r_valid=None, ts_is_seconds=False,
                          ts_partition_val=None, fraction=0.5, left_by_reference=False):
    """Fused AS-OF join: the joined table itself, in (key, ts) order, written with streaming
    accesses only (the payload columns ride along with the last radix pass of their table).

    left_cols / right_cols: lists of 8-byte-element device columns (float64 / int64);
    right_col_masks[c]: uint8 validity of right column c or None.  Returns dict(perm, key, ts,
    left=[columns], right=[(column, valid uint8)], missing=int32[1 + n_masks], mask_of=[...]).
    """
    nL, nR = ts_l.numel(), ts_r.numel()
    dev = ts_l.device
    _dev(ts_l, torch.int64, "ts_l")
    _dev(ts_r, torch.int64, "ts_r")
    _dev(key_l, torch.int32, "key_l", nL)
    _dev(key_r, torch.int32, "key_r", nR)
    _dev(seq_r, torch.int64, "seq_r", nR)
    _dev(seq_r_valid, torch.uint8, "seq_r_valid", nR)
    if len(left_cols) > _lib.TMP_MAX_JOIN_COLS or len(right_cols) > _lib.TMP_MAX_JOIN_COLS:
        raise NotImplementedError(f"at most {_lib.TMP_MAX_JOIN_COLS} payload columns per side")
    for i, c in enumerate(left_cols):
        if c.element_size() != 8:
            raise TypeError("fused join columns must have 8-byte elements")
        _dev(c, c.dtype, f"left_cols[{i}]", nL)
    masks, mask_of, rule, vmask = [], [], [], []
    for i, c in enumerate(right_cols):
        if c.element_size() != 8:
            raise TypeError("fused join columns must have 8-byte elements")
        _dev(c, c.dtype, f"right_cols[{i}]", nR)
        m = right_col_masks[i]
        if m is None:
            mask_of.append(-1); rule.append(-1); vmask.append(-1)
            continue
        _dev(m, torch.uint8, f"right_col_masks[{i}]", nR)
        slot = next((k for k, mm in enumerate(masks) if _same_mask_key(mm) == _same_mask_key(m)), None)
        if slot is None:
            slot = len(masks)
            masks.append(m)
        mask_of.append(slot)
        rule.append(slot if skip_nulls else -1)
        vmask.append(-1 if skip_nulls else slot)
    if len(masks) > _lib.TMP_MAX_MASKS:
        raise NotImplementedError(f"at most {_lib.TMP_MAX_MASKS} nullable right columns per call")
    perm = torch.empty(nL, dtype=torch.int32, device=dev)
    key_s = torch.empty(nL, dtype=torch.int32, device=dev)
    ts_s = torch.empty(nL, dtype=torch.int64, device=dev)
    missing = torch.empty(1 + len(masks), dtype=torch.int32, device=dev)
    out_left = [torch.empty(nL, dtype=c.dtype, device=dev) for c in left_cols]
    out_right = [torch.empty(nL, dtype=c.dtype, device=dev) for c in right_cols]
    out_valid = [torch.empty(nL, dtype=torch.uint8, device=dev) for _ in right_cols]
    skew = ts_partition_val is not None
    a = AsofArgs()
    a.n_left, a.n_right = nL, nR
    a.key_left, a.ts_left, a.key_right, a.ts_right = _p(key_l), _p(ts_l), _p(key_r), _p(ts_r)
    a.seq_right, a.seq_right_valid = _p(seq_r), _p(seq_r_valid)
    a.n_keys, a.ts_is_seconds, a.n_masks = int(n_keys), int(bool(ts_is_seconds)), len(masks)
    mask_ptrs = ptr_array([m.data_ptr() for m in masks])
    a.masks = mask_ptrs
    a.skew = int(skew)
    a.v_is_int = int(skew and isinstance(ts_partition_val, int))
    a.ts_partition_val = float(ts_partition_val) if skew else 0.0
    a.one_minus_fraction = float(1 - fraction)
    a.out_idx, a.out_missing = None, _p(missing)
    a.sorted_output, a.out_perm, a.out_key, a.out_ts = 1, _p(perm), _p(key_s), _p(ts_s)
    keep = [ptr_array([c.data_ptr() for c in left_cols]), ptr_array([c.data_ptr() for c in out_left]),
            ptr_array([c.data_ptr() for c in right_cols]), ptr_array([c.data_ptr() for c in out_right]),
            ptr_array([c.data_ptr() for c in out_valid]), (C.c_int32 * max(len(rule), 1))(*rule),
            (C.c_int32 * max(len(vmask), 1))(*vmask)]
    a.n_left_cols, a.left_cols, a.out_left_cols = len(left_cols), keep[0], keep[1]
    a.n_right_cols, a.right_cols, a.out_right_cols, a.out_right_valid = len(right_cols), keep[2], keep[3], keep[4]
    a.right_col_rule, a.right_col_valid = keep[5], keep[6]
    in_place = C.c_int32(0)
    if left_by_reference:
        a.left_in_place = C.pointer(in_place)
    ws = _workspace(lib().tmp_asof_join_workspace_bytes(C.byref(a)), dev)
    check(lib().tmp_asof_join(C.byref(a), ws.data_ptr(), ws.numel(), _stream()), "tmp_asof_join")
    if in_place.value:  # the left table was in order: its own columns are the result's
        perm, ts_s, out_left = None, ts_l, list(left_cols)
        key_s = key_l if key_l is not None else key_s.zero_()
    return dict(perm=perm, key=key_s, ts=ts_s, left=out_left, right=list(zip(out_right, out_valid)),
                missing=missing, mask_of=mask_of, in_place=bool(in_place.value))


def gather(src, idx, src_valid=None, want_valid=True):
    """out[i] = src[idx[i]] (0 / invalid where idx[i] < 0 or the source is NULL)."""
    n = idx.numel()
    if not isinstance(idx, torch.Tensor) or not idx.is_cuda or idx.dtype != torch.int32 or idx.dim() != 1:
        raise TypeError("idx must be a 1-D int32 CUDA (ROCm) tensor; tempo_amd has no CPU path")
    idx_stride = idx.stride(0) if n > 1 else 1
    if idx_stride < 1:
        raise ValueError("idx must have a positive stride")
    if not src.is_cuda or src.dim() != 1 or not src.is_contiguous():
        raise ValueError("src must be a contiguous 1-D CUDA tensor")
    if src.element_size() not in (1, 4, 8):
        raise TypeError("gather supports 1-, 4- and 8-byte element types")
    _dev(src_valid, torch.uint8, "src_valid", src.numel())
    out = torch.empty(n, dtype=src.dtype, device=src.device)
    out_valid = torch.empty(n, dtype=torch.uint8, device=src.device) if want_valid else None
    check(lib().tmp_gather(_p(src), _p(src_valid), src.element_size(), _p(idx), idx_stride, n, _p(out),
                           _p(out_valid), _stream()), "tmp_gather")
    return out, out_valid


def range_stats(key, ts, x, x_valid, n_keys, range_back_secs, ts_is_seconds=False, any_order=False):
    """withRangeStats core (python/tempo/tsdf.py:673-721).  Outputs in (key, sec) order, out["perm"]
    naming the input row of each; any_order=True: in whatever row order the library finds cheapest
    (a time-ordered feed over few (key, second) cells comes back in its own order, perm = identity)."""
    n = ts.numel()
    dev = ts.device
    _dev(ts, torch.int64, "ts")
    _dev(key, torch.int32, "key", n)
    is_i64 = x.dtype == torch.int64
    _dev(x, torch.int64 if is_i64 else torch.float64, "x", n)
    _dev(x_valid, torch.uint8, "x_valid", n)
    vt = torch.int64 if is_i64 else torch.float64
    out = dict(perm=torch.empty(n, dtype=torch.int32, device=dev),
               mean=torch.empty(n, dtype=torch.float64, device=dev),
               count=torch.empty(n, dtype=torch.int64, device=dev),
               min=torch.empty(n, dtype=vt, device=dev), max=torch.empty(n, dtype=vt, device=dev),
               sum=torch.empty(n, dtype=vt, device=dev),
               stddev=torch.empty(n, dtype=torch.float64, device=dev),
               zscore=torch.empty(n, dtype=torch.float64, device=dev),
               valid=torch.empty(n, dtype=torch.uint8, device=dev))
    a = RangeStatsArgs(n, _p(key), int(n_keys), _p(ts), int(bool(ts_is_seconds)), int(is_i64),
                       _p(x), _p(x_valid), int(range_back_secs), _p(out["perm"]), _p(out["mean"]),
                       _p(out["count"]), _p(out["min"]), _p(out["max"]), _p(out["sum"]),
                       _p(out["stddev"]), _p(out["zscore"]), _p(out["valid"]), int(bool(any_order)))
    _run_with_wide_fallback(lib().tmp_range_stats, lib().tmp_range_stats_workspace_bytes,
                            lib().tmp_range_stats_wide_workspace_bytes, a, n, dev, "tmp_range_stats")
    return out


# Tables up to this many rows get the workspace of the long-frame path (about 60 bytes per row more)
# straight away, so that their call never has to be made twice.
WIDE_WORKSPACE_UPFRONT_ROWS = 1 << 25


def _run_with_wide_fallback(fn, plain_bytes, wide_bytes, a, n, dev, what):
    """Calls a range-statistics entry point.  Small tables bring the long-frame workspace along from
    the start; a large one is first tried with the plain workspace and run again with the larger one
    only if the library reports a frame that leaves the on-chip region (TMP_EWIDE)."""
    upfront = n <= WIDE_WORKSPACE_UPFRONT_ROWS
    ws = _workspace((wide_bytes if upfront else plain_bytes)(C.byref(a)), dev)
    rc = fn(C.byref(a), ws.data_ptr(), ws.numel(), _stream())
    if rc == _lib.TMP_EWIDE and not upfront:
        del ws
        ws = _workspace(wide_bytes(C.byref(a)), dev)
        rc = fn(C.byref(a), ws.data_ptr(), ws.numel(), _stream())
    check(rc, what)


def range_stats_cols(key, ts, xs, xs_valid, n_keys, range_back_secs, ts_is_seconds=False):
    """withRangeStats over several metric columns of one table (python/tempo/tsdf.py:707-717): one
    ordering step, every column travelling with the rows, then the statistics column by column.
    xs: float64 or int64 device columns; xs_valid: uint8 masks or None.  Returns (perm, [out dict per
    column as range_stats gives it, without perm]).  More than _lib.TMP_MAX_STATS_COLS columns go in
    groups of that many (each group orders the rows again)."""
    n = ts.numel()
    dev = ts.device
    _dev(ts, torch.int64, "ts")
    _dev(key, torch.int32, "key", n)
    perm = torch.empty(n, dtype=torch.int32, device=dev)
    outs = []
    for g0 in range(0, len(xs), _lib.TMP_MAX_STATS_COLS):
        gx, gv = xs[g0:g0 + _lib.TMP_MAX_STATS_COLS], xs_valid[g0:g0 + _lib.TMP_MAX_STATS_COLS]
        k = len(gx)
        is_i64, group = [], []
        for i, (x, v) in enumerate(zip(gx, gv)):
            i64 = x.dtype == torch.int64
            _dev(x, torch.int64 if i64 else torch.float64, f"xs[{g0 + i}]", n)
            _dev(v, torch.uint8, f"xs_valid[{g0 + i}]", n)
            vt = torch.int64 if i64 else torch.float64
            is_i64.append(int(i64))
            group.append(dict(mean=torch.empty(n, dtype=torch.float64, device=dev),
                              count=torch.empty(n, dtype=torch.int64, device=dev),
                              min=torch.empty(n, dtype=vt, device=dev), max=torch.empty(n, dtype=vt, device=dev),
                              sum=torch.empty(n, dtype=vt, device=dev),
                              stddev=torch.empty(n, dtype=torch.float64, device=dev),
                              zscore=torch.empty(n, dtype=torch.float64, device=dev),
                              valid=torch.empty(n, dtype=torch.uint8, device=dev)))
        a = _lib.RangeStatsColsArgs()
        a.n, a.key, a.n_keys, a.ts, a.ts_is_seconds = n, _p(key), int(n_keys), _p(ts), int(bool(ts_is_seconds))
        a.range_back_secs, a.n_cols, a.out_perm = int(range_back_secs), k, _p(perm)
        keep = [ptr_array([x.data_ptr() for x in gx]), ptr_array([_p(v) for v in gv]), (C.c_int32 * k)(*is_i64)]
        a.x, a.x_valid, a.x_is_i64 = keep
        for name, field in (("mean", "out_mean"), ("count", "out_count"), ("min", "out_min"), ("max", "out_max"),
                            ("sum", "out_sum"), ("stddev", "out_stddev"), ("zscore", "out_zscore"),
                            ("valid", "out_valid")):
            arr = ptr_array([g[name].data_ptr() for g in group])
            keep.append(arr)
            setattr(a, field, arr)
        if n:
            _run_with_wide_fallback(lib().tmp_range_stats_cols, lib().tmp_range_stats_cols_workspace_bytes,
                                    lib().tmp_range_stats_cols_wide_workspace_bytes, a, n, dev, "tmp_range_stats_cols")
        outs.extend(group)
    return perm, outs


def ema_weights(window, exp_factor):
    """Tap weights exactly as python/tempo/tsdf.py:629 computes them (Python float)."""
    return [exp_factor * (1 - exp_factor) ** i for i in range(window)]


def ema(key, ts, x, x_valid, n_keys, window=30, exp_factor=0.2, ts_is_seconds=False, seq=None,
        seq_valid=None):
    """EMA core (python/tempo/tsdf.py:615-635).  Returns (perm, ema) in window order."""
    n = ts.numel()
    dev = ts.device
    _dev(ts, torch.int64, "ts")
    _dev(key, torch.int32, "key", n)
    _dev(x, torch.float64, "x", n)
    _dev(x_valid, torch.uint8, "x_valid", n)
    _dev(seq, torch.int64, "seq", n)
    _dev(seq_valid, torch.uint8, "seq_valid", n)
    if window > _lib.TMP_MAX_TAPS:
        raise NotImplementedError(f"EMA window is limited to {_lib.TMP_MAX_TAPS} taps")
    w = ema_weights(window, exp_factor)
    warr = (C.c_double * max(window, 1))(*w)
    perm = torch.empty(n, dtype=torch.int32, device=dev)
    out = torch.empty(n, dtype=torch.float64, device=dev)
    a = EmaArgs(n, _p(key), int(n_keys), _p(ts), int(bool(ts_is_seconds)), _p(seq), _p(seq_valid),
                _p(x), _p(x_valid), int(window), warr, _p(perm), _p(out))
    ws = _workspace(lib().tmp_ema_workspace_bytes(C.byref(a)), dev)
    check(lib().tmp_ema(C.byref(a), ws.data_ptr(), ws.numel(), _stream()), "tmp_ema")
    return perm, out


def lookback(key, ts, xs, xs_valid, n_keys, lookback_size, exact_size=True, ts_is_seconds=False,
             seq=None, seq_valid=None):
    """withLookbackFeatures core (python/tempo/tsdf.py:637-671).

    Returns (rows, lengths or None, feat[n_out, L, F], feat_valid or None); n_out is the number
    of kept rows (all rows when exact_size is False)."""
    n = ts.numel()
    dev = ts.device
    _dev(ts, torch.int64, "ts")
    _dev(key, torch.int32, "key", n)
    _dev(seq, torch.int64, "seq", n)
    _dev(seq_valid, torch.uint8, "seq_valid", n)
    F, L = len(xs), int(lookback_size)
    if F < 1 or F > _lib.TMP_MAX_FEATURES:
        raise ValueError(f"between 1 and {_lib.TMP_MAX_FEATURES} feature columns are supported")
    if L < 1:
        raise ValueError("lookbackWindowSize must be an integer >= 1")
    for i, x in enumerate(xs):
        _dev(x, torch.float64, f"xs[{i}]", n)
        _dev(xs_valid[i], torch.uint8, f"xs_valid[{i}]", n)
    any_nulls = any(v is not None for v in xs_valid)
    rows = torch.empty(n, dtype=torch.int32, device=dev)
    lens = None if exact_size else torch.empty(n, dtype=torch.int32, device=dev)
    feat = torch.empty((n, L, F), dtype=torch.float64, device=dev)
    fv = torch.empty((n, L, F), dtype=torch.uint8, device=dev) if (any_nulls or not exact_size) else None
    count = torch.zeros(1, dtype=torch.int64, device=dev)
    xp = ptr_array([x.data_ptr() for x in xs])
    vp = ptr_array([_p(v) for v in xs_valid])
    a = LookbackArgs(n, _p(key), int(n_keys), _p(ts), int(bool(ts_is_seconds)), _p(seq),
                     _p(seq_valid), F, xp, vp, L, int(bool(exact_size)), _p(rows), _p(lens),
                     _p(feat), _p(fv), _p(count))
    ws = _workspace(lib().tmp_lookback_workspace_bytes(C.byref(a)), dev)
    check(lib().tmp_lookback(C.byref(a), ws.data_ptr(), ws.numel(), _stream()), "tmp_lookback")
    n_out = int(count.item())
    return rows[:n_out], lens, feat[:n_out], (None if fv is None else fv[:n_out])


def resample(key, ts, n_keys, dur_us, func, cols, cols_valid):
    """resample core (python/tempo/resample.py:38-117).  cols: float64 or int64 device tensors.

    Returns dict(n, key, start, f64[C, n], i64[C, n], valid[C, n], row[n], is_i64)."""
    n = ts.numel()
    dev = ts.device
    _dev(ts, torch.int64, "ts")
    _dev(key, torch.int32, "key", n)
    k = len(cols)
    if k > _lib.TMP_MAX_COLS:
        raise NotImplementedError(f"at most {_lib.TMP_MAX_COLS} columns per resample call")
    is_i64 = []
    for i, c in enumerate(cols):
        if c.dtype not in (torch.float64, torch.int64):
            raise TypeError("resample columns must be float64 or int64 device tensors")
        _dev(c, c.dtype, f"cols[{i}]", n)
        _dev(cols_valid[i], torch.uint8, f"cols_valid[{i}]", n)
        is_i64.append(int(c.dtype == torch.int64))
    cap = max(n, 1)
    out_key = torch.empty(cap, dtype=torch.int32, device=dev)
    out_start = torch.empty(cap, dtype=torch.int64, device=dev)
    reduce_func = func in ("min", "max", "mean")
    kk = max(k, 1) if reduce_func else 1
    # every cell of the first `count` buckets is written by the call; nothing beyond them is read
    out_f64 = torch.empty((kk, cap), dtype=torch.float64, device=dev)
    out_i64 = torch.empty((kk, cap), dtype=torch.int64, device=dev)
    out_valid = torch.empty((kk, cap), dtype=torch.uint8, device=dev)
    out_row = torch.empty(cap, dtype=torch.int32, device=dev)
    count = torch.zeros(1, dtype=torch.int64, device=dev)
    a = ResampleArgs(n, _p(key), int(n_keys), _p(ts), int(dur_us), _lib.FUNC_CODES[func], k,
                     ptr_array([c.data_ptr() for c in cols]),
                     (C.c_int32 * max(k, 1))(*is_i64), ptr_array([_p(v) for v in cols_valid]),
                     _p(out_key), _p(out_start), _p(out_f64), _p(out_i64), _p(out_valid),
                     _p(out_row), _p(count))
    ws = _workspace(lib().tmp_resample_workspace_bytes(C.byref(a)), dev)
    check(lib().tmp_resample(C.byref(a), ws.data_ptr(), ws.numel(), _stream()), "tmp_resample")
    nb = int(count.item())
    return dict(n=nb, key=out_key[:nb], start=out_start[:nb], f64=out_f64[:, :nb],
                i64=out_i64[:, :nb], valid=out_valid[:, :nb], row=out_row[:nb], is_i64=is_i64)


def grouped_stats(key, ts, n_keys, dur_us, cols, cols_valid):
    """withGroupedStats core (python/tempo/tsdf.py:723-759).  cols: float64 or int64 device
    tensors.  Returns dict(n, key, start, mean, count, min, max, sum, stddev); the per-column
    entries are lists of tensors of length n (min / max / sum in the column's own type)."""
    n = ts.numel()
    dev = ts.device
    _dev(ts, torch.int64, "ts")
    _dev(key, torch.int32, "key", n)
    if len(cols) > _lib.TMP_MAX_COLS:
        raise NotImplementedError(f"at most {_lib.TMP_MAX_COLS} columns per call")
    is_i64 = []
    for i, c in enumerate(cols):
        if c.dtype not in (torch.float64, torch.int64):
            raise TypeError("grouped_stats columns must be float64 or int64")
        _dev(c, c.dtype, f"cols[{i}]", n)
        _dev(cols_valid[i], torch.uint8, f"cols_valid[{i}]", n)
        is_i64.append(int(c.dtype == torch.int64))
    k, cap = len(cols), max(n, 1)
    out_key = torch.empty(cap, dtype=torch.int32, device=dev)
    out_start = torch.empty(cap, dtype=torch.int64, device=dev)
    mean = torch.empty((max(k, 1), cap), dtype=torch.float64, device=dev)
    cnt = torch.empty((max(k, 1), cap), dtype=torch.int64, device=dev)
    mn = torch.empty((max(k, 1), cap), dtype=torch.int64, device=dev)
    mx = torch.empty_like(mn)
    sm = torch.empty_like(mn)
    sd = torch.empty_like(mean)
    count = torch.zeros(1, dtype=torch.int64, device=dev)
    a = _lib.GroupedStatsArgs()
    a.n, a.key, a.n_keys, a.ts, a.dur_us, a.n_cols = n, _p(key), int(n_keys), _p(ts), int(dur_us), k
    keep = [ptr_array([c.data_ptr() for c in cols]), (C.c_int32 * max(k, 1))(*is_i64),
            ptr_array([0 if v is None else v.data_ptr() for v in cols_valid])]
    a.cols, a.col_is_i64, a.col_valid = keep
    a.out_key, a.out_start, a.out_mean, a.out_n = _p(out_key), _p(out_start), _p(mean), _p(cnt)
    a.out_min, a.out_max, a.out_sum, a.out_stddev, a.out_count = _p(mn), _p(mx), _p(sm), _p(sd), _p(count)
    ws = _workspace(lib().tmp_grouped_stats_workspace_bytes(C.byref(a)), dev)
    check(lib().tmp_grouped_stats(C.byref(a), ws.data_ptr(), ws.numel(), _stream()), "tmp_grouped_stats")
    nb = int(count.item())

    def cell(t, c):
        return t[c, :nb] if is_i64[c] else t[c, :nb].view(torch.float64)

    return dict(n=nb, key=out_key[:nb], start=out_start[:nb], mean=[mean[c, :nb] for c in range(k)],
                count=[cnt[c, :nb] for c in range(k)], min=[cell(mn, c) for c in range(k)],
                max=[cell(mx, c) for c in range(k)], sum=[cell(sm, c) for c in range(k)],
                stddev=[sd[c, :nb] for c in range(k)], is_i64=is_i64)


FILL_CODES = {"zero": 0, "null": 1, "bfill": 2, "ffill": 3, "linear": 4}


def interpolate(key, ts, n_keys, freq_us, method, cols, cols_valid):
    """Interpolation core (python/tempo/interpol.py:260-378) on a sampled table.  cols: float64
    device tensors (at most 16).  Returns dict(n, src, key, ts, ts_interp, val[c], valid[c],
    interp[c]) with rows in (key, ts) order."""
    n = ts.numel()
    dev = ts.device
    _dev(ts, torch.int64, "ts")
    _dev(key, torch.int32, "key", n)
    if len(cols) > _lib.TMP_MAX_MASKS:
        raise NotImplementedError(f"at most {_lib.TMP_MAX_MASKS} target columns per call")
    for i, c in enumerate(cols):
        _dev(c, torch.float64, f"cols[{i}]", n)
        _dev(cols_valid[i], torch.uint8, f"cols_valid[{i}]", n)
    k = len(cols)
    count = torch.zeros(1, dtype=torch.int64, device=dev)
    a = _lib.InterpolateArgs()
    a.n, a.key, a.n_keys, a.ts, a.freq_us = n, _p(key), int(n_keys), _p(ts), int(freq_us)
    a.method, a.n_cols = FILL_CODES[method], k
    keep = [ptr_array([c.data_ptr() for c in cols]), ptr_array([0 if v is None else v.data_ptr() for v in cols_valid])]
    a.cols, a.col_valid = keep
    a.out_count = _p(count)
    ws = _workspace(lib().tmp_interpolate_workspace_bytes(C.byref(a)), dev)
    # first call sizes the result (capacity 0), the second one fills it
    a.capacity = 0
    rc = lib().tmp_interpolate(C.byref(a), ws.data_ptr(), ws.numel(), _stream())
    if rc not in (0, _lib.TMP_ECAPACITY):
        check(rc, "tmp_interpolate")
    m = int(count.item())
    cap = max(m, 1)
    src = torch.empty(cap, dtype=torch.int32, device=dev)
    okey = torch.empty(cap, dtype=torch.int32, device=dev)
    ots = torch.empty(cap, dtype=torch.int64, device=dev)
    tsi = torch.empty(cap, dtype=torch.uint8, device=dev)
    val = torch.empty((max(k, 1), cap), dtype=torch.float64, device=dev)
    valid = torch.empty((max(k, 1), cap), dtype=torch.uint8, device=dev)
    interp = torch.empty((max(k, 1), cap), dtype=torch.uint8, device=dev)
    if m:
        a.capacity = cap
        a.out_src, a.out_key, a.out_ts, a.out_ts_interp = _p(src), _p(okey), _p(ots), _p(tsi)
        a.out_val, a.out_valid, a.out_interp = _p(val), _p(valid), _p(interp)
        check(lib().tmp_interpolate(C.byref(a), ws.data_ptr(), ws.numel(), _stream()), "tmp_interpolate")
    return dict(n=m, src=src[:m], key=okey[:m], ts=ots[:m], ts_interp=tsi[:m], val=[val[c, :m] for c in range(k)],
                valid=[valid[c, :m] for c in range(k)], interp=[interp[c, :m] for c in range(k)])


def _route_args(a, key, n_parts, owner_table, slices, ts, halo):
    n = key.numel()
    _dev(key, torch.int32, "key")
    _dev(owner_table, torch.int32, "owner_table")
    a.n, a.key, a.n_parts = n, _p(key), int(n_parts)
    a.owner_table, a.n_keys = _p(owner_table), 0 if owner_table is None else owner_table.numel()
    if slices is not None:
        _dev(ts, torch.int64, "ts", n)
        for name in ("split_slices", "split_first", "slice_owner"):
            _dev(slices[name], torch.int32, name)
        a.ts, a.ts_is_seconds, a.v_is_int = _p(ts), int(bool(slices["ts_is_seconds"])), int(bool(slices["v_is_int"]))
        a.ts_partition_val, a.one_minus_fraction = float(slices["ts_partition_val"]), float(slices["one_minus_fraction"])
        a.bracket_min, a.bracket_count = int(slices["bracket_min"]), int(slices["bracket_count"])
        a.n_split = slices["split_slices"].numel()
        a.split_slices, a.split_first, a.slice_owner = (_p(slices["split_slices"]), _p(slices["split_first"]),
                                                        _p(slices["slice_owner"]))
        a.halo = int(bool(halo))
    elif halo:
        raise ValueError("halo routing needs sliced keys")


def hash_partition(key, n_parts, owner_table=None, slices=None, ts=None, halo=False, static_scratch=False):
    """Bucket rows by owner: owner_table[key] when a table (int32[n_keys], device) is given, else
    hash(key) % n_parts.  Returns (perm int32[n], counts int64[n_parts]).

    slices (with ts): time-sliced ownership of hot keys for the skew AS-OF join -- a dict with
    ts_partition_val, v_is_int, one_minus_fraction, ts_is_seconds, bracket_min, bracket_count and
    the device int32 tensors split_slices, split_first, slice_owner (include/tempo_hip.h,
    tmp_partition_args).  halo=True routes the copies of carried-over rows instead of the rows;
    only the first counts.sum() entries of perm are meaningful then.

    static_scratch=True goes through tmp_hash_partition_static: no workspace is allocated, the
    library's built-in device array serves instead (at most _lib.TMP_STATIC_PARTITION_MAX_ROWS
    rows; more raise)."""
    perm = torch.empty(key.numel(), dtype=torch.int32, device=key.device)
    counts = torch.zeros(n_parts, dtype=torch.int64, device=key.device)
    a = PartitionArgs()
    _route_args(a, key, n_parts, owner_table, slices, ts, halo)
    a.out_perm, a.out_counts = _p(perm), _p(counts)
    if static_scratch:
        check(lib().tmp_hash_partition_static(C.byref(a), _stream()), "tmp_hash_partition_static")
        return perm, counts
    ws = _workspace(lib().tmp_hash_partition_workspace_bytes(C.byref(a)), key.device)
    check(lib().tmp_hash_partition(C.byref(a), ws.data_ptr(), ws.numel(), _stream()),
          "tmp_hash_partition")
    return perm, counts


def partition_move(key, columns, n_parts, owner_table=None, slices=None, ts=None, halo=False):
    """Route rows like hash_partition and move `columns` (1-, 4- or 8-byte elements, at most 12)
    into bucketed order in the same call (n_parts <= 64).  Returns ([bucketed columns],
    counts int64[n_parts]); part r's rows sit at [sum(counts[:r]), sum(counts[:r+1])) in input
    order.  In halo mode the outputs hold just the copies (counts.sum() rows)."""
    n = key.numel()
    for i, c in enumerate(columns):
        _dev(c, c.dtype, f"columns[{i}]", n)
        if c.element_size() not in (1, 4, 8):
            raise TypeError("partition_move columns must have 1-, 4- or 8-byte elements")
    counts = torch.zeros(n_parts, dtype=torch.int64, device=key.device)
    capacity = max(1024, n // 64) if halo else n
    while True:
        outs = [torch.empty(capacity, dtype=c.dtype, device=c.device) for c in columns]
        a = _lib.PartitionMoveArgs()
        _route_args(a.route, key, n_parts, owner_table, slices, ts, halo)
        a.route.out_counts = _p(counts)
        keep = [ptr_array([c.data_ptr() for c in columns]), ptr_array([c.data_ptr() for c in outs]),
                (C.c_int32 * max(len(columns), 1))(*[c.element_size() for c in columns])]
        a.n_cols, a.cols, a.out_cols, a.col_bytes = len(columns), C.cast(keep[0], C.c_void_p), \
            C.cast(keep[1], C.c_void_p), C.cast(keep[2], C.c_void_p)
        a.capacity = capacity
        ws = _workspace(lib().tmp_partition_move_workspace_bytes(C.byref(a)), key.device)
        rc = lib().tmp_partition_move(C.byref(a), ws.data_ptr(), ws.numel(), _stream())
        if rc == _lib.TMP_ECAPACITY and capacity < n:
            capacity = n if not halo else min(n, int(counts.sum().item()))
            continue
        check(rc, "tmp_partition_move")
        if halo:
            m = int(counts.sum().item())
            outs = [o[:m] for o in outs]
        return outs, counts


def wire_code(encode, key, key_bits, key_word, masks, ints, int_min, int_bytes, wire_lo, wire_hi):
    """tmp_wire_encode / tmp_wire_decode over device tensors: key (int32) + masks (uint8 0/1 columns)
    <-> key_word (uint8 / int16 / int32: key | flags << key_bits), ints[k] (int64) <-> wire_lo[k]
    (int32, the low four bytes of ints[k] - int_min[k]) and wire_hi[k] (uint8 / int16 or None).
    encode=True reads the first of each pair and writes the second, False the other way round."""
    n = key.numel()
    if len(masks) > _lib.TMP_WIRE_MAX_MASKS or len(ints) > _lib.TMP_WIRE_MAX_INTS:
        raise NotImplementedError("too many columns for one tmp_wire call")
    a = _lib.WireArgs()
    a.n, a.key, a.key_bits, a.key_bytes = n, _p(key), int(key_bits), key_word.element_size()
    a.n_masks, a.n_ints, a.wire_key = len(masks), len(ints), _p(key_word)
    for m, t in enumerate(masks):
        _dev(t, torch.uint8, f"masks[{m}]", n)
        a.masks[m] = _p(t)
    for k, t in enumerate(ints):
        _dev(t, torch.int64, f"ints[{k}]", n)
        _dev(wire_lo[k], torch.int32, f"wire_lo[{k}]", n)
        a.ints[k], a.int_min[k], a.int_bytes[k] = _p(t), int(int_min[k]), int(int_bytes[k])
        a.wire_lo[k] = _p(wire_lo[k])
        a.wire_hi[k] = None if wire_hi[k] is None else _p(wire_hi[k])
    _dev(key, torch.int32, "key")
    fn = lib().tmp_wire_encode if encode else lib().tmp_wire_decode
    check(fn(C.byref(a), _stream()), "tmp_wire_encode" if encode else "tmp_wire_decode")


def key_counts(key, n_keys, stride=1):
    """Rows per key id over every stride-th row: int64[n_keys] on the device."""
    _dev(key, torch.int32, "key")
    counts = torch.empty(int(n_keys), dtype=torch.int64, device=key.device)
    a = _lib.KeyCountsArgs(key.numel(), _p(key), int(n_keys), int(stride), _p(counts))
    check(lib().tmp_key_counts(C.byref(a), None, 0, _stream()), "tmp_key_counts")
    return counts


_ARROW_KINDS = {"i32": (_lib.ARROW_I32, torch.int32), "i64": (_lib.ARROW_I64, torch.int64),
                "f64": (_lib.ARROW_F64, torch.float64), "i64_raw": (_lib.ARROW_I64_RAW, torch.int64)}


def _arrow_range_check(arr, kind):
    """The conversions that can lose values -- uint64 into int64, 64-bit integers into int32,
    second / millisecond timestamps scaled to microseconds -- are refused when a value of the
    column does not survive them (Spark would keep the value or raise; wrapping silently is
    neither).  One min / max over the column, and only for those types."""
    import pyarrow as pa
    import pyarrow.compute as pc
    t = arr.type
    if pa.types.is_dictionary(t):
        t = t.index_type
    lo, hi = -2 ** 63, 2 ** 63 - 1
    scale = 1
    if kind == "i32" and pa.types.is_integer(t) and t.bit_width == 64:
        lo, hi = -2 ** 31, 2 ** 31 - 1
    elif kind in ("i64", "i64_raw", "i32") and pa.types.is_uint64(t):
        if kind == "i32":
            lo, hi = -2 ** 31, 2 ** 31 - 1
    elif kind == "i64" and pa.types.is_timestamp(t) and t.unit in ("s", "ms"):
        scale = 1_000_000 if t.unit == "s" else 1_000
    else:
        return
    if len(arr) == 0 or arr.null_count == len(arr):
        return
    values = arr.indices if pa.types.is_dictionary(arr.type) and hasattr(arr, "indices") else arr
    if pa.types.is_timestamp(t):
        values = values.cast(pa.int64())
    mm = pc.min_max(values)
    vmin, vmax = mm["min"].as_py(), mm["max"].as_py()
    if vmin * scale < lo or vmax * scale > hi:
        raise OverflowError(f"arrow_import: column of type {arr.type} holds values ({vmin} .. {vmax}) that do not "
                            f"fit the device column ({kind}) after conversion")


def arrow_import(arr, kind, device=None):
    """A pyarrow Array or ChunkedArray to a device column through the Arrow C data interface: the
    value buffers are copied from where Arrow holds them, chunk by chunk, into one device tensor.

    kind: "i32" | "i64" (timestamps -> microseconds) | "i64_raw" (timestamps as stored) | "f64".
    Dictionary arrays give their indices.  Returns (values, valid uint8 or None, null_count)."""
    code, dtype = _ARROW_KINDS[kind]
    _arrow_range_check(arr, kind)
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    chunks = arr.chunks if hasattr(arr, "chunks") else [arr]
    n = sum(len(c) for c in chunks)
    any_nulls = any(c.null_count for c in chunks)
    out = torch.empty(n, dtype=dtype, device=dev)
    valid = torch.empty(n, dtype=torch.uint8, device=dev) if any_nulls else None
    total_nulls, at, exported = 0, 0, []
    try:
        for c in chunks:
            m = len(c)
            if m == 0:
                continue
            c_arr, c_sch = _lib.ArrowArray(), _lib.ArrowSchema()
            c._export_to_c(C.addressof(c_arr), C.addressof(c_sch))
            exported.append((c_arr, c_sch))
            ws = _workspace(lib().tmp_arrow_import_workspace_bytes(C.byref(c_arr), C.byref(c_sch)), dev)
            nulls = C.c_int64(0)
            check(lib().tmp_arrow_import(C.byref(c_arr), C.byref(c_sch), code, out[at:].data_ptr(),
                                         None if valid is None else valid[at:].data_ptr(), C.byref(nulls),
                                         ws.data_ptr(), ws.numel(), _stream()), "tmp_arrow_import")
            total_nulls += nulls.value
            at += m
    finally:
        # the copies read the Arrow buffers: let the stream pass them before they are released
        torch.cuda.current_stream().synchronize()
        for c_arr, c_sch in exported:
            if c_arr.release:
                c_arr.release(C.byref(c_arr))
            if c_sch.release:
                c_sch.release(C.byref(c_sch))
    return out, valid, total_nulls


def write_layout(key, ts, n_keys):
    """Derived columns and row order of a date-partitioned, (key, time)-clustered table
    (python/tempo/io.py:29-39).

    Returns (event_dt int32[n] days since the epoch, event_time float64[n] HHMMSS.ffffff,
    perm int32[n] = input row at position p of the (event_dt, key, ts) order,
    dt_range int32[2] = first and last event_dt).  key may be None (no partition columns).
    """
    n = ts.numel()
    _dev(ts, torch.int64, "ts")
    _dev(key, torch.int32, "key", n)
    dev = ts.device
    event_dt = torch.empty(n, dtype=torch.int32, device=dev)
    event_time = torch.empty(n, dtype=torch.float64, device=dev)
    perm = torch.empty(n, dtype=torch.int32, device=dev)
    dt_range = torch.zeros(2, dtype=torch.int32, device=dev)
    a = _lib.WriteLayoutArgs(n, _p(key), int(n_keys), _p(ts), _p(event_dt), _p(event_time), _p(perm),
                             _p(dt_range))
    ws = _workspace(lib().tmp_write_layout_workspace_bytes(C.byref(a)), dev)
    check(lib().tmp_write_layout(C.byref(a), ws.data_ptr(), ws.numel(), _stream()),
          "tmp_write_layout")
    return event_dt, event_time, perm, dt_range
