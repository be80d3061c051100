"""ctypes binding of libtempo_hip.so (declarations: include/tempo_hip.h).

There is no CPU fallback: loading fails loudly when the library has not been built
(`make -C tempo_amd/csrc`, or `python -c "import __graft_entry__ as g; g.build()"`).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# TEMPO_HIP_LIB: another build of the same library (kernel experiments); the default is the in-tree one
LIB_PATH = os.environ.get("TEMPO_HIP_LIB") or os.path.join(_HERE, "libtempo_hip.so")

TMP_MAX_MASKS = 16
TMP_MAX_JOIN_COLS = 16
TMP_MAX_TAPS = 1024
TMP_MAX_FEATURES = 64
TMP_MAX_COLS = 64
TMP_EINVAL = -1
TMP_ECAPACITY = -5
TMP_EWIDE = -6
TMP_STATIC_PARTITION_MAX_ROWS = 1048576
FUNC_CODES = {"floor": 0, "min": 1, "max": 2, "mean": 3, "ceil": 4}

_vp = C.c_void_p


class SortArgs(C.Structure):
    _fields_ = [("n", C.c_int64), ("key", _vp), ("n_keys", C.c_int32), ("ts", _vp),
                ("ts_unit", C.c_int32), ("seq", _vp), ("seq_valid", _vp), ("out_perm", _vp)]


class AsofArgs(C.Structure):
    _fields_ = [("n_left", C.c_int64), ("n_right", C.c_int64), ("key_left", _vp),
                ("ts_left", _vp), ("key_right", _vp), ("ts_right", _vp), ("seq_right", _vp),
                ("seq_right_valid", _vp), ("n_keys", C.c_int32), ("ts_is_seconds", C.c_int32),
                ("n_masks", C.c_int32), ("masks", C.POINTER(_vp)), ("skew", C.c_int32),
                ("v_is_int", C.c_int32), ("ts_partition_val", C.c_double),
                ("one_minus_fraction", C.c_double), ("out_idx", _vp), ("sorted_output", C.c_int32),
                ("out_perm", _vp), ("out_key", _vp), ("out_ts", _vp),
                ("n_left_cols", C.c_int32), ("left_cols", C.POINTER(_vp)), ("out_left_cols", C.POINTER(_vp)),
                ("n_right_cols", C.c_int32), ("right_cols", C.POINTER(_vp)),
                ("right_col_rule", C.POINTER(C.c_int32)), ("right_col_valid", C.POINTER(C.c_int32)),
                ("out_right_cols", C.POINTER(_vp)), ("out_right_valid", C.POINTER(_vp)),
                ("out_missing", _vp), ("left_in_place", C.POINTER(C.c_int32))]


class RangeStatsArgs(C.Structure):
    _fields_ = [("n", C.c_int64), ("key", _vp), ("n_keys", C.c_int32), ("ts", _vp),
                ("ts_is_seconds", C.c_int32), ("x_is_i64", C.c_int32), ("x", _vp),
                ("x_valid", _vp), ("range_back_secs", C.c_int64), ("out_perm", _vp),
                ("out_mean", _vp), ("out_count", _vp), ("out_min", _vp), ("out_max", _vp),
                ("out_sum", _vp), ("out_stddev", _vp), ("out_zscore", _vp), ("out_valid", _vp),
                ("any_order", C.c_int32)]


TMP_MAX_STATS_COLS = 4


class RangeStatsColsArgs(C.Structure):
    _fields_ = [("n", C.c_int64), ("key", _vp), ("n_keys", C.c_int32), ("ts", _vp),
                ("ts_is_seconds", C.c_int32), ("range_back_secs", C.c_int64), ("n_cols", C.c_int32),
                ("x", C.POINTER(_vp)), ("x_valid", C.POINTER(_vp)), ("x_is_i64", C.POINTER(C.c_int32)),
                ("out_perm", _vp), ("out_mean", C.POINTER(_vp)), ("out_count", C.POINTER(_vp)),
                ("out_min", C.POINTER(_vp)), ("out_max", C.POINTER(_vp)), ("out_sum", C.POINTER(_vp)),
                ("out_stddev", C.POINTER(_vp)), ("out_zscore", C.POINTER(_vp)), ("out_valid", C.POINTER(_vp))]


TMP_WIRE_MAX_MASKS = 8
TMP_WIRE_MAX_INTS = 8


class WireArgs(C.Structure):
    _fields_ = [("n", C.c_int64), ("key", _vp), ("key_bits", C.c_int32), ("key_bytes", C.c_int32),
                ("n_masks", C.c_int32), ("n_ints", C.c_int32), ("masks", _vp * TMP_WIRE_MAX_MASKS),
                ("ints", _vp * TMP_WIRE_MAX_INTS), ("int_min", C.c_int64 * TMP_WIRE_MAX_INTS),
                ("int_bytes", C.c_int32 * TMP_WIRE_MAX_INTS), ("wire_key", _vp),
                ("wire_lo", _vp * TMP_WIRE_MAX_INTS), ("wire_hi", _vp * TMP_WIRE_MAX_INTS)]


class EmaArgs(C.Structure):
    _fields_ = [("n", C.c_int64), ("key", _vp), ("n_keys", C.c_int32), ("ts", _vp),
                ("ts_is_seconds", C.c_int32), ("seq", _vp), ("seq_valid", _vp), ("x", _vp),
                ("x_valid", _vp), ("n_taps", C.c_int32), ("weights", C.POINTER(C.c_double)),
                ("out_perm", _vp), ("out_ema", _vp)]


class LookbackArgs(C.Structure):
    _fields_ = [("n", C.c_int64), ("key", _vp), ("n_keys", C.c_int32), ("ts", _vp),
                ("ts_is_seconds", C.c_int32), ("seq", _vp), ("seq_valid", _vp),
                ("n_feat", C.c_int32), ("x", C.POINTER(_vp)), ("x_valid", C.POINTER(_vp)),
                ("lookback", C.c_int32), ("exact_size", C.c_int32), ("out_rows", _vp),
                ("out_len", _vp), ("out_feat", _vp), ("out_feat_valid", _vp), ("out_count", _vp)]


class ResampleArgs(C.Structure):
    _fields_ = [("n", C.c_int64), ("key", _vp), ("n_keys", C.c_int32), ("ts", _vp),
                ("dur_us", C.c_int64), ("func", C.c_int32), ("n_cols", C.c_int32),
                ("cols", C.POINTER(_vp)), ("col_is_i64", C.POINTER(C.c_int32)),
                ("col_valid", C.POINTER(_vp)), ("out_key", _vp), ("out_start", _vp),
                ("out_f64", _vp), ("out_i64", _vp), ("out_valid", _vp), ("out_row", _vp),
                ("out_count", _vp)]


class GroupedStatsArgs(C.Structure):
    _fields_ = [("n", C.c_int64), ("key", _vp), ("n_keys", C.c_int32), ("ts", _vp),
                ("dur_us", C.c_int64), ("n_cols", C.c_int32), ("cols", C.POINTER(_vp)),
                ("col_is_i64", C.POINTER(C.c_int32)), ("col_valid", C.POINTER(_vp)),
                ("out_key", _vp), ("out_start", _vp), ("out_mean", _vp), ("out_n", _vp),
                ("out_min", _vp), ("out_max", _vp), ("out_sum", _vp), ("out_stddev", _vp),
                ("out_count", _vp)]


class InterpolateArgs(C.Structure):
    _fields_ = [("n", C.c_int64), ("key", _vp), ("n_keys", C.c_int32), ("ts", _vp),
                ("freq_us", C.c_int64), ("method", C.c_int32), ("n_cols", C.c_int32),
                ("cols", C.POINTER(_vp)), ("col_valid", C.POINTER(_vp)), ("capacity", C.c_int64),
                ("out_src", _vp), ("out_key", _vp), ("out_ts", _vp), ("out_ts_interp", _vp),
                ("out_val", _vp), ("out_valid", _vp), ("out_interp", _vp), ("out_count", _vp)]


class PartitionArgs(C.Structure):
    _fields_ = [("n", C.c_int64), ("key", _vp), ("n_parts", C.c_int32), ("out_perm", _vp),
                ("out_counts", _vp), ("owner_table", _vp), ("n_keys", C.c_int32),
                ("ts", _vp), ("ts_is_seconds", C.c_int32), ("v_is_int", C.c_int32),
                ("ts_partition_val", C.c_double), ("one_minus_fraction", C.c_double),
                ("bracket_min", C.c_int64), ("bracket_count", C.c_int64), ("n_split", C.c_int32),
                ("split_slices", _vp), ("split_first", _vp), ("slice_owner", _vp), ("halo", C.c_int32)]


class PartitionMoveArgs(C.Structure):
    _fields_ = [("route", PartitionArgs), ("n_cols", C.c_int32), ("cols", _vp), ("col_bytes", _vp),
                ("out_cols", _vp), ("capacity", C.c_int64)]


class KeyCountsArgs(C.Structure):
    _fields_ = [("n", C.c_int64), ("key", _vp), ("n_keys", C.c_int32), ("stride", C.c_int32),
                ("out_counts", _vp)]


class WriteLayoutArgs(C.Structure):
    _fields_ = [("n", C.c_int64), ("key", _vp), ("n_keys", C.c_int32), ("ts", _vp),
                ("out_event_dt", _vp), ("out_event_time", _vp), ("out_perm", _vp),
                ("out_dt_range", _vp)]


class ArrowSchema(C.Structure):
    """struct ArrowSchema of the Arrow C data interface."""


ArrowSchema._fields_ = [("format", C.c_char_p), ("name", C.c_char_p), ("metadata", _vp), ("flags", C.c_int64),
                        ("n_children", C.c_int64), ("children", _vp), ("dictionary", C.POINTER(ArrowSchema)),
                        ("release", C.CFUNCTYPE(None, C.POINTER(ArrowSchema))), ("private_data", _vp)]


class ArrowArray(C.Structure):
    """struct ArrowArray of the Arrow C data interface."""


ArrowArray._fields_ = [("length", C.c_int64), ("null_count", C.c_int64), ("offset", C.c_int64),
                       ("n_buffers", C.c_int64), ("n_children", C.c_int64), ("buffers", C.POINTER(_vp)),
                       ("children", _vp), ("dictionary", C.POINTER(ArrowArray)),
                       ("release", C.CFUNCTYPE(None, C.POINTER(ArrowArray))), ("private_data", _vp)]

ARROW_I32, ARROW_I64, ARROW_F64, ARROW_I64_RAW = 0, 1, 2, 3


# every symbol include/tempo_hip.h declares: name -> (restype, argtypes)
_OP = lambda argt: (C.c_int, [C.POINTER(argt), _vp, C.c_size_t, _vp])
_WS = lambda argt: (C.c_size_t, [C.POINTER(argt)])
SYMBOLS = {
    "tmp_abi_version": (C.c_int, []),
    "tmp_last_error": (C.c_char_p, []),
    "tmp_sort_order_workspace_bytes": _WS(SortArgs),
    "tmp_sort_order": _OP(SortArgs),
    "tmp_asof_join_workspace_bytes": _WS(AsofArgs),
    "tmp_asof_join": _OP(AsofArgs),
    "tmp_gather": (C.c_int, [_vp, _vp, C.c_int32, _vp, C.c_int64, C.c_int64, _vp, _vp, _vp]),
    "tmp_range_stats_workspace_bytes": _WS(RangeStatsArgs),
    "tmp_range_stats_wide_workspace_bytes": _WS(RangeStatsArgs),
    "tmp_range_stats": _OP(RangeStatsArgs),
    "tmp_range_stats_cols_workspace_bytes": _WS(RangeStatsColsArgs),
    "tmp_range_stats_cols_wide_workspace_bytes": _WS(RangeStatsColsArgs),
    "tmp_range_stats_cols": _OP(RangeStatsColsArgs),
    "tmp_ema_workspace_bytes": _WS(EmaArgs),
    "tmp_ema": _OP(EmaArgs),
    "tmp_lookback_workspace_bytes": _WS(LookbackArgs),
    "tmp_lookback": _OP(LookbackArgs),
    "tmp_resample_workspace_bytes": _WS(ResampleArgs),
    "tmp_resample": _OP(ResampleArgs),
    "tmp_grouped_stats_workspace_bytes": _WS(GroupedStatsArgs),
    "tmp_grouped_stats": _OP(GroupedStatsArgs),
    "tmp_interpolate_workspace_bytes": _WS(InterpolateArgs),
    "tmp_interpolate": _OP(InterpolateArgs),
    "tmp_hash_partition_workspace_bytes": _WS(PartitionArgs),
    "tmp_hash_partition": _OP(PartitionArgs),
    "tmp_hash_partition_static": (C.c_int, [C.POINTER(PartitionArgs), _vp]),
    "tmp_wire_encode": (C.c_int, [C.POINTER(WireArgs), _vp]),
    "tmp_wire_decode": (C.c_int, [C.POINTER(WireArgs), _vp]),
    "tmp_partition_move_workspace_bytes": _WS(PartitionMoveArgs),
    "tmp_partition_move": _OP(PartitionMoveArgs),
    "tmp_key_counts_workspace_bytes": _WS(KeyCountsArgs),
    "tmp_key_counts": _OP(KeyCountsArgs),
    "tmp_key_owner": (C.c_int32, [C.c_int32, C.c_int32]),
    "tmp_write_layout_workspace_bytes": _WS(WriteLayoutArgs),
    "tmp_write_layout": _OP(WriteLayoutArgs),
    "tmp_arrow_import_workspace_bytes": (C.c_size_t, [C.POINTER(ArrowArray), C.POINTER(ArrowSchema)]),
    "tmp_arrow_import": (C.c_int, [C.POINTER(ArrowArray), C.POINTER(ArrowSchema), C.c_int32, _vp, _vp,
                                   C.POINTER(C.c_int64), _vp, C.c_size_t, _vp]),
    "tmp_profile_enable": (C.c_int, [C.c_int]),
    "tmp_profile_reset": (C.c_int, []),
    "tmp_profile_count": (C.c_int, []),
    "tmp_profile_dropped": (C.c_int, []),
    "tmp_profile_get": (C.c_int, [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_float)]),
    "tmp_profile_set_tag": (C.c_int, [C.c_int]),
    "tmp_profile_tag": (C.c_int, [C.c_int]),
}

_LIB = None


class TempoHipError(RuntimeError):
    pass


def lib():
    """The loaded library; raises if it is missing (no fallback path exists)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise TempoHipError(
                f"{LIB_PATH} not found: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
                "(hipcc, gfx950). tempo_amd has no CPU fallback.")
        # torch must be loaded first: the library has to bind to the HIP runtime torch already
        # brought into the process, or device pointers and streams would belong to another one.
        import torch  # noqa: F401
        handle = C.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SYMBOLS.items():
            fn = getattr(handle, name)  # AttributeError if the build lacks a declared symbol
            fn.restype = restype
            fn.argtypes = argtypes
        if handle.tmp_abi_version() != 2:
            raise TempoHipError("libtempo_hip.so ABI version mismatch")
        _LIB = handle
    return _LIB


def check(rc, what):
    if rc != 0:
        msg = lib().tmp_last_error().decode("utf-8", "replace")
        if rc == -4:
            raise NotImplementedError(f"{what}: {msg}")
        if rc == -1:
            raise ValueError(f"{what}: {msg}")
        raise TempoHipError(f"{what} failed with code {rc}: {msg}")


def ptr_array(ptrs):
    arr = (_vp * max(len(ptrs), 1))()
    for i, p in enumerate(ptrs):
        arr[i] = p
    return arr


def profile_enable(on=True):
    check(lib().tmp_profile_enable(int(on)), "tmp_profile_enable")


def profile_reset():
    lib().tmp_profile_reset()


def profile_tag(tag):
    lib().tmp_profile_set_tag(int(tag))


def profile_dropped():
    """Launches since the last reset that found the event pool full and were not recorded."""
    return int(lib().tmp_profile_dropped())


def profile_read():
    """[(kernel name, tag, ms), ...] for every launch since the last reset, in launch order
    (the stream must have been synchronised)."""
    out = []
    name, ms = C.c_char_p(), C.c_float()
    for i in range(lib().tmp_profile_count()):
        check(lib().tmp_profile_get(i, C.byref(name), C.byref(ms)), "tmp_profile_get")
        out.append((name.value.decode(), lib().tmp_profile_tag(i), ms.value))
    return out
