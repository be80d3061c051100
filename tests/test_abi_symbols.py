"""libtempo_hip.so loads (no GPU needed) and exports every function include/tempo_hip.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "tempo_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tmp_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_operator_set():
    names = _declared()
    for must in ("tmp_asof_join", "tmp_range_stats", "tmp_ema", "tmp_lookback", "tmp_resample", "tmp_sort_order",
                 "tmp_gather", "tmp_hash_partition"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from tempo_amd import _lib
    handle = _lib.lib()  # binds every entry of _lib.SYMBOLS, raising if one is missing
    for name in _declared():
        assert hasattr(handle, name), f"{name} is declared in tempo_hip.h but not exported"
        assert name in _lib.SYMBOLS, f"{name} has no ctypes signature in tempo_amd/_lib.py"
    assert handle.tmp_abi_version() == 2


def test_argument_errors_need_no_gpu():
    from tempo_amd import _lib
    h = _lib.lib()
    a = _lib.AsofArgs()
    a.n_left = -1
    assert h.tmp_asof_join(ctypes.byref(a), None, 0, None) == -1
    assert b"row counts" in h.tmp_last_error()
    r = _lib.RangeStatsArgs()
    r.n, r.range_back_secs = 5, -3
    assert h.tmp_range_stats(ctypes.byref(r), None, 0, None) == -1
    assert h.tmp_asof_join_workspace_bytes(ctypes.byref(_lib.AsofArgs())) > 0
    assert 0 <= h.tmp_key_owner(12345, 8) < 8
