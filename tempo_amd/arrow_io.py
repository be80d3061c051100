"""Arrow edge of the hot path: pyarrow Tables in, pyarrow Tables out, no pandas in between.

What a Spark-side caller (`mapInArrow` / `toArrow` per key-hash partition, or any Arrow
producer) hands over and takes back for the two headline operators.  Columns go to HBM from
the Arrow buffers themselves, through the Arrow C data interface (tmp_arrow_import: one
host-to-device copy per chunk out of Arrow's own buffer, type conversion and the expansion of
the validity bitmap on the device): partition keys are dictionary-encoded by Arrow (one
dictionary shared by both sides of a join) and their indices imported, `timestamp[*]` columns
become int64 microseconds, numeric columns become float64 / int64 with the Arrow validity
bitmap as the NULL mask.

  asof_join(left, right, ...)   TSDF.asofJoin (python/tempo/tsdf.py:463-560) through the fused
                                device join: the joined table in (key, ts) order
  range_stats(table, ...)       TSDF.withRangeStats (python/tempo/tsdf.py:673-721)
"""
import concurrent.futures
import os

import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import torch

from . import columns as col
from . import ops


def _combined(arr):
    if isinstance(arr, pa.ChunkedArray):
        return arr.combine_chunks() if arr.num_chunks != 1 else arr.chunk(0)
    return arr


# Dictionary-encoding a string column and materialising one from dictionary codes are the two
# host-side passes that scale with the row count; Arrow runs each on one thread and releases the
# GIL meanwhile, so long columns are cut into slices that a few threads take side by side.
HOST_THREADS = max(1, int(os.environ.get("TEMPO_HOST_THREADS", "8")))
PARALLEL_MIN_ROWS = 1 << 18
_POOL = None


def _slices(n):
    """[(start, length)] of up to HOST_THREADS slices of n rows; one slice for short columns."""
    if n < PARALLEL_MIN_ROWS or HOST_THREADS == 1:
        return [(0, n)]
    step = -(-n // HOST_THREADS)
    return [(at, min(step, n - at)) for at in range(0, n, step)]


def _map(fn, items):
    global _POOL
    if len(items) <= 1:
        return [fn(x) for x in items]
    if _POOL is None:
        _POOL = concurrent.futures.ThreadPoolExecutor(max_workers=HOST_THREADS, thread_name_prefix="tempo-host")
    return list(_POOL.map(fn, items))


def dictionary_encode(arr, null_encoding="mask"):
    """pc.dictionary_encode(arr, null_encoding) as (dictionary, int32 indices): same dictionary (values
    in order of first appearance) and same indices, the indices as a ChunkedArray when the column was
    long enough to be encoded slice by slice."""
    arr = _combined(arr)
    parts = _slices(len(arr))
    if len(parts) == 1:
        enc = pc.dictionary_encode(arr, null_encoding=null_encoding)
        return enc.dictionary, enc.indices
    encs = _map(lambda sl: pc.dictionary_encode(arr.slice(*sl), null_encoding=null_encoding), parts)
    # values in order of first appearance over the slices in order = over the whole column
    dictionary = pc.unique(pa.concat_arrays([e.dictionary for e in encs]))

    def remap(e):
        to_global = pc.index_in(e.dictionary, value_set=dictionary).cast(pa.int32())
        return to_global.take(e.indices)   # NULL indices (null_encoding="mask") stay NULL
    return dictionary, pa.chunked_array(_map(remap, encs), type=pa.int32())


def dictionary_take(dictionary, indices):
    """dictionary.take(indices) -- the column a dictionary and its codes stand for -- as an Array, or
    as a ChunkedArray built slice by slice on several threads when it is long."""
    parts = _slices(len(indices))
    if len(parts) == 1:
        return dictionary.take(indices)
    taken = _map(lambda sl: dictionary.take(indices.slice(*sl)), parts)
    chunks = [c for t in taken for c in (t.chunks if isinstance(t, pa.ChunkedArray) else [t])]
    return pa.chunked_array(chunks, type=dictionary.type)


def encode_keys(tables, cols):
    """Dictionary-encode the partition columns of several tables with one shared dictionary.
    Returns ([int32 numpy codes per table], n_keys, decode) where decode(codes) -> list of
    Arrow arrays, one per partition column.  NULL keys form their own group (as partitionBy)."""
    if not cols:
        return [None for _ in tables], 1, (lambda codes: [])
    lens = [t.num_rows for t in tables]
    dense, dicts, col_codes = None, [], []
    for c in cols:
        whole = pa.concat_arrays([_combined(t[c]) for t in tables])
        dictionary, indices = dictionary_encode(whole, null_encoding="encode")
        codes = _combined(indices).to_numpy(zero_copy_only=False).astype(np.int64)
        dicts.append(dictionary)
        col_codes.append(codes)
        if dense is None:
            dense = codes
        else:
            # re-densified after every column: the running code stays below the row count, so
            # (rows x cardinality) cannot wrap int64 however many columns there are
            _, dense = np.unique(dense * max(len(dictionary), 1) + codes, return_inverse=True)
            dense = dense.astype(np.int64)
    if len(cols) > 1:
        n_keys = int(dense.max()) + 1 if len(dense) else 1
        # one row per key id, to read the key's per-column dictionary codes back from
        _, first_row = np.unique(dense, return_index=True)
    else:
        n_keys = max(len(dicts[0]), 1)
        first_row = None
    if n_keys >= 2 ** 31:
        raise NotImplementedError("more than 2^31 distinct partition keys")

    def decode(codes):
        codes = np.asarray(codes, dtype=np.int64)
        if first_row is None:
            return [dictionary_take(dicts[0], pa.array(codes, type=pa.int64()))]
        rows = first_row[codes]
        return [dictionary_take(dicts[j], pa.array(col_codes[j][rows], type=pa.int64())) for j in range(len(cols))]

    parts, at = [], 0
    for n in lens:
        parts.append(dense[at:at + n].astype(np.int32))
        at += n
    return parts, n_keys, decode


def encode_keys_dev(tables, cols):
    """encode_keys with the codes on the device: ([int32 device codes per table], n_keys, decode).
    One partition column: the dictionary indices Arrow computed are imported as they are."""
    if len(cols) != 1:
        parts, n_keys, decode = encode_keys(tables, cols)
        return [None if p is None else col.to_dev(p) for p in parts], n_keys, decode
    whole = pa.concat_arrays([_combined(t[cols[0]]) for t in tables])
    dictionary, indices = dictionary_encode(whole, null_encoding="encode")
    n_keys = max(len(dictionary), 1)
    codes, _, _ = ops.arrow_import(indices, "i32")
    parts, at = [], 0
    for t in tables:
        parts.append(codes[at:at + t.num_rows])
        at += t.num_rows
    return parts, n_keys, (lambda c: [dictionary_take(dictionary, pa.array(np.asarray(c, dtype=np.int64), type=pa.int64()))])


def timestamps_dev(arr):
    """(int64 device tensor of microseconds, Arrow type) of a timestamp column, imported from the
    Arrow buffers (nanoseconds floored, as Spark keeps microseconds); NULLs are rejected."""
    if not pa.types.is_timestamp(arr.type):
        raise TypeError(f"time index column must be an Arrow timestamp, got {arr.type}")
    if arr.null_count:
        raise ValueError("NULL timestamps are not supported in the time index column")
    return ops.arrow_import(arr, "i64")[0], arr.type


def timestamps_us(arr):
    """(int64 numpy microseconds, Arrow type) of a timestamp column; NULLs are rejected."""
    arr = _combined(arr)
    if not pa.types.is_timestamp(arr.type):
        raise TypeError(f"time index column must be an Arrow timestamp, got {arr.type}")
    if arr.null_count:
        raise ValueError("NULL timestamps are not supported in the time index column")
    raw = arr.cast(pa.int64()).to_numpy(zero_copy_only=False)
    unit = arr.type.unit
    if unit == "ns":
        raw = np.floor_divide(raw, 1000)  # Spark keeps microseconds; floor also before the epoch
    elif unit != "us":
        raw = raw * {"s": 1_000_000, "ms": 1_000}[unit]
    return raw, arr.type


def payload_to_device(arr):
    """(8-byte device tensor, uint8 validity tensor or None, decode) for a payload column of the
    fused join: integers, floats, timestamps and dictionary-encodable columns (strings).
    decode(values, ok) takes host arrays -- int64 / float64 values and a bool (or None) mask of the
    rows that are not NULL -- and returns the Arrow array of the column's own type."""
    t = arr.type
    if pa.types.is_floating(t) and t != pa.float16():
        vals, valid, _ = ops.arrow_import(arr, "f64")
        dec = lambda v, ok: _masked(v, pa.float64(), ok).cast(t)
    elif pa.types.is_integer(t):
        vals, valid, _ = ops.arrow_import(arr, "i64")
        dec = lambda v, ok: _masked(v, pa.int64(), ok).cast(t)
    elif pa.types.is_timestamp(t):
        vals, valid, _ = ops.arrow_import(arr, "i64_raw")
        dec = lambda v, ok: _masked(v, pa.int64(), ok).cast(t)
    else:
        dictionary, indices = dictionary_encode(arr)
        vals, valid, _ = ops.arrow_import(indices, "i64")
        dec = lambda v, ok: dictionary_take(dictionary, _masked(v, pa.int64(), ok))
    return vals, valid, dec


def _masked(values, typ, ok):
    """Arrow array over a host numpy array without copying it; `ok`: None (no NULLs), a bool array,
    or a (bit-packed validity bytes, null count) pair as to_host() returns it for a mask."""
    values = np.ascontiguousarray(values)
    n = len(values)
    if ok is None:
        return pa.Array.from_buffers(typ, n, [None, pa.py_buffer(values)], null_count=0)
    if isinstance(ok, tuple):
        bits, nulls = ok
        return pa.Array.from_buffers(typ, n, [pa.py_buffer(bits), pa.py_buffer(values)], null_count=int(nulls))
    return pa.array(values, type=typ, mask=~ok)


def pack_validity(valid):
    """Device uint8 mask (1 = not NULL) -> (bit-packed device bytes as Arrow lays a validity bitmap
    out, least significant bit first; NULL count as a device scalar)."""
    n = valid.numel()
    pad = (-n) % 8
    v = valid if pad == 0 else torch.cat([valid, valid.new_zeros(pad)])
    weights = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.int32, device=valid.device)
    bits = (v.view(-1, 8).to(torch.int32) * weights).sum(1).to(torch.uint8)
    return bits, n - valid.sum()


def to_host(tensors):
    """Device tensors -> host numpy arrays through page-locked staging buffers, all copies in flight
    together and one wait at the end (torch keeps freed page-locked blocks for the next call).  The
    arrays own their memory: Arrow buffers built over them keep it alive."""
    outs = []
    for t in tensors:
        if t is None:
            outs.append(None)
            continue
        h = torch.empty(t.shape, dtype=t.dtype, device="cpu", pin_memory=True)
        h.copy_(t, non_blocking=True)
        outs.append(h)
    torch.cuda.current_stream().synchronize()
    return [None if h is None else h.numpy() for h in outs]


def _device_permutable(t):
    return (pa.types.is_integer(t) or (pa.types.is_floating(t) and t != pa.float16()) or pa.types.is_timestamp(t))


def key_of(partition_cols, codes, decode):
    """permute_table's `key` argument for a table with exactly one partition column."""
    return (partition_cols[0], codes, decode) if len(partition_cols) == 1 and codes is not None else None


def permute_table(table, perm, key=None):
    """table.take(perm) with the fixed-width columns reordered on the device (one import, one gather,
    one copy back per column) and only the others -- strings, nested types -- taken on the host.
    perm: int32 device tensor.  key = (column name, int32 device codes, decode): that column is
    rebuilt from its dictionary codes, reordered on the device, instead of being taken on the host."""
    names, arrays, perm_host = [], [], None
    pending = []  # (position, values tensor, validity tensor or None, type)
    key_at = key_codes = None
    for c in table.column_names:
        arr = table[c]
        names.append(c)
        if key is not None and c == key[0] and table.num_rows:
            key_at, key_codes = len(arrays), ops.gather(key[1], perm, want_valid=False)[0]
            arrays.append(None)
        elif _device_permutable(arr.type) and table.num_rows:
            kind = "f64" if pa.types.is_floating(arr.type) else ("i64_raw" if pa.types.is_timestamp(arr.type) else "i64")
            vals, valid, _ = ops.arrow_import(arr, kind)
            moved = ops.gather(vals, perm, want_valid=False)[0]
            mvalid = None if valid is None else ops.gather(valid, perm, want_valid=False)[0]
            pending.append((len(arrays), moved, mvalid, arr.type))
            arrays.append(None)
        else:
            if perm_host is None:
                perm_host = pa.array(to_host([perm])[0])
            arrays.append(dictionary_take(arr, perm_host))
    flat = [key_codes] if key_codes is not None else []
    for _, moved, mvalid, _ in pending:
        flat.append(moved)
        if mvalid is not None:
            bits, nulls = pack_validity(mvalid)
            flat += [bits, nulls.reshape(1)]
    host = iter(to_host(flat))
    if key_codes is not None:
        arrays[key_at] = key[2](next(host))[0]
    for at, _, mvalid, typ in pending:
        v = next(host)
        ok = None if mvalid is None else (next(host), int(next(host)[0]))
        wide = pa.float64() if pa.types.is_floating(typ) else pa.int64()
        arrays[at] = _masked(v, wide, ok).cast(typ)
    return names, arrays


def spark_dtype(t):
    """The Spark SQL type name of an Arrow type, as the reference's dtype checks would see it
    (python/tempo/tsdf.py:697, python/tempo/resample.py:111)."""
    if pa.types.is_dictionary(t):
        t = t.value_type
    if pa.types.is_boolean(t):
        return "boolean"
    if pa.types.is_timestamp(t):
        return "timestamp"
    if pa.types.is_integer(t):
        return {8: "tinyint", 16: "smallint", 32: "int", 64: "bigint"}[t.bit_width]
    if pa.types.is_floating(t):
        return "float" if t.bit_width <= 32 else "double"
    return "string"


def asof_join(left, right, ts_col="event_ts", partition_cols=(), right_prefix="right", skip_nulls=True,
              ts_partition_val=None, fraction=0.5, left_prefix=None, right_ts_col=None, with_info=False):
    """AS-OF join of two Arrow tables.  Returns the joined table ordered by (partition, ts):
    partition columns, the left columns (`<left_prefix>_<name>` when a left prefix is given), then
    `<right_prefix>_<name>` for every right column ('' as a prefix keeps the name).

    Only keys, timestamps and the right table's columns travel to the device; the left table's
    rows follow by the row permutation the join returns -- or, when the left table arrives in
    (partition, ts) order already, stay where they are: its columns are then the result's, untouched.
    with_info: also return dict(missing=flags per rule, mask_of={right column: its NULL-mask slot},
    in_place=bool)."""
    partition_cols = list(partition_cols)
    right_ts_col = ts_col if right_ts_col is None else right_ts_col
    (key_l, key_r), n_keys, decode_keys = encode_keys_dev([left, right], partition_cols)
    ts_l, ts_type = timestamps_dev(left[ts_col])
    ts_r, ts_type_r = timestamps_dev(right[right_ts_col])
    if ts_type != ts_type_r:
        raise ValueError("left and right dataframe timestamp index columns should have same type")
    left_cols = [c for c in left.column_names if c not in partition_cols]
    right_cols = [c for c in right.column_names if c not in partition_cols]
    lname = {c: (c if left_prefix in (None, "") else f"{left_prefix}_{c}") for c in left_cols}
    rname = {c: (c if right_prefix == "" else f"{right_prefix}_{c}") for c in right_cols}
    clash = set(lname.values()) & set(rname.values())
    if clash:
        raise ValueError(f"duplicate column names after prefixing: {sorted(clash)}")
    renc = []
    for c in right_cols:
        if c == right_ts_col:  # the join's own time column: already on the device
            renc.append((ts_r, None, (lambda v, ok, t=ts_type_r: _masked(v, pa.int64(), ok)
                                      .cast(pa.timestamp("us", tz=t.tz)).cast(t))))
        else:
            renc.append(payload_to_device(right[c]))
    j = ops.asof_join_table(key_l, ts_l, key_r, ts_r, n_keys, [], [t for t, _, _ in renc], [v for _, v, _ in renc],
                            skip_nulls=skip_nulls, ts_partition_val=ts_partition_val, fraction=fraction,
                            left_by_reference=True)
    if j["in_place"]:
        lnames, larrays = list(left.column_names), [left[c] for c in left.column_names]
    else:
        lnames, larrays = permute_table(left, j["perm"], key_of(partition_cols, key_l, decode_keys))
    by_name = dict(zip(lnames, larrays))
    names = list(partition_cols) + [lname[c] for c in left_cols]
    arrays = [by_name[c] for c in partition_cols] + [by_name[c] for c in left_cols]
    flat = []
    for vals, ok in j["right"]:
        bits, nulls = pack_validity(ok)
        flat += [vals, bits, nulls.reshape(1)]
    host = iter(to_host(flat))
    for i, c in enumerate(right_cols):
        v, bits, nulls = next(host), next(host), int(next(host)[0])
        names.append(rname[c])
        arrays.append(renc[i][2](v, (bits, nulls)))
    out = pa.table(arrays, names=names)
    if not with_info:
        return out
    mask_of = {c: j["mask_of"][i] for i, c in enumerate(right_cols) if j["mask_of"][i] >= 0}
    return out, dict(missing=j["missing"].cpu().numpy(), mask_of=mask_of, in_place=j["in_place"], names=rname)


def range_stats(table, ts_col="event_ts", partition_cols=(), cols=(), range_back_secs=1000):
    """withRangeStats on an Arrow table: the input rows in (partition, ts) order plus
    mean_/count_/min_/max_/sum_/stddev_/zscore_<col> for every column in `cols` -- one device call
    for all of them (the rows are ordered once)."""
    partition_cols = list(partition_cols)
    (key_d,), n_keys, decode_keys = encode_keys_dev([table], partition_cols)
    ts_d, _ = timestamps_dev(table[ts_col])
    xs, xv, types = [], [], []
    for c in cols:
        arr = _combined(table[c])
        if not (pa.types.is_floating(arr.type) or pa.types.is_integer(arr.type)):
            raise TypeError(f"column {c} of type {arr.type} is not numeric")
        x, valid, _ = payload_to_device(arr)
        xs.append(x)
        xv.append(valid)
        types.append(arr.type)
    extra = {}
    if xs and table.num_rows:
        perm, outs = ops.range_stats_cols(key_d, ts_d, xs, xv, n_keys, range_back_secs)
    else:
        perm = (ops.sort_order(key_d, ts_d, n_keys, floor_seconds=True) if table.num_rows
                else torch.empty(0, dtype=torch.int32))
        outs = []
    n = table.num_rows
    flat = []
    for r in outs:
        flags = r["valid"]
        for bit in (1, 2, 4):
            bits, nulls = pack_validity((flags & bit) != 0)
            flat += [bits, nulls.reshape(1)]
        flat += [r[k] for k in ("mean", "count", "min", "max", "sum", "stddev", "zscore")]
    identity = n == 0 or bool((perm == torch.arange(n, dtype=torch.int32, device=perm.device)).all())
    host = iter(to_host(flat))
    for c, typ, r in zip(cols, types, outs):
        ok, sd_ok, z_ok = ((next(host), int(next(host)[0])) for _ in range(3))
        mean, count, mn, mx, sm, sd, zs = (next(host) for _ in range(7))
        as_in = pa.float64() if pa.types.is_floating(typ) else pa.int64()
        extra["mean_" + c] = _masked(mean, pa.float64(), ok)
        extra["count_" + c] = _masked(count, pa.int64(), None)
        extra["min_" + c] = _masked(mn, as_in, ok).cast(typ)
        extra["max_" + c] = _masked(mx, as_in, ok).cast(typ)
        extra["sum_" + c] = _masked(sm, as_in, ok)
        extra["stddev_" + c] = _masked(sd, pa.float64(), sd_ok)
        extra["zscore_" + c] = _masked(zs, pa.float64(), z_ok)
    if identity:
        out = table
    else:
        names, arrays = permute_table(table, perm, key_of(partition_cols, key_d, decode_keys))
        out = pa.table(arrays, names=names)
    zs = [k for k in extra if k.startswith("zscore_")]
    for k in [k for k in extra if not k.startswith("zscore_")] + zs:  # zscore columns last (tsdf.py:718-719)
        out = out.append_column(k, extra[k])
    return out
