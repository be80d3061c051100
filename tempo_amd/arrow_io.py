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
        enc = pc.dictionary_encode(whole, null_encoding="encode")
        codes = enc.indices.to_numpy(zero_copy_only=False).astype(np.int64)
        dicts.append(enc.dictionary)
        col_codes.append(codes)
        if dense is None:
            dense = codes
        else:
            # re-densified after every column: the running code stays below the row count, so
            # (rows x cardinality) cannot wrap int64 however many columns there are
            _, dense = np.unique(dense * max(len(enc.dictionary), 1) + codes, return_inverse=True)
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
            return [dicts[0].take(pa.array(codes, type=pa.int64()))]
        rows = first_row[codes]
        return [dicts[j].take(pa.array(col_codes[j][rows], type=pa.int64())) for j in range(len(cols))]

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
    enc = pc.dictionary_encode(whole, null_encoding="encode")
    dictionary, indices = enc.dictionary, enc.indices
    n_keys = max(len(dictionary), 1)
    codes, _, _ = ops.arrow_import(indices, "i32")
    parts, at = [], 0
    for t in tables:
        parts.append(codes[at:at + t.num_rows])
        at += t.num_rows
    return parts, n_keys, (lambda c: [dictionary.take(pa.array(np.asarray(c, dtype=np.int64), type=pa.int64()))])


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
    fused join: integers, floats, timestamps and dictionary-encodable columns (strings)."""
    t = arr.type
    if pa.types.is_floating(t) and t != pa.float16():
        vals, valid, _ = ops.arrow_import(arr, "f64")
        dec = lambda v, ok: pa.array(v, type=pa.float64(), mask=~ok).cast(t)
    elif pa.types.is_integer(t):
        vals, valid, _ = ops.arrow_import(arr, "i64")
        dec = lambda v, ok: pa.array(v, type=pa.int64(), mask=~ok).cast(t)
    elif pa.types.is_timestamp(t):
        vals, valid, _ = ops.arrow_import(arr, "i64_raw")
        dec = lambda v, ok: pa.array(v, type=pa.int64(), mask=~ok).cast(t)
    else:
        enc = pc.dictionary_encode(_combined(arr))
        vals, valid, _ = ops.arrow_import(enc.indices, "i64")
        dictionary = enc.dictionary
        dec = lambda v, ok: dictionary.take(pa.array(v, type=pa.int64(), mask=~ok))
    return vals, valid, dec


def _stamps(us, typ):
    return pa.array(us, type=pa.int64()).cast(pa.timestamp("us", tz=typ.tz)).cast(typ)


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
    (key_l, key_r), n_keys, _ = encode_keys_dev([left, right], partition_cols)
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
            renc.append((ts_r, None, (lambda v, ok, t=ts_type_r: pa.array(v, type=pa.int64(), mask=~ok)
                                      .cast(pa.timestamp("us", tz=t.tz)).cast(t))))
        else:
            renc.append(payload_to_device(right[c]))
    j = ops.asof_join_table(key_l, ts_l, key_r, ts_r, n_keys, [], [t for t, _, _ in renc], [v for _, v, _ in renc],
                            skip_nulls=skip_nulls, ts_partition_val=ts_partition_val, fraction=fraction,
                            left_by_reference=True)
    if j["in_place"]:
        ordered_left = left
    else:
        ordered_left = left.take(pa.array(j["perm"].cpu().numpy()))
    names = list(partition_cols) + [lname[c] for c in left_cols]
    arrays = [ordered_left[c] for c in partition_cols] + [ordered_left[c] for c in left_cols]
    for i, c in enumerate(right_cols):
        vals, ok = j["right"][i]
        names.append(rname[c])
        arrays.append(renc[i][2](vals.cpu().numpy(), ok.cpu().numpy().astype(bool)))
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
    (key_d,), n_keys, _ = encode_keys_dev([table], partition_cols)
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
        perm = perm.cpu().numpy()
    else:
        perm = ops.sort_order(key_d, ts_d, n_keys, floor_seconds=True).cpu().numpy() if table.num_rows else np.arange(0)
        outs = []
    for c, typ, r in zip(cols, types, outs):
        flags = r["valid"].cpu().numpy()
        ok, sd_ok, z_ok = (flags & 1) != 0, (flags & 2) != 0, (flags & 4) != 0
        as_in = pa.float64() if pa.types.is_floating(typ) else pa.int64()
        extra["mean_" + c] = pa.array(r["mean"].cpu().numpy(), mask=~ok)
        extra["count_" + c] = pa.array(r["count"].cpu().numpy())
        extra["min_" + c] = pa.array(r["min"].cpu().numpy(), type=as_in, mask=~ok).cast(typ)
        extra["max_" + c] = pa.array(r["max"].cpu().numpy(), type=as_in, mask=~ok).cast(typ)
        extra["sum_" + c] = pa.array(r["sum"].cpu().numpy(), type=as_in, mask=~ok)
        extra["stddev_" + c] = pa.array(r["stddev"].cpu().numpy(), mask=~sd_ok)
        extra["zscore_" + c] = pa.array(r["zscore"].cpu().numpy(), mask=~z_ok)
    identity = len(perm) > 0 and perm[0] == 0 and perm[-1] == len(perm) - 1 and bool((np.diff(perm) == 1).all())
    out = table if identity else table.take(pa.array(perm.astype(np.int64)))
    zs = [k for k in extra if k.startswith("zscore_")]
    for k in [k for k in extra if not k.startswith("zscore_")] + zs:  # zscore columns last (tsdf.py:718-719)
        out = out.append_column(k, extra[k])
    return out
