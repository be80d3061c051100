"""DataFrame columns <-> device columns.

The boundary the reference crosses implicitly (Spark rows -> Tungsten): pandas / Arrow columns
become contiguous HBM arrays that the C ABI understands -- partition keys are
dictionary-encoded to int32 ids, timestamps become int64 microseconds (Spark TimestampType),
numeric columns become float64 / int64 with a separate validity mask for NULLs, strings become
order-preserving int64 codes.  Nothing here computes results; it only re-encodes columns.
"""
import numpy as np
import pandas as pd
import torch


def device():
    if not torch.cuda.is_available():
        raise RuntimeError("tempo_amd needs an MI355X (torch.cuda.is_available() is False); "
                           "there is no CPU execution path")
    return torch.device("cuda", torch.cuda.current_device())


def to_pandas(df):
    """Accept pandas DataFrames and pyarrow Tables."""
    if isinstance(df, pd.DataFrame):
        return df
    return to_arrow(df).to_pandas()


def to_arrow(df):
    """A pyarrow Table from a pyarrow Table / RecordBatch or a Spark DataFrame (collected through
    Arrow); anything else is refused as the TSDF constructor's `df`."""
    import pyarrow as pa
    if isinstance(df, pa.Table):
        return df
    if isinstance(df, pa.RecordBatch) or type(df).__module__.startswith("pyspark"):
        from . import spark_glue
        return spark_glue.to_arrow(df)
    raise TypeError(f"df must be a pandas DataFrame, a pyarrow Table or a Spark DataFrame, got {type(df)}")


def to_dev(a):
    if a is None:
        return None
    a = np.ascontiguousarray(a)
    if not a.flags.writeable:  # Arrow buffers are read-only; torch wants a writable source
        a = a.copy()
    return torch.from_numpy(a).to(device())


def spark_dtype(series):
    """The Spark SQL type name the reference's dtype checks would see for this column
    (python/tempo/tsdf.py:697, python/tempo/resample.py:111)."""
    dt = series.dtype
    if pd.api.types.is_bool_dtype(dt):
        return "boolean"
    if pd.api.types.is_datetime64_any_dtype(dt):
        return "timestamp"
    if pd.api.types.is_integer_dtype(dt):
        size = getattr(dt, "itemsize", None) or np.dtype(dt.numpy_dtype).itemsize
        return {1: "tinyint", 2: "smallint", 4: "int", 8: "bigint"}[size]
    if pd.api.types.is_float_dtype(dt):
        size = getattr(dt, "itemsize", None) or np.dtype(dt.numpy_dtype).itemsize
        return "float" if size == 4 else "double"
    return "string"


def encode_keys(frames, cols):
    """Dictionary-encode the partition columns of several frames with ONE shared dictionary.

    Returns ([int32 codes per frame], n_keys).  No partition columns -> (None per frame, 1).
    NULL keys form their own group, as Spark's partitionBy does."""
    if not cols:
        return [None for _ in frames], 1
    lens = [len(f) for f in frames]
    combined = None
    for c in cols:
        vals = pd.concat([f[c] for f in frames], ignore_index=True)
        codes, uniq = pd.factorize(vals, use_na_sentinel=False)
        codes = codes.astype(np.int64)
        if combined is None:
            combined = codes
        else:
            # the running code is re-factorized after every column, so it stays below the row
            # count and (rows x cardinality) cannot wrap int64 however many columns there are
            combined, _ = pd.factorize(combined * max(len(uniq), 1) + codes)
            combined = combined.astype(np.int64)
    n_keys = int(combined.max()) + 1 if len(combined) else 1
    if n_keys >= 2 ** 31:
        raise NotImplementedError("more than 2^31 distinct partition keys")
    out, at = [], 0
    for n in lens:
        out.append(combined[at:at + n].astype(np.int32))
        at += n
    return out, n_keys


def ts_to_int64(series):
    """(int64 array, ts_is_seconds).  datetime64 -> microseconds since the epoch (floor), tz-aware
    values converted to UTC; integer columns pass through and are their own cast("long")."""
    dt = series.dtype
    if pd.api.types.is_datetime64_any_dtype(dt):
        if series.isna().any():
            raise ValueError("NULL timestamps are not supported in the time index column")
        if getattr(dt, "tz", None) is not None:
            series = series.dt.tz_convert("UTC").dt.tz_localize(None)
        return series.to_numpy().astype("datetime64[us]").astype(np.int64), False
    if pd.api.types.is_integer_dtype(dt):
        if series.isna().any():
            raise ValueError("NULL timestamps are not supported in the time index column")
        return series.to_numpy(dtype=np.int64), True
    raise TypeError(f"time index column must be a timestamp or an integer column, got {dt}")


def numeric_to_device(series):
    """(values tensor float64|int64, validity tensor uint8 or None, kind) for a numeric column.
    pandas float NaN is SQL NULL (pandas has no separate NULL for floats)."""
    dt = series.dtype
    if pd.api.types.is_bool_dtype(dt):
        raise TypeError("boolean columns are not numeric")
    if pd.api.types.is_integer_dtype(dt):
        mask = series.isna().to_numpy()
        vals = series.fillna(0).to_numpy(dtype=np.int64) if mask.any() else series.to_numpy(dtype=np.int64)
        return to_dev(vals), (to_dev((~mask).astype(np.uint8)) if mask.any() else None), "i64"
    if pd.api.types.is_float_dtype(dt):
        vals = series.to_numpy(dtype=np.float64, na_value=np.nan)
        mask = np.isnan(vals)
        return to_dev(vals), (to_dev((~mask).astype(np.uint8)) if mask.any() else None), "f64"
    raise TypeError(f"column of dtype {dt} is not numeric")


def any_to_device(series):
    """Encode ANY column for a device gather: returns (tensor, validity or None, decode) where
    decode(values ndarray, valid ndarray bool) rebuilds a pandas Series of the original dtype."""
    dt = series.dtype
    name = series.name
    if pd.api.types.is_datetime64_any_dtype(dt):
        mask = series.isna().to_numpy()
        tz = getattr(dt, "tz", None)
        s = series.dt.tz_convert("UTC").dt.tz_localize(None) if tz is not None else series
        raw = s.to_numpy().astype("datetime64[ns]").astype(np.int64)

        def dec(v, ok):
            out = pd.Series(v.astype("datetime64[ns]"), name=name)
            if tz is not None:
                out = out.dt.tz_localize("UTC").dt.tz_convert(tz)
            return out.where(ok, pd.NaT)
        return to_dev(raw), (to_dev((~mask).astype(np.uint8)) if mask.any() else None), dec
    if pd.api.types.is_bool_dtype(dt) and not series.isna().any():
        def dec(v, ok):
            out = pd.Series(v.astype(bool), name=name)
            return out if ok.all() else out.astype(object).where(ok, None)
        return to_dev(series.to_numpy(dtype=np.uint8)), None, dec
    if pd.api.types.is_integer_dtype(dt):
        t, valid, _ = numeric_to_device(series)

        def dec(v, ok):
            if ok.all():
                return pd.Series(v, name=name).astype(dt)
            ext = dt if isinstance(dt, pd.api.extensions.ExtensionDtype) else pd.Int64Dtype()
            return pd.Series(pd.array(v, dtype="int64"), name=name).astype(ext).where(ok, pd.NA)
        return t, valid, dec
    if pd.api.types.is_float_dtype(dt):
        t, valid, _ = numeric_to_device(series)

        def dec(v, ok):
            return pd.Series(np.where(ok, v, np.nan), name=name).astype(dt)
        return t, valid, dec
    # strings / objects: dictionary codes
    codes, uniq = pd.factorize(series, use_na_sentinel=True)
    mask = codes < 0
    codes = np.where(mask, 0, codes).astype(np.int32)

    def dec(v, ok):
        vals = np.asarray(uniq, dtype=object)[v] if len(uniq) else np.full(len(v), None, dtype=object)
        out = pd.Series(vals, name=name, dtype=object)
        return out.where(ok, None)
    return to_dev(codes), (to_dev((~mask).astype(np.uint8)) if mask.any() else None), dec


def ordered_codes(series):
    """Order-preserving int64 codes of a non-numeric column (sorted dictionary) for min / max /
    struct comparisons on the device.  Returns (codes tensor, validity or None, uniques)."""
    mask = series.isna().to_numpy()
    uniq = np.array(sorted(series[~mask].unique()), dtype=object)
    # codes by one vectorised lookup (a categorical with the sorted dictionary), NULL -> code 0
    codes = pd.Categorical(series, categories=uniq).codes.astype(np.int64)
    codes[codes < 0] = 0
    return to_dev(codes), (to_dev((~mask).astype(np.uint8)) if mask.any() else None), uniq
