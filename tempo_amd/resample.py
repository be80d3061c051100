"""resample: the host side of python/tempo/resample.py on the HIP engine.

Same module-level names, argument meaning and error behaviour as the reference module
(python/tempo/resample.py:8-23 constants, :38-117 aggregate, :120-136 checkAllowableFreq,
:139-143 validateFuncExists); the window() + groupBy plan is replaced by tmp_resample.
"""
import numpy as np
import pandas as pd

from . import columns as col
from . import ops

# define global frequency options (python/tempo/resample.py:8-11)
SEC = 'sec'
MIN = 'min'
HR = 'hr'
DAY = 'day'

# define global aggregate function options for downsampling (python/tempo/resample.py:14-18)
floor = "floor"
min = "min"
max = "max"
average = "mean"
ceiling = "ceil"

freq_dict = {'sec': 'seconds', 'min': 'minutes', 'hr': 'hours', 'day': 'days', 'hour': 'hours'}

allowableFreqs = [SEC, MIN, HR, DAY]
allowableFuncs = [floor, min, max, average, ceiling]

_UNIT_US = {'seconds': 1_000_000, 'minutes': 60_000_000, 'hours': 3_600_000_000, 'days': 86_400_000_000}


def checkAllowableFreq(tsdf, freq):
    """(periods, unit) exactly as python/tempo/resample.py:120-136 (including its None return
    for an unknown unit)."""
    if freq not in allowableFreqs:
        try:
            periods = freq.lower().split(" ")[0].strip()
            units = freq.lower().split(" ")[1].strip()
        except Exception:
            raise ValueError("Allowable grouping frequencies are sec (second), min (minute), hr (hour), day. "
                             "Reformat your frequency as <integer> <day/hour/minute/second>")
        if units.startswith(SEC):
            return (periods, SEC)
        elif units.startswith(MIN):
            return (periods, MIN)
        elif (units.startswith("hour") | units.startswith(HR)):
            return (periods, "hour")
        elif units.startswith(DAY):
            return (periods, DAY)
    elif freq in allowableFreqs:
        return (1, freq)


def validateFuncExists(func):
    if func is None:
        raise ValueError("Aggregate function missing. Provide one of the allowable functions: "
                         + ", ".join(allowableFuncs))
    elif func not in allowableFuncs:
        raise ValueError("Aggregate function is not in the valid list. Provide one of the allowable functions: "
                         + ", ".join(allowableFuncs))


def bucket_duration_us(tsdf, freq):
    """Microseconds of the tumbling window "<n> <unit>" that resample.py:33 hands to f.window."""
    parsed = checkAllowableFreq(tsdf, freq)  # None for an unknown unit -> TypeError, as upstream
    periods, unit = parsed[0], freq_dict[parsed[1]]
    return int(periods) * _UNIT_US[unit], int(periods), unit


def aggregate(tsdf, freq, func, metricCols=None, prefix=None, fill=None):
    """aggregate a data frame by a coarser timestamp than the initial TSDF ts_col
    (python/tempo/resample.py:38-117).  Returns a pandas DataFrame: partition cols, ts_col
    (= bucket start), then the aggregated columns sorted by name."""
    df = tsdf.df
    dur_us, periods, unit = bucket_duration_us(tsdf, freq)
    ts_series = df[tsdf.ts_col]
    if not pd.api.types.is_datetime64_any_dtype(ts_series.dtype):
        raise TypeError("resample needs a timestamp ts_col (Spark's window() does)")
    if metricCols is None:
        metricCols = [c for c in df.columns if c not in tsdf.partitionCols + [tsdf.ts_col]]
    prefix = '' if prefix is None else prefix + '_'

    (key,), n_keys = col.encode_keys([df], tsdf.partitionCols)
    ts, _ = col.ts_to_int64(ts_series)
    key_d, ts_d = col.to_dev(key), col.to_dev(ts)

    dev_cols, dev_valid, decoders = [], [], []
    for c in metricCols:
        s = df[c]
        kind = col.spark_dtype(s)
        if kind in ("tinyint", "smallint", "int", "bigint", "float", "double"):
            t, v, _ = col.numeric_to_device(s)
            decoders.append(("num", s.dtype))
        elif func == average:
            t = v = None  # avg of a non-numeric column is NULL (resample.py:68-69 on Spark)
            decoders.append(("null", None))
        elif kind == "timestamp":
            t, v, dec = col.any_to_device(s)
            decoders.append(("ts", dec))
        else:
            t, v, uniq = col.ordered_codes(s)
            decoders.append(("codes", uniq))
        dev_cols.append(t)
        dev_valid.append(v)

    live = [i for i, t in enumerate(dev_cols) if t is not None]
    r = ops.resample(key_d, ts_d, n_keys, dur_us, func, [dev_cols[i] for i in live],
                     [dev_valid[i] for i in live])
    nb = r["n"]
    starts = r["start"].cpu().numpy()
    out = {}
    if func in (floor, ceiling):
        rows = r["row"].cpu().numpy()
        first = df.iloc[rows].reset_index(drop=True)
        for c in tsdf.partitionCols:
            out[c] = first[c]
        for c in metricCols:
            out[prefix + c] = first[c].rename(prefix + c)
    else:
        # one representative row per bucket carries the partition column values
        rep = ops.resample(key_d, ts_d, n_keys, dur_us, floor, [], [])["row"].cpu().numpy()
        first = df.iloc[rep].reset_index(drop=True)
        for c in tsdf.partitionCols:
            out[c] = first[c]
        f64 = r["f64"].cpu().numpy()
        i64 = r["i64"].cpu().numpy()
        valid = r["valid"].cpu().numpy().astype(bool)
        pos = {orig: k for k, orig in enumerate(live)}
        for i, c in enumerate(metricCols):
            kind, info = decoders[i]
            name = prefix + c
            if kind == "null":
                out[name] = pd.Series(np.full(nb, np.nan), name=name)
                continue
            k = pos[i]
            ok = valid[k]
            if func == average:
                out[name] = pd.Series(np.where(ok, f64[k], np.nan), name=name)
            elif kind == "num":
                if r["is_i64"][k]:
                    vals = pd.Series(i64[k], name=name)
                    out[name] = vals.astype(info) if ok.all() else vals.astype("Int64").where(ok, pd.NA)
                else:
                    out[name] = pd.Series(np.where(ok, f64[k], np.nan), name=name).astype(info)
            elif kind == "ts":
                out[name] = info(i64[k], ok).rename(name)
            else:
                vals = np.asarray(info, dtype=object)[i64[k]] if len(info) else np.full(nb, None, object)
                out[name] = pd.Series(vals, name=name, dtype=object).where(ok, None)
    tz = getattr(ts_series.dtype, "tz", None)
    if tz is None:
        ts_out = pd.Series(starts.astype("datetime64[us]").astype(ts_series.to_numpy().dtype))
    else:
        ts_out = pd.Series(pd.to_datetime(starts, unit="us", utc=True).tz_convert(tz))
    non_part = sorted(k for k in out if k not in tsdf.partitionCols)  # resample.py:98-100
    pieces = [out[c].reset_index(drop=True).rename(c) for c in tsdf.partitionCols]
    pieces.append(ts_out.rename(tsdf.ts_col))
    pieces += [out[name].reset_index(drop=True).rename(name) for name in non_part]
    res = pd.concat(pieces, axis=1) if nb else pd.DataFrame(
        {p.name: p for p in pieces}, columns=[p.name for p in pieces])

    if fill and nb:
        res = _fill_grid(res, tsdf, dur_us, r["key"], r["start"], n_keys)
    return res


def _fill_grid(res, tsdf, dur_us, bucket_key, bucket_start, n_keys):
    """fill=True (python/tempo/resample.py:102-115): per key a regular grid from the first to the
    last bucket, left-joined with the aggregates, numeric NULLs replaced by 0.

    The grid is the device's: tmp_interpolate expands every bucket row of a key into the run of
    grid points up to the key's next bucket and reports, per grid row, the bucket row it stems from
    and whether it was generated.  bucket_key / bucket_start are the device columns tmp_resample
    returned (row b of `res` is bucket b)."""
    g = ops.interpolate(bucket_key, bucket_start, n_keys, dur_us, "zero", [], [])
    src = g["src"].cpu().numpy()
    generated = g["ts_interp"].cpu().numpy().astype(bool)
    ts_col = tsdf.ts_col
    out = res.iloc[src].reset_index(drop=True)
    ts_dtype = res[ts_col].dtype
    tz = getattr(ts_dtype, "tz", None)
    grid_ts = g["ts"].cpu().numpy()
    if tz is None:
        out[ts_col] = pd.Series(grid_ts.astype("datetime64[us]").astype(res[ts_col].to_numpy().dtype))
    else:
        out[ts_col] = pd.Series(pd.to_datetime(grid_ts, unit="us", utc=True).tz_convert(tz))
    for c in out.columns:
        if c in tsdf.partitionCols or c == ts_col:
            continue
        if col.spark_dtype(res[c]) in ("bigint", "double", "int", "float"):
            out[c] = out[c].where(~generated, 0).fillna(0).astype(res[c].dtype)
        else:
            out[c] = out[c].where(~generated, None)
    return out[list(res.columns)]
