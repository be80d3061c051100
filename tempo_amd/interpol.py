"""Interpolation: the host side of python/tempo/interpol.py on the HIP engine.

Same module-level names, class, method signature, validation order and error texts as the
reference module (python/tempo/interpol.py:8-64, :260-378); the lead / last / first windows, the
explode(sequence()) and the per-method CASE expressions are replaced by tmp_interpolate.
"""
from typing import List

import numpy as np
import pandas as pd
import torch

from . import columns as col
from . import ops
from . import resample as rs

# Interpolation fill options (python/tempo/interpol.py:9-10)
method_options = ["zero", "null", "bfill", "ffill", "linear"]
supported_target_col_types = ["int", "bigint", "float", "double"]


class Interpolation:
    def __init__(self, is_resampled: bool):
        self.is_resampled = is_resampled

    def __validate_fill(self, method: str):  # interpol.py:17-26
        if method not in method_options:
            raise ValueError(f"Please select from one of the following fill options: {method_options}")

    def __validate_col(self, df, partition_cols: List[str], target_cols: List[str], ts_col: str):
        """interpol.py:28-64, including its substring test against the printed column list."""
        printed = str(list(df.columns))
        for column in partition_cols:
            if column not in printed:
                raise ValueError(f"Partition Column: '{column}' does not exist in DataFrame.")
        for column in target_cols:
            if column not in printed:
                raise ValueError(f"Target Column: '{column}' does not exist in DataFrame.")
            if col.spark_dtype(df[column]) not in supported_target_col_types:
                raise ValueError(
                    f"Target Column needs to be one of the following types: {supported_target_col_types}")
        if ts_col not in printed:
            raise ValueError(f"Timestamp Column: '{ts_col}' does not exist in DataFrame.")
        if col.spark_dtype(df[ts_col]) != "timestamp":
            raise ValueError("Timestamp Column needs to be of timestamp type.")

    def interpolate(self, tsdf, ts_col: str, partition_cols: List[str], target_cols: List[str], freq: str,
                    func: str, method: str, show_interpolated: bool):
        """Apply interpolation (python/tempo/interpol.py:260-378).  Returns a pandas DataFrame:
        partition columns, ts_col, the target columns, then (show_interpolated) is_ts_interpolated
        and one is_interpolated_<col> per target; rows ordered by (partition, ts)."""
        self.__validate_fill(method)
        self.__validate_col(tsdf.df, partition_cols, target_cols, ts_col)

        input_cols = [*partition_cols, ts_col, *target_cols]
        sampled_input = tsdf.df[input_cols]
        if self.is_resampled is False:
            # Resample and normalize input (interpol.py:292-296)
            sampled_input = tsdf.resample(freq=freq, func=func, metricCols=target_cols).df
        targets = [c for c in sampled_input.columns if c in target_cols]
        freq_us, _, _ = rs.bucket_duration_us(tsdf, freq)

        (key,), n_keys = col.encode_keys([sampled_input], partition_cols)
        ts_series = sampled_input[ts_col]
        ts, _ = col.ts_to_int64(ts_series)
        xs, xv, dtypes = [], [], []
        for c in targets:
            x, v, kind = col.numeric_to_device(sampled_input[c])
            # integer targets are carried as doubles (exact below 2**53) and cast back below
            xs.append(x.to(torch.float64) if kind == "i64" else x)
            xv.append(v)
            dtypes.append((kind, sampled_input[c].dtype))
        r = ops.interpolate(col.to_dev(key), col.to_dev(ts), n_keys, freq_us, method, xs, xv)

        m = r["n"]
        src = r["src"].cpu().numpy()
        out = {c: sampled_input[c].iloc[src].reset_index(drop=True) for c in partition_cols}
        stamps = r["ts"].cpu().numpy()
        tz = getattr(ts_series.dtype, "tz", None)
        if tz is None:
            out[ts_col] = pd.Series(stamps.astype("datetime64[us]").astype(ts_series.to_numpy().dtype))
        else:
            out[ts_col] = pd.Series(pd.to_datetime(stamps, unit="us", utc=True).tz_convert(tz))
        for i, c in enumerate(targets):
            ok = r["valid"][i].cpu().numpy().astype(bool)
            vals = r["val"][i].cpu().numpy()
            kind, dt = dtypes[i]
            if kind == "i64" and method != "linear":  # a CASE over integer branches stays integer
                s = pd.Series(vals.astype(np.int64))
                out[c] = s.astype(dt) if ok.all() else s.astype("Int64").where(ok, pd.NA)
            else:
                out[c] = pd.Series(np.where(ok, vals, np.nan))
        if show_interpolated:
            out["is_ts_interpolated"] = pd.Series(r["ts_interp"].cpu().numpy().astype(bool))
            for i, c in enumerate(targets):
                out[f"is_interpolated_{c}"] = pd.Series(r["interp"][i].cpu().numpy().astype(bool))
        names = list(out)
        if m == 0:
            return pd.DataFrame({k: out[k] for k in names}, columns=names)
        return pd.concat([out[k].reset_index(drop=True).rename(k) for k in names], axis=1)
