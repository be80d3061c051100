"""TSDF: the reference's time-series DataFrame API on the MI355X engine.

Same class, method names, arguments, column naming and error behaviour as
python/tempo/tsdf.py for the hot path (asofJoin incl. the tsPartitionVal/fraction skew mode,
withRangeStats, EMA, withLookbackFeatures, resample); `df` is a pandas DataFrame, a pyarrow Table
or a Spark DataFrame (taken over as Arrow).  Every O(rows) computation runs in libtempo_hip.so
on the GPU; this module only re-encodes columns, names outputs and raises the reference's errors.

A TSDF built on Arrow data stays on Arrow: asofJoin and withRangeStats move the Arrow buffers to
the device as they are, assemble their result as an Arrow table and hand back a TSDF that holds
it; `.df` turns it into pandas only when somebody asks.
"""
import logging

import numpy as np
import pandas as pd
import torch

from . import columns as col
from . import io as tio
from . import ops
from . import resample as rs
from .interpol import Interpolation

logger = logging.getLogger(__name__)

_SUMMARIZABLE = ['int', 'bigint', 'float', 'double']  # python/tempo/tsdf.py:697


class TSDF:

    def __init__(self, df, ts_col="event_ts", partition_cols=None, sequence_col=None):
        """python/tempo/tsdf.py:24-36.  df: pandas DataFrame, pyarrow Table / RecordBatch, or a Spark
        DataFrame (collected through Arrow)."""
        self._pandas, self._table = None, None
        if isinstance(df, pd.DataFrame):
            self._pandas = df
            names = list(df.columns)
        else:
            self._table = col.to_arrow(df)
            names = list(self._table.column_names)
        self.ts_col = self.__validated_column(names, ts_col)
        self.partitionCols = [] if partition_cols is None else self.__validated_columns(names, partition_cols)
        self.sequence_col = '' if sequence_col is None else sequence_col

    @property
    def df(self):
        """The table as a pandas DataFrame (built from the Arrow table on first use)."""
        if self._pandas is None:
            self._pandas = self._table.to_pandas()
        return self._pandas

    @df.setter
    def df(self, value):
        self._pandas, self._table = value, None

    @property
    def table(self):
        """The table as a pyarrow Table (built from the pandas frame on first use)."""
        if self._table is None:
            import pyarrow as pa
            self._table = pa.Table.from_pandas(self._pandas, preserve_index=False)
        return self._table

    def _on_arrow(self):
        return self._table is not None

    ##
    ## Helper functions
    ##

    def __validated_column(self, names, colname):  # tsdf.py:45-50
        if type(colname) != str:
            raise TypeError(f"Column names must be of type str; found {type(colname)} instead!")
        if colname.lower() not in [c.lower() for c in names]:
            raise ValueError(f"Column {colname} not found in Dataframe")
        return colname

    def __validated_columns(self, df, colnames):  # tsdf.py:52-64
        if type(colnames) == str:
            colnames = [colnames]
        if colnames is None:
            colnames = []
        elif type(colnames) != list:
            raise TypeError(f"Columns must be of type list, str, or None; found {type(colnames)} instead!")
        for c in colnames:
            self.__validated_column(df, c)
        return colnames

    def __checkPartitionCols(self, tsdf_right):  # tsdf.py:66-69
        for left_col, right_col in zip(self.partitionCols, tsdf_right.partitionCols):
            if left_col != right_col:
                raise ValueError("left and right dataframe partition columns should have same name in same order")

    def _spark_dtype(self, name):
        if self._on_arrow():
            from . import arrow_io
            return arrow_io.spark_dtype(self._table.schema.field(name).type)
        return col.spark_dtype(self.df[name])

    def __validateTsColMatch(self, right_tsdf):  # tsdf.py:71-75
        left_ts_datatype = self._spark_dtype(self.ts_col)
        right_ts_datatype = right_tsdf._spark_dtype(right_tsdf.ts_col)
        if left_ts_datatype != right_ts_datatype:
            raise ValueError("left and right dataframe timestamp index columns should have same type")

    @staticmethod
    def __prefixed(name, prefix):  # tsdf.py:77-94: '' keeps the name, otherwise '<prefix>_<name>'
        return name if prefix == '' else prefix + '_' + name

    ##
    ## AS-OF join
    ##

    def asofJoin(self, right_tsdf, left_prefix=None, right_prefix="right", tsPartitionVal=None, fraction=0.5,
                 skipNulls=True, sql_join_opt=False, suppress_null_warning=False):
        """AS-OF join (python/tempo/tsdf.py:463-560).  Every left row is joined with the latest
        right row of the same partition key at or before it; with skipNulls each right column
        independently takes its latest non-NULL value.  tsPartitionVal / fraction select the
        time-bracketed ("skew") variant.  sql_join_opt (a Spark broadcast-join plan with
        different semantics, tsdf.py:492-509) does not apply to this engine and is ignored."""
        if tsPartitionVal is not None:
            logger.warning("You are using the skew version of the AS OF join. This may result in null values if "
                           "there are any values outside of the maximum lookback. For maximum efficiency, choose "
                           "smaller values of maximum lookback, trading off performance and potential blank AS OF "
                           "values for sparse keys")
        self.__checkPartitionCols(right_tsdf)
        self.__validateTsColMatch(right_tsdf)
        if skipNulls is False and tsPartitionVal is not None:  # tsdf.py:123-125
            raise ValueError("Disabling null skipping with a partition value is not supported yet.")
        if tsPartitionVal is not None and not isinstance(tsPartitionVal, (int, float)):
            raise TypeError("tsPartitionVal must be an int or a float")

        if (self._on_arrow() and right_tsdf._on_arrow() and not right_tsdf.sequence_col
                and self._spark_dtype(self.ts_col) == "timestamp"):
            return self.__asofJoinArrow(right_tsdf, left_prefix, right_prefix, tsPartitionVal, fraction, skipNulls,
                                        suppress_null_warning)

        left_df, right_df = self.df, right_tsdf.df
        part = self.partitionCols
        left_cols = [c for c in left_df.columns if c not in part]
        right_cols = [c for c in right_df.columns if c not in part]
        lname = {c: (self.__prefixed(c, left_prefix) if left_prefix is not None else c) for c in left_cols}
        rname = {c: self.__prefixed(c, right_prefix) for c in right_cols}
        clash = set(lname.values()) & set(rname.values())
        if clash:
            raise ValueError(f"duplicate column names after prefixing: {sorted(clash)}")

        (key_l, key_r), n_keys = col.encode_keys([left_df, right_df], part)
        ts_l, secs = col.ts_to_int64(left_df[self.ts_col])
        ts_r, _ = col.ts_to_int64(right_df[right_tsdf.ts_col])
        seq = seq_valid = None
        if right_tsdf.sequence_col:
            seq, seq_valid, kind = col.numeric_to_device(right_df[right_tsdf.sequence_col])
            if kind != "i64":
                seq = seq.to(torch.int64)

        # One fused device call delivers the joined right columns in (key, ts) order -- the order
        # Spark's window leaves them in -- together with the left row every output row stems from;
        # the left columns follow by that permutation on the host.  Every right column travels as
        # an 8-byte device column (narrower encodings are widened and narrowed back on the way out);
        # the right time index is the join's own time column.
        enc, dev_cols, dev_masks = {}, [], []
        for c in right_cols:
            if c == right_tsdf.ts_col:
                enc[c] = None
                dev_cols.append(col.to_dev(ts_r))
                dev_masks.append(None)
                continue
            values, valid, decode = col.any_to_device(right_df[c])
            enc[c] = (values.dtype, decode)
            if values.dtype == torch.float32:
                values = values.to(torch.float64)
            elif values.dtype not in (torch.float64, torch.int64):
                values = values.to(torch.int64)
            dev_cols.append(values)
            dev_masks.append(valid)
        if len({m.data_ptr() for m in dev_masks if m is not None}) > 16:
            raise NotImplementedError("asofJoin supports at most 16 right columns with NULLs per call")
        j = ops.asof_join_table(col.to_dev(key_l), col.to_dev(ts_l), col.to_dev(key_r), col.to_dev(ts_r), n_keys,
                                [], dev_cols, dev_masks, skip_nulls=bool(skipNulls), seq_r=seq, seq_r_valid=seq_valid,
                                ts_is_seconds=secs, ts_partition_val=tsPartitionVal, fraction=fraction)
        perm = j["perm"].cpu().numpy()
        left_sorted = left_df.iloc[perm].reset_index(drop=True)
        out = {c: left_sorted[c] for c in part}
        for c in left_cols:
            out[lname[c]] = left_sorted[c].rename(lname[c])
        mask_of = {}
        for k, c in enumerate(right_cols):
            values, valid = j["right"][k]
            v, ok = values.cpu().numpy(), valid.cpu().numpy().astype(bool)
            if enc[c] is None:  # the right time index: microseconds (or the integral column itself)
                ts_dtype = right_df[c].dtype
                if secs:
                    ser = pd.Series(pd.array(np.where(ok, v, 0), dtype="int64"), name=rname[c])
                    ser = ser.astype(ts_dtype) if ok.all() else ser.astype(pd.Int64Dtype()).where(ok, pd.NA)
                else:
                    ser = pd.Series(v.astype("datetime64[us]").astype("datetime64[ns]"), name=rname[c])
                    tz = getattr(ts_dtype, "tz", None)
                    if tz is not None:
                        ser = ser.dt.tz_localize("UTC").dt.tz_convert(tz)
                    ser = ser.where(ok, pd.NaT)
                out[rname[c]] = ser
            else:
                dtype, decode = enc[c]
                if dtype == torch.float32:
                    v = v.astype(np.float32)
                elif dtype not in (torch.float64, torch.int64):
                    v = v.astype(torch.empty(0, dtype=dtype).numpy().dtype)
                out[rname[c]] = decode(v, ok).rename(rname[c])
            if j["mask_of"][k] >= 0:
                mask_of[c] = j["mask_of"][k]
        joined = pd.concat([out[k].rename(k) for k in out], axis=1)
        missing = j["missing"]

        if tsPartitionVal is not None and (not suppress_null_warning or logger.isEnabledFor(logging.WARNING)):
            miss = missing.cpu().numpy()
            for c in right_cols:  # tsdf.py:150-158
                if miss[1 + mask_of[c]] if c in mask_of else miss[0]:
                    logger.warning("Column " + rname[c] + " had no values within the lookback window. Consider "
                                   "using a larger window to avoid missing values. If this is the first record "
                                   "in the data frame, this warning can be ignored.")
        return TSDF(joined, lname[self.ts_col], part)

    def __asofJoinArrow(self, right_tsdf, left_prefix, right_prefix, tsPartitionVal, fraction, skipNulls,
                        suppress_null_warning):
        """asofJoin of two Arrow-backed tables (timestamp time index, no sequence column): Arrow buffers
        to the device, the joined table assembled as Arrow; same names and warnings as the pandas way."""
        from . import arrow_io
        joined, info = arrow_io.asof_join(self._table, right_tsdf._table, self.ts_col, self.partitionCols,
                                          right_prefix=right_prefix, skip_nulls=bool(skipNulls),
                                          ts_partition_val=tsPartitionVal, fraction=fraction, left_prefix=left_prefix,
                                          right_ts_col=right_tsdf.ts_col, with_info=True)
        if tsPartitionVal is not None and (not suppress_null_warning or logger.isEnabledFor(logging.WARNING)):
            miss = info["missing"]
            for c, name in info["names"].items():  # tsdf.py:150-158
                if miss[1 + info["mask_of"][c]] if c in info["mask_of"] else miss[0]:
                    logger.warning("Column " + name + " had no values within the lookback window. Consider "
                                   "using a larger window to avoid missing values. If this is the first record "
                                   "in the data frame, this warning can be ignored.")
        ts_name = self.ts_col if left_prefix in (None, '') else left_prefix + '_' + self.ts_col
        return TSDF(joined, ts_name, self.partitionCols)

    ##
    ## window ordering shared by withRangeStats / EMA / withLookbackFeatures (tsdf.py:563-580)
    ##

    def _order_inputs(self, with_seq):
        (key,), n_keys = col.encode_keys([self.df], self.partitionCols)
        ts, secs = col.ts_to_int64(self.df[self.ts_col])
        seq = seq_valid = None
        if with_seq and self.sequence_col:
            seq, seq_valid, kind = col.numeric_to_device(self.df[self.sequence_col])
            if kind != "i64":  # seq.cast("long") truncates
                seq = seq.to(torch.int64)
        return col.to_dev(key), col.to_dev(ts), secs, n_keys, seq, seq_valid

    def withPartitionCols(self, partitionCols):  # tsdf.py:583-590
        return TSDF(self.df, self.ts_col, partitionCols)

    def select(self, *cols):  # tsdf.py:319-343
        seq_col_stub = [] if bool(self.sequence_col) == False else [self.sequence_col]
        mandatory_cols = [self.ts_col] + self.partitionCols + seq_col_stub
        if set(mandatory_cols).issubset(set(cols)):
            return TSDF(self.df[list(cols)], self.ts_col, self.partitionCols, self.sequence_col or None)
        raise Exception("In TSDF's select statement original ts_col, partitionCols and seq_col_stub(optional) "
                        "must be present")

    def EMA(self, colName, window=30, exp_factor=0.2):
        """Truncated exponential moving average (python/tempo/tsdf.py:615-635):
        EMA_<colName> = sum_{i < window} exp_factor * (1 - exp_factor)**i * lag(colName, i), NULL or
        missing lags contributing 0.  Rows come back in window order."""
        emaColName = "_".join(["EMA", colName])
        key, ts, secs, n_keys, seq, seq_valid = self._order_inputs(True)
        x, x_valid, kind = col.numeric_to_device(self.df[colName])
        if kind == "i64":
            x = x.to(torch.float64)
        perm, ema = ops.ema(key, ts, x, x_valid, n_keys, window=window, exp_factor=exp_factor,
                            ts_is_seconds=secs, seq=seq, seq_valid=seq_valid)
        df = self.df.iloc[perm.cpu().numpy()].reset_index(drop=True)
        df[emaColName] = ema.cpu().numpy()
        return TSDF(df, self.ts_col, self.partitionCols)

    def withLookbackFeatures(self, featureCols, lookbackWindowSize, exactSize=True, featureColName="features"):
        """2-D trailing feature tensor per row (python/tempo/tsdf.py:637-671).  Each cell of
        `featureColName` is an array of shape [n_lookback, len(featureCols)], oldest row first.
        exactSize=True keeps only rows with a full window and returns a DataFrame (as upstream)."""
        key, ts, secs, n_keys, seq, seq_valid = self._order_inputs(True)
        xs, xv = [], []
        for c in featureCols:
            x, v, kind = col.numeric_to_device(self.df[c])
            xs.append(x.to(torch.float64) if kind == "i64" else x)
            xv.append(v)
        rows, lens, feat, fv = ops.lookback(key, ts, xs, xv, n_keys, lookbackWindowSize, exact_size=exactSize,
                                            ts_is_seconds=secs, seq=seq, seq_valid=seq_valid)
        rows = rows.cpu().numpy()
        feat = feat.cpu().numpy()
        if fv is not None:
            feat = np.where(fv.cpu().numpy().astype(bool), feat, np.nan)
        df = self.df.iloc[rows].reset_index(drop=True)
        # one object cell per row without a Python loop: a list of the leading axis' views
        cells = np.empty(len(rows), dtype=object)
        if exactSize:
            cells[:] = list(feat)
        else:
            ln = lens.cpu().numpy()
            cells[:] = [f[:k] for f, k in zip(feat, ln)]
        df[featureColName] = pd.Series(cells, dtype=object, index=df.index)
        if exactSize:
            return df
        return TSDF(df, self.ts_col, self.partitionCols)

    def withRangeStats(self, type='range', colsToSummarize=[], rangeBackWindowSecs=1000):
        """Rolling mean/count/min/max/sum/stddev/zscore over the trailing rangeBackWindowSecs
        seconds per partition key (python/tempo/tsdf.py:673-721).  Rows come back in window
        order."""
        if self.sequence_col:
            # Spark rejects a RANGE frame with two ORDER BY keys (tsdf.py:684 notes the limitation)
            raise ValueError("withRangeStats does not support a sequence column: a range window needs "
                             "exactly one ordering expression")
        if not colsToSummarize:
            prohibited_cols = [self.ts_col.lower()] + [pc.lower() for pc in self.partitionCols]
            names = self._table.column_names if self._on_arrow() else self.df.columns
            colsToSummarize = [c for c in names
                               if self._spark_dtype(c) in _SUMMARIZABLE and c.lower() not in prohibited_cols]
        if self._on_arrow() and self._spark_dtype(self.ts_col) == "timestamp":
            from . import arrow_io
            out = arrow_io.range_stats(self._table, self.ts_col, self.partitionCols, list(colsToSummarize),
                                       rangeBackWindowSecs)
            return TSDF(out, self.ts_col, self.partitionCols)
        key, ts, secs, n_keys, _, _ = self._order_inputs(False)
        # one device call: the rows are ordered once, every metric column travels with them, and the
        # frames' statistics are written column by column from that order
        xs, xv, kinds = [], [], []
        for metric in colsToSummarize:
            x, x_valid, kind = col.numeric_to_device(self.df[metric])
            xs.append(x)
            xv.append(x_valid)
            kinds.append(kind)
        if xs and len(self.df):
            perm, outs = ops.range_stats_cols(key, ts, xs, xv, n_keys, rangeBackWindowSecs, ts_is_seconds=secs)
        elif len(self.df):
            perm, outs = ops.sort_order(key, ts, n_keys, floor_seconds=not secs), []
        else:
            perm = torch.empty(0, dtype=torch.int32)
            outs = [{k: torch.empty(0, dtype=torch.int64 if (k == "count" or (kind == "i64" and k in ("min", "max", "sum")))
                                    else torch.uint8 if k == "valid" else torch.float64)
                     for k in ("mean", "count", "min", "max", "sum", "stddev", "zscore", "valid")} for kind in kinds]
        df = self.df.iloc[perm.cpu().numpy()].reset_index(drop=True)
        derived = {}
        for metric, kind, out in zip(colsToSummarize, kinds, outs):
            s = df[metric]
            valid = out["valid"].cpu().numpy()
            ok, sd_ok, z_ok = (valid & 1) != 0, (valid & 2) != 0, (valid & 4) != 0
            mean = np.where(ok, out["mean"].cpu().numpy(), np.nan)
            mn, mx, sm = out["min"].cpu().numpy(), out["max"].cpu().numpy(), out["sum"].cpu().numpy()
            df['mean_' + metric] = mean
            df['count_' + metric] = out["count"].cpu().numpy()
            if kind == "i64":
                as_in = (lambda v: pd.Series(v).astype(s.dtype) if ok.all()
                         else pd.Series(v).astype("Int64").where(ok, pd.NA))
                df['min_' + metric] = as_in(mn).values
                df['max_' + metric] = as_in(mx).values
                df['sum_' + metric] = (pd.Series(sm) if ok.all() else pd.Series(sm).astype("Int64").where(ok, pd.NA)).values
            else:
                df['min_' + metric] = np.where(ok, mn, np.nan).astype(s.dtype)
                df['max_' + metric] = np.where(ok, mx, np.nan).astype(s.dtype)
                df['sum_' + metric] = np.where(ok, sm, np.nan)
            df['stddev_' + metric] = np.where(sd_ok, out["stddev"].cpu().numpy(), np.nan)
            derived["zscore_" + metric] = np.where(z_ok, out["zscore"].cpu().numpy(), np.nan)
        for name, vals in derived.items():  # tsdf.py:718-719: zscore columns come last
            df[name] = vals
        return TSDF(df, self.ts_col, self.partitionCols)

    def write(self, spark, tabName, optimizationCols=None):  # python/tempo/tsdf.py:761-762
        tio.write(self, spark, tabName, optimizationCols)

    def withGroupedStats(self, metricCols=[], freq=None):
        """mean/count/min/max/sum/stddev of every metric column per partition key and tumbling
        `freq` bucket (python/tempo/tsdf.py:723-759).  Columns come back as the reference orders
        them: partition columns, the six statistics per metric, then ts_col (= bucket start)."""
        if not metricCols:
            prohibited_cols = [self.ts_col.lower()] + [pc.lower() for pc in self.partitionCols]
            metricCols = [c for c in self.df.columns
                          if col.spark_dtype(self.df[c]) in _SUMMARIZABLE and c.lower() not in prohibited_cols]
        dur_us, _, _ = rs.bucket_duration_us(self, freq)
        ts_series = self.df[self.ts_col]
        if not pd.api.types.is_datetime64_any_dtype(ts_series.dtype):
            raise TypeError("withGroupedStats needs a timestamp ts_col (Spark's window() does)")
        (key,), n_keys = col.encode_keys([self.df], self.partitionCols)
        ts, _ = col.ts_to_int64(ts_series)
        key_d, ts_d = col.to_dev(key), col.to_dev(ts)
        xs, xv, kinds = [], [], []
        for metric in metricCols:
            x, v, kind = col.numeric_to_device(self.df[metric])
            xs.append(x)
            xv.append(v)
            kinds.append(kind)
        r = ops.grouped_stats(key_d, ts_d, n_keys, dur_us, xs, xv)
        nb = r["n"]
        # one representative row per bucket carries the partition column values
        rep = ops.resample(key_d, ts_d, n_keys, dur_us, rs.floor, [], [])["row"].cpu().numpy()
        first = self.df.iloc[rep].reset_index(drop=True)
        out = {c: first[c] for c in self.partitionCols}
        for i, metric in enumerate(metricCols):
            s = self.df[metric]
            cnt = r["count"][i].cpu().numpy()
            ok, sd_ok = cnt > 0, cnt > 1
            mn, mx, sm = (r[k][i].cpu().numpy() for k in ("min", "max", "sum"))
            out['mean_' + metric] = pd.Series(np.where(ok, r["mean"][i].cpu().numpy(), np.nan))
            out['count_' + metric] = pd.Series(cnt)
            if kinds[i] == "i64":
                as_in = (lambda v: pd.Series(v).astype(s.dtype) if ok.all()
                         else pd.Series(v).astype("Int64").where(ok, pd.NA))
                out['min_' + metric] = as_in(mn)
                out['max_' + metric] = as_in(mx)
                out['sum_' + metric] = pd.Series(sm) if ok.all() else pd.Series(sm).astype("Int64").where(ok, pd.NA)
            else:
                out['min_' + metric] = pd.Series(np.where(ok, mn, np.nan).astype(s.dtype))
                out['max_' + metric] = pd.Series(np.where(ok, mx, np.nan).astype(s.dtype))
                out['sum_' + metric] = pd.Series(np.where(ok, sm, np.nan))
            out['stddev_' + metric] = pd.Series(np.where(sd_ok, r["stddev"][i].cpu().numpy(), np.nan))
        starts = r["start"].cpu().numpy()
        tz = getattr(ts_series.dtype, "tz", None)
        if tz is None:
            out[self.ts_col] = pd.Series(starts.astype("datetime64[us]").astype(ts_series.to_numpy().dtype))
        else:
            out[self.ts_col] = pd.Series(pd.to_datetime(starts, unit="us", utc=True).tz_convert(tz))
        names = list(out)
        summary = (pd.concat([out[k].reset_index(drop=True).rename(k) for k in names], axis=1) if nb
                   else pd.DataFrame({k: out[k] for k in names}, columns=names))
        return TSDF(summary, self.ts_col, self.partitionCols)

    def resample(self, freq, func=None, metricCols=None, prefix=None, fill=None):
        """Downsample to `freq` buckets with floor / min / max / mean / ceil
        (python/tempo/tsdf.py:764-776)."""
        rs.validateFuncExists(func)
        enriched_df = rs.aggregate(self, freq, func, metricCols, prefix, fill)
        return _ResampledTSDF(enriched_df, ts_col=self.ts_col, partition_cols=self.partitionCols, freq=freq,
                              func=func)

    def interpolate(self, freq, func, method, target_cols=None, ts_col=None, partition_cols=None,
                    show_interpolated=False):
        """Resample to `freq` with `func`, generate the missing timestamps and fill them by
        `method` (zero / null / bfill / ffill / linear) -- python/tempo/tsdf.py:778-811."""
        if ts_col is None:
            ts_col = self.ts_col
        if partition_cols is None:
            partition_cols = self.partitionCols
        if target_cols is None:
            prohibited_cols = partition_cols + [ts_col]
            target_cols = [c for c in self.df.columns
                           if col.spark_dtype(self.df[c]) in _SUMMARIZABLE and c.lower() not in prohibited_cols]
        interpolate_service = Interpolation(is_resampled=False)
        tsdf_input = TSDF(self.df, ts_col=ts_col, partition_cols=partition_cols)
        interpolated_df = interpolate_service.interpolate(tsdf_input, ts_col, partition_cols, target_cols, freq,
                                                          func, method, show_interpolated)
        return TSDF(interpolated_df, ts_col=ts_col, partition_cols=partition_cols)

    def calc_bars(tsdf, freq, func=None, metricCols=None, fill=None):
        """OHLC bars = floor / min / max / ceil per bucket joined on (key, ts)
        (python/tempo/tsdf.py:813-826)."""
        resample_open = tsdf.resample(freq=freq, func='floor', metricCols=metricCols, prefix='open', fill=fill)
        resample_low = tsdf.resample(freq=freq, func='min', metricCols=metricCols, prefix='low', fill=fill)
        resample_high = tsdf.resample(freq=freq, func='max', metricCols=metricCols, prefix='high', fill=fill)
        resample_close = tsdf.resample(freq=freq, func='ceil', metricCols=metricCols, prefix='close', fill=fill)
        join_cols = resample_open.partitionCols + [resample_open.ts_col]
        bars = (resample_open.df.merge(resample_high.df, on=join_cols).merge(resample_low.df, on=join_cols)
                .merge(resample_close.df, on=join_cols))
        non_part_cols = set(bars.columns) - set(resample_open.partitionCols) - set([resample_open.ts_col])
        sel_and_sort = resample_open.partitionCols + [resample_open.ts_col] + sorted(non_part_cols)
        return TSDF(bars[sel_and_sort], resample_open.ts_col, resample_open.partitionCols)


class _ResampledTSDF(TSDF):  # python/tempo/tsdf.py:905-909
    def __init__(self, df, ts_col="event_ts", partition_cols=None, sequence_col=None, freq=None, func=None):
        super(_ResampledTSDF, self).__init__(df, ts_col, partition_cols, sequence_col)
        self.__freq = freq
        self.__func = func

    def interpolate(self, method, target_cols=None, show_interpolated=False):
        """Fill an already sampled table (python/tempo/tsdf.py:911-944)."""
        if target_cols is None:
            prohibited_cols = self.partitionCols + [self.ts_col]
            target_cols = [c for c in self.df.columns
                           if col.spark_dtype(self.df[c]) in _SUMMARIZABLE and c.lower() not in prohibited_cols]
        interpolate_service = Interpolation(is_resampled=True)
        tsdf_input = TSDF(self.df, ts_col=self.ts_col, partition_cols=self.partitionCols)
        interpolated_df = interpolate_service.interpolate(
            tsdf=tsdf_input, ts_col=self.ts_col, partition_cols=self.partitionCols, target_cols=target_cols,
            freq=self.__freq, func=self.__func, method=method, show_interpolated=show_interpolated)
        return TSDF(interpolated_df, ts_col=self.ts_col, partition_cols=self.partitionCols)
