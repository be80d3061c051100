"""Spark-side glue: the functions a PySpark job passes to `mapInArrow` / `applyInArrow`, and the
way a Spark DataFrame becomes the Arrow table the engine takes (python/tempo/tsdf.py:24-36 keeps
a Spark DataFrame; python/tempo/tsdf.py:121 and :571 partition it by the partition columns).

Spark hands a Python worker its partition as an iterator of `pyarrow.RecordBatch`; every series
must be whole inside one partition, so the job repartitions by the partition columns first:

    fn = spark_glue.range_stats_in_arrow("event_ts", ["symbol"], ["price"], 600)
    out = df.repartition("symbol").mapInArrow(fn, schema=<df.schema + spark_glue.range_stats_fields(arrow schema of df, ["price"])>)

    fn = spark_glue.asof_join_in_arrow("event_ts", ["symbol"], right_prefix="right")
    out = left.groupBy("symbol").cogroup(right.groupBy("symbol")).applyInArrow(fn, schema=...)

The functions only touch pyarrow objects, so they are exercised without Spark; pyspark itself is
imported nowhere but in `to_arrow`, and only when the object handed in is a Spark DataFrame.
"""
import pyarrow as pa

from . import arrow_io

STAT_NAMES = ("mean", "count", "min", "max", "sum", "stddev", "zscore")


def _table(batches):
    batches = [b for b in batches]
    if not batches:
        return None
    if isinstance(batches[0], pa.Table):
        return pa.concat_tables(batches)
    return pa.Table.from_batches(batches)


def range_stats_in_arrow(ts_col, partition_cols, cols, range_back_secs=1000, max_rows_per_batch=1 << 20):
    """f(iterator of RecordBatch) -> iterator of RecordBatch for `DataFrame.mapInArrow`: the
    partition's rows in (partition, ts) order with the withRangeStats columns appended
    (python/tempo/tsdf.py:673-721)."""
    def fn(batches):
        table = _table(batches)
        if table is None or table.num_rows == 0:
            return
        out = arrow_io.range_stats(table, ts_col, partition_cols, cols, range_back_secs)
        yield from out.to_batches(max_chunksize=max_rows_per_batch)
    return fn


def asof_join_in_arrow(ts_col, partition_cols, right_prefix="right", skip_nulls=True, ts_partition_val=None,
                       fraction=0.5):
    """f(left Table, right Table) -> Table for a cogrouped `applyInArrow`: the AS-OF join of the
    two co-partitioned sides (python/tempo/tsdf.py:463-560)."""
    def fn(left, right):
        return arrow_io.asof_join(left, right, ts_col, partition_cols, right_prefix, skip_nulls, ts_partition_val,
                                  fraction)
    return fn


def range_stats_fields(arrow_schema, cols):
    """The fields withRangeStats appends to a table of `arrow_schema`, in output order (the
    zscore columns last, python/tempo/tsdf.py:718-719)."""
    fields = []
    for c in cols:
        t = arrow_schema.field(c).type
        wide = pa.float64() if pa.types.is_floating(t) else pa.int64()
        kinds = dict(mean=pa.float64(), count=pa.int64(), min=t, max=t, sum=wide, stddev=pa.float64())
        fields += [pa.field(f"{k}_{c}", typ) for k, typ in kinds.items()]
    fields += [pa.field(f"zscore_{c}", pa.float64()) for c in cols]
    return fields


def to_arrow(df):
    """A pyarrow Table from what the caller holds: a pyarrow Table / RecordBatch, a pandas
    DataFrame, or a Spark DataFrame (collected through Arrow: `toArrow()` where Spark has it,
    the Arrow batches behind `toPandas()` otherwise)."""
    if isinstance(df, pa.Table):
        return df
    if isinstance(df, pa.RecordBatch):
        return pa.Table.from_batches([df])
    mod = type(df).__module__
    if mod.startswith("pyspark"):
        if hasattr(df, "toArrow"):
            return df.toArrow()
        return pa.Table.from_batches(df._collect_as_arrow())
    if mod.startswith("pandas"):
        return pa.Table.from_pandas(df, preserve_index=False)
    raise TypeError(f"cannot turn a {type(df).__name__} into an Arrow table")
