"""Table writer: `write(tsdf, spark, tabName, optimizationCols=None)` of python/tempo/io.py:10-43.

The reference derives `event_dt` and `event_time`, rotates the column list by one, saves a Delta
table partitioned by `event_dt` and then -- on Databricks only -- Z-orders the files by
(partitionCols, optimizationCols, event_time); elsewhere it logs a warning and leaves the files
unclustered.  Here the derived columns and the row order come from the device (`tmp_write_layout`:
rows sorted by (event_dt, partition key, ts)), so every date directory holds one file whose rows
are clustered by key and time on any platform.  The result is an ordinary Delta table directory:
Parquet files under `event_dt=YYYY-MM-DD/` plus a `_delta_log` with one commit.

There is no SparkSession in this engine.  The `spark` argument keeps its position and may be a
warehouse directory (str / os.PathLike), None (Spark's default `spark-warehouse` under the
current directory) or any object whose `conf.get("spark.sql.warehouse.dir")` names one.
"""
import json
import logging
import os
import shutil
import time
import uuid
from collections import deque

import numpy as np
import pandas as pd

from . import columns as col
from . import ops

logger = logging.getLogger(__name__)

_DELTA_TYPES = {"tinyint": "byte", "smallint": "short", "int": "integer", "bigint": "long",
                "float": "float", "double": "double", "boolean": "boolean", "timestamp": "timestamp",
                "string": "string"}


def warehouse_dir(spark):
    if spark is None:
        return os.path.join(os.getcwd(), "spark-warehouse")
    if isinstance(spark, (str, os.PathLike)):
        return os.fspath(spark)
    conf = getattr(spark, "conf", None)
    if conf is not None and hasattr(conf, "get"):
        path = conf.get("spark.sql.warehouse.dir")
        return path[5:] if path.startswith("file:") else path
    raise TypeError("spark must be a warehouse directory, None, or an object with conf.get()")


def table_path(spark, tabName):
    """Directory of table `tabName` inside the warehouse.  The name must stay inside it: empty
    names, absolute paths and names that climb out through '..' are rejected."""
    if not isinstance(tabName, str) or not tabName.strip():
        raise ValueError("tabName must be a non-empty table name")
    root = os.path.realpath(warehouse_dir(spark))
    path = os.path.realpath(os.path.join(root, tabName))
    if os.path.isabs(tabName) or path == root or os.path.commonpath([root, path]) != root:
        raise ValueError(f"tabName {tabName!r} does not name a table inside the warehouse {root!r}")
    return path


def _schema_string(df, partition_col):
    fields = []
    for c in df.columns:
        typ = "date" if c == partition_col else _DELTA_TYPES[col.spark_dtype(df[c])]
        fields.append({"name": c, "type": typ, "nullable": True, "metadata": {}})
    return json.dumps({"type": "struct", "fields": fields}, separators=(",", ":"))


def write(tsdf, spark, tabName, optimizationCols=None):
    """
    param: tsdf: input TSDF object to write
    param: spark: warehouse directory (see the module docstring)
    param: tabName Delta output table name
    param: optimizationCols list of columns to optimize on (time)
    """
    import pyarrow as pa
    import pyarrow.parquet as pq

    df = tsdf.df
    ts_col = tsdf.ts_col
    partitionCols = tsdf.partitionCols
    if optimizationCols:
        missing = [c for c in optimizationCols if c not in df.columns]
        if missing:
            raise ValueError(f"optimizationCols not in the table: {missing}")
        optimizationCols = optimizationCols + ['event_time']
    else:
        optimizationCols = ['event_time']
    if not pd.api.types.is_datetime64_any_dtype(df[ts_col].dtype):
        raise TypeError("write needs a timestamp ts_col: event_dt / event_time are read off its date and "
                        "time-of-day (io.py:29-30)")

    (key,), n_keys = col.encode_keys([df], partitionCols)
    ts, _ = col.ts_to_int64(df[ts_col])
    event_dt, event_time, perm, _ = ops.write_layout(col.to_dev(key), col.to_dev(ts), n_keys)
    order = perm.cpu().numpy()
    view_df = df.iloc[order].reset_index(drop=True)
    days = event_dt.cpu().numpy()[order]
    view_df["event_dt"] = days.astype("datetime64[D]")
    view_df["event_time"] = event_time.cpu().numpy()[order]
    view_cols = deque(view_df.columns)
    view_cols.rotate(1)  # io.py:31-33: event_time moves to the front
    view_df = view_df[list(view_cols)]

    final_path = table_path(spark, tabName)
    if os.path.exists(final_path) and not os.path.isdir(os.path.join(final_path, "_delta_log")):
        # mode("overwrite") replaces a Delta table, never an arbitrary directory or file
        raise ValueError(f"{final_path!r} exists and is not a Delta table (no _delta_log); not overwriting it")
    # the new table is built in a sibling directory and swapped in after its commit file is
    # written, so a failure on the way leaves the previous table as it was
    os.makedirs(os.path.dirname(final_path), exist_ok=True)
    path = f"{final_path}.__writing-{uuid.uuid4().hex}"
    os.makedirs(os.path.join(path, "_delta_log"))
    try:
        _write_files(path, view_df, days, partitionCols, optimizationCols)
    except BaseException:
        shutil.rmtree(path, ignore_errors=True)
        raise
    old = None
    if os.path.isdir(final_path):
        old = f"{final_path}.__old-{uuid.uuid4().hex}"
        os.replace(final_path, old)
    os.replace(path, final_path)
    if old:
        shutil.rmtree(old, ignore_errors=True)
    logger.info("wrote %d rows to %s, clustered by %s", len(view_df), final_path, partitionCols + [ts_col])


def _write_files(path, view_df, days, partitionCols, optimizationCols):
    """Parquet files under event_dt=.../ and the single commit of _delta_log, into `path`."""
    import pyarrow as pa
    import pyarrow.parquet as pq

    now_ms = int(time.time() * 1000)
    actions = [
        {"commitInfo": {"timestamp": now_ms, "operation": "CREATE OR REPLACE TABLE AS SELECT",
                        "operationParameters": {"partitionBy": "[\"event_dt\"]",
                                                "clusterBy": json.dumps(partitionCols + optimizationCols)},
                        "engineInfo": "tempo_amd"}},
        {"protocol": {"minReaderVersion": 1, "minWriterVersion": 2}},
        {"metaData": {"id": str(uuid.uuid4()), "format": {"provider": "parquet", "options": {}},
                      "schemaString": _schema_string(view_df, "event_dt"),
                      "partitionColumns": ["event_dt"], "configuration": {}, "createdTime": now_ms}},
    ]
    data_cols = [c for c in view_df.columns if c != "event_dt"]
    # rows are ordered by event_dt first: every date is one contiguous run
    starts = np.flatnonzero(np.r_[True, days[1:] != days[:-1]]) if len(days) else np.arange(0)
    ends = np.r_[starts[1:], len(days)]
    for lo, hi in zip(starts, ends):
        day = str(np.datetime64(int(days[lo]), "D"))
        rel_dir = f"event_dt={day}"
        os.makedirs(os.path.join(path, rel_dir), exist_ok=True)
        rel = f"{rel_dir}/part-00000-{uuid.uuid4()}.c000.snappy.parquet"
        table = pa.Table.from_pandas(view_df.iloc[lo:hi][data_cols], preserve_index=False)
        pq.write_table(table, os.path.join(path, rel), compression="snappy", coerce_timestamps="us",
                       allow_truncated_timestamps=True)
        actions.append({"add": {"path": rel, "partitionValues": {"event_dt": day},
                                "size": os.path.getsize(os.path.join(path, rel)),
                                "modificationTime": now_ms, "dataChange": True}})
    with open(os.path.join(path, "_delta_log", "%020d.json" % 0), "w") as f:
        for a in actions:
            f.write(json.dumps(a, separators=(",", ":")) + "\n")


def read_table(spark, tabName):
    """The table `write` produced, as a pandas DataFrame (what `spark.table(tabName)` is to the
    reference's test, python/tests/tsdf_tests.py:784-787): the files the last commit of
    `_delta_log` lists, in log order, with `event_dt` restored from the partition values."""
    import pyarrow.parquet as pq

    path = table_path(spark, tabName)
    log_dir = os.path.join(path, "_delta_log")
    commits = sorted(f for f in os.listdir(log_dir) if f.endswith(".json"))
    files, columns = {}, None
    for name in commits:
        with open(os.path.join(log_dir, name)) as f:
            for line in f:
                action = json.loads(line)
                if "add" in action:
                    files[action["add"]["path"]] = action["add"]["partitionValues"]
                elif "remove" in action:
                    files.pop(action["remove"]["path"], None)
                elif "metaData" in action:
                    columns = [fld["name"] for fld in json.loads(action["metaData"]["schemaString"])["fields"]]
    parts = []
    for rel, values in files.items():
        part = pq.read_table(os.path.join(path, rel)).to_pandas()
        for name, value in values.items():
            part[name] = np.datetime64(value, "D")
        parts.append(part)
    if not parts:
        return pd.DataFrame(columns=columns)
    return pd.concat(parts, ignore_index=True)[columns]
