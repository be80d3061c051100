"""Host logic of tempo_amd.TSDF (column encoding, prefixes, output naming/dtypes, reference
errors) on CPU: the device operators are swapped for the oracle by monkeypatching, so the same
golden cases run without a GPU.  The GPU suite runs them on the real kernels."""
import pytest
import torch

import oracle_ops
import tsdf_golden_cases as cases


@pytest.fixture()
def host_tsdf(monkeypatch, orc):
    import tempo_amd
    from tempo_amd import columns, ops
    monkeypatch.setattr(columns, "device", lambda: torch.device("cpu"))
    for name in ("sort_order", "asof_join", "asof_join_table", "gather", "range_stats", "range_stats_cols", "ema", "lookback", "resample", "grouped_stats", "interpolate",
                 "write_layout"):
        monkeypatch.setattr(ops, name, getattr(oracle_ops, name))
    return tempo_amd.TSDF


@pytest.mark.parametrize("case", cases.ALL_CASES, ids=lambda c: c.__name__)
def test_reference_case_host_logic(host_tsdf, case):
    case(host_tsdf)


def test_ops_refuse_cpu_tensors():
    """The product path has no CPU fallback: device operators reject host tensors outright."""
    from tempo_amd import ops
    ts = torch.zeros(4, dtype=torch.int64)
    with pytest.raises(TypeError, match="no CPU path"):
        ops.sort_order(None, ts, 1)
    with pytest.raises(TypeError, match="no CPU path"):
        ops.asof_join(None, ts, None, ts, 1)


def test_device_required_without_gpu():
    from tempo_amd import columns
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="no CPU execution path"):
        columns.device()


def test_freq_parsing_matches_reference_rules():
    from tempo_amd import resample as rs
    assert rs.checkAllowableFreq(None, "min") == (1, "min")
    assert rs.checkAllowableFreq(None, "5 minutes") == ("5", "min")
    assert rs.checkAllowableFreq(None, "2 HOURS") == ("2", "hour")
    assert rs.checkAllowableFreq(None, "3 hr") == ("3", "hour")
    assert rs.checkAllowableFreq(None, "30 seconds") == ("30", "sec")
    assert rs.checkAllowableFreq(None, "1 day") == ("1", "day")
    assert rs.checkAllowableFreq(None, "1 fortnight") is None  # upstream falls through, resample.py:127-134
    assert rs.bucket_duration_us(None, "5 minutes")[0] == 300_000_000
    assert rs.bucket_duration_us(None, "hr")[0] == 3_600_000_000
    assert rs.bucket_duration_us(None, "day")[0] == 86_400_000_000
    with pytest.raises(ValueError):
        rs.checkAllowableFreq(None, "minutes")


def test_key_encoding_is_shared_across_frames():
    import pandas as pd
    from tempo_amd import columns
    a = pd.DataFrame({"k": ["x", "y", None], "j": [1, 1, 2]})
    b = pd.DataFrame({"k": ["y", "z"], "j": [1, 2]})
    (ca, cb), n = columns.encode_keys([a, b], ["k"])
    assert ca[1] == cb[0] and len({*ca, *cb}) == n == 4  # NULL is its own key
    (ca, cb), n = columns.encode_keys([a, b], ["k", "j"])
    assert ca[1] == cb[0] and n == 4
    (ca, cb), n = columns.encode_keys([a, b], [])
    assert ca is None and cb is None and n == 1


def test_table_path_stays_inside_the_warehouse(tmp_path):
    """io.write replaces a directory: the table name may not be empty, absolute or climb out of
    the warehouse, and the path it resolves to lies strictly below the warehouse root."""
    import os

    import pytest

    from tempo_amd import io as tio
    root = str(tmp_path / "wh")
    assert tio.table_path(root, "my_table") == os.path.join(os.path.realpath(root), "my_table")
    assert tio.table_path(root, "db/tab") == os.path.join(os.path.realpath(root), "db", "tab")
    for bad in ("", "  ", ".", "..", "../x", "a/../..", "/etc", str(tmp_path), "a/../../wh2"):
        with pytest.raises(ValueError):
            tio.table_path(root, bad)
    with pytest.raises(ValueError):
        tio.table_path(root, None)


def test_constructor_takes_arrow_tables_and_spark_frames():
    """python/tempo/tsdf.py:24-36 holds a Spark DataFrame; here a pyarrow Table is kept as it is (pandas
    only on request) and anything from a pyspark module is taken over through its Arrow export."""
    import pandas as pd
    import pyarrow as pa
    from tempo_amd import TSDF
    frame = pd.DataFrame({"symbol": ["a", "b"], "event_ts": pd.to_datetime(["2020-01-01", "2020-01-02"]), "x": [1.0, 2.0]})
    table = pa.Table.from_pandas(frame, preserve_index=False)
    t = TSDF(table, partition_cols=["symbol"])
    assert t._pandas is None and t.table is table and t.partitionCols == ["symbol"]
    assert t.df.equals(frame) and t._pandas is not None
    with pytest.raises(ValueError):
        TSDF(table, ts_col="nope")
    with pytest.raises(TypeError):
        TSDF(table, ts_col=3)
    with pytest.raises(TypeError):
        TSDF([1, 2, 3])

    class FakeSparkFrame:   # stands in for pyspark.sql.DataFrame: only its module name and toArrow() matter
        def toArrow(self):
            return table
    FakeSparkFrame.__module__ = "pyspark.sql.dataframe"
    s = TSDF(FakeSparkFrame(), partition_cols=["symbol"])
    assert s.table is table and s.ts_col == "event_ts"
