"""The reference's own unit tests (python/tests/tsdf_tests.py) restated over pandas frames.

`run_all(TSDF)` drives them through a TSDF class; the GPU suite passes the real class, the CPU
suite passes the class with its device operators swapped for the oracle (host logic only)."""
import json
import os

import numpy as np
import pytest
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _golden():
    with open(os.path.join(ROOT, "tests", "golden", "reference_vectors.json")) as f:
        return json.load(f)


def build_df(columns, rows, ts_cols=("event_ts",), float32=()):
    """buildTestDF of tsdf_tests.py:33-48: strings -> timestamps, FloatType -> float32."""
    df = pd.DataFrame(rows, columns=columns)
    for c in ts_cols:
        if c in df.columns:
            df[c] = pd.to_datetime(df[c])
    for c in float32:
        if c in df.columns:
            df[c] = df[c].astype(np.float32)
    return df


def assert_frames_equal(a, b, decimals=None):
    """assertDataFramesEqual of tsdf_tests.py:88-103: same column set, same multiset of rows."""
    assert sorted(a.columns) == sorted(b.columns), (sorted(a.columns), sorted(b.columns))
    cols = sorted(a.columns)

    def norm(df):
        out = []
        for row in df[cols].itertuples(index=False):
            r = []
            for v in row:
                if v is None or v is pd.NaT or (isinstance(v, float) and np.isnan(v)) or v is pd.NA:
                    r.append(None)
                elif isinstance(v, (float, np.floating)):
                    r.append(round(float(np.float32(v)) if decimals is None else float(v), 4 if decimals is None else decimals))
                elif isinstance(v, (np.integer,)):
                    r.append(int(v))
                else:
                    r.append(v)
            out.append(tuple(r))
        return sorted(out, key=repr)
    na, nb = norm(a), norm(b)
    assert na == nb, f"\n{na}\n!=\n{nb}"


def case_asof_basic(TSDF):
    fx = _golden()["asof_basic"]
    left = build_df(fx["left_columns"], fx["left"], float32=["trade_pr"])
    right = build_df(fx["right_columns"], fx["right"], float32=["bid_pr", "ask_pr"])
    expected = build_df(fx["expected_columns"], fx["expected"], ts_cols=["left_event_ts", "right_event_ts"],
                        float32=["left_trade_pr", "right_bid_pr", "right_ask_pr"])
    tl = TSDF(left, ts_col="event_ts", partition_cols=["symbol"])
    tr = TSDF(right, ts_col="event_ts", partition_cols=["symbol"])
    joined = tl.asofJoin(tr, left_prefix="left", right_prefix="right")
    assert joined.ts_col == "left_event_ts" and joined.partitionCols == ["symbol"]
    assert_frames_equal(joined.df, expected)
    exp2 = build_df(fx["expected_columns_no_right_prefix"], fx["expected"], ts_cols=["left_event_ts", "event_ts"],
                    float32=["left_trade_pr", "bid_pr", "ask_pr"])
    assert_frames_equal(tl.asofJoin(tr, left_prefix="left", right_prefix='').df, exp2)


def case_asof_skip_nulls(TSDF):
    fx = _golden()["asof_skip_nulls"]
    left = build_df(fx["left_columns"], fx["left"], float32=["trade_pr"])
    right = build_df(fx["right_columns"], fx["right"], float32=["bid_pr", "ask_pr"])
    f32 = ["left_trade_pr", "right_bid_pr", "right_ask_pr"]
    tl = TSDF(left, ts_col="event_ts", partition_cols=["symbol"])
    tr = TSDF(right, ts_col="event_ts", partition_cols=["symbol"])
    exp = build_df(fx["expected_columns"], fx["expected_skip_nulls"], ts_cols=["left_event_ts", "right_event_ts"], float32=f32)
    assert_frames_equal(tl.asofJoin(tr, left_prefix="left", right_prefix="right").df, exp)
    exp = build_df(fx["expected_columns"], fx["expected_skip_nulls_disabled"], ts_cols=["left_event_ts", "right_event_ts"], float32=f32)
    assert_frames_equal(tl.asofJoin(tr, left_prefix="left", right_prefix="right", skipNulls=False).df, exp)


def case_asof_sequence(TSDF):
    fx = _golden()["asof_sequence"]
    left = build_df(fx["left_columns"], fx["left"], float32=["trade_pr"])
    right = build_df(fx["right_columns"], fx["right"], float32=["bid_pr", "ask_pr"])
    exp = build_df(fx["expected_columns"], fx["expected"], ts_cols=["event_ts", "right_event_ts"],
                   float32=["trade_pr", "right_bid_pr", "right_ask_pr"])
    tl = TSDF(left, partition_cols=["symbol"])
    tr = TSDF(right, partition_cols=["symbol"], sequence_col="seq_nb")
    assert_frames_equal(tl.asofJoin(tr, right_prefix='right').df, exp)


def case_asof_skew(TSDF):
    fx = _golden()["asof_skew"]
    left = build_df(fx["left_columns"], fx["left"], float32=["trade_pr"])
    right = build_df(fx["right_columns"], fx["right"], float32=["bid_pr", "ask_pr"])
    exp = build_df(fx["expected_columns"], fx["expected"], ts_cols=["left_event_ts", "right_event_ts"],
                   float32=["left_trade_pr", "right_bid_pr", "right_ask_pr"])
    tl = TSDF(left, ts_col="event_ts", partition_cols=["symbol"])
    tr = TSDF(right, ts_col="event_ts", partition_cols=["symbol"])
    joined = tl.asofJoin(tr, left_prefix="left", right_prefix="right", tsPartitionVal=fx["tsPartitionVal"],
                         fraction=fx["fraction"])
    assert_frames_equal(joined.df, exp)
    # scala AsOfJoinTests.scala:106-108
    assert len(tl.asofJoin(tr, left_prefix="left", right_prefix="right", tsPartitionVal=1, fraction=0.1).df) == 7


def case_range_stats(TSDF):
    fx = _golden()["range_stats"]
    df = build_df(fx["columns"], fx["data"], float32=fx["float32_columns"])
    exp = build_df(fx["expected_columns"], fx["expected"])
    out = TSDF(df, partition_cols=["symbol"]).withRangeStats(rangeBackWindowSecs=fx["rangeBackWindowSecs"]).df
    assert list(out.columns) == ["symbol", "event_ts", "trade_pr", "mean_trade_pr", "count_trade_pr", "min_trade_pr",
                                 "max_trade_pr", "sum_trade_pr", "stddev_trade_pr", "zscore_trade_pr"]
    assert out["min_trade_pr"].dtype == np.float32 and out["count_trade_pr"].dtype == np.int64
    assert_frames_equal(out.drop(columns=["trade_pr"]), exp, decimals=2)  # cast decimal(5,2), tsdf_tests.py:482-499


def case_grouped_stats(TSDF):
    """tsdf_tests.py:504-564."""
    fx = _golden()["grouped_stats"]
    df = build_df(fx["columns"], fx["data"], float32=fx["float32_columns"])
    df["index"] = df["index"].astype(np.int32)
    exp = build_df(fx["expected_columns"], fx["expected"])
    out = TSDF(df, partition_cols=["symbol"]).withGroupedStats(freq=fx["freq"]).df
    stats = ["mean_", "count_", "min_", "max_", "sum_", "stddev_"]
    assert list(out.columns) == (["symbol"] + [s + "trade_pr" for s in stats] + [s + "index" for s in stats]
                                 + ["event_ts"])
    assert out["min_trade_pr"].dtype == np.float32 and out["count_trade_pr"].dtype == np.int64
    assert out["sum_index"].tolist() == [2, 2] and out["min_index"].dtype == np.int32
    assert out["stddev_index"].tolist() == [0.0, 0.0] and out["mean_index"].tolist() == [1.0, 1.0]
    assert_frames_equal(out[fx["expected_columns"]], exp, decimals=2)  # cast decimal(5,2), tsdf_tests.py:546-561
    # explicit metric list, coarser bucket: everything lands in one bucket per month
    one = TSDF(df, partition_cols=["symbol"]).withGroupedStats(metricCols=["trade_pr"], freq="1 day").df
    assert list(one.columns) == ["symbol"] + [s + "trade_pr" for s in stats] + ["event_ts"]
    assert one["count_trade_pr"].tolist() == [2, 2]


def case_interpolate(TSDF):
    """interpol_tests.py:154-595: every fill method on the simple series, the TSDF entry points,
    the resample().interpolate() chain and the validation errors."""
    fx = _golden()["interpolate"]
    df = build_df(fx["columns"], fx["data"], float32=fx["float32_columns"])
    t = TSDF(df, partition_cols=["partition_a", "partition_b"], ts_col="event_ts")
    for method, rows in fx["expected"].items():
        exp = build_df(fx["expected_columns"], rows)
        got = t.interpolate(freq=fx["freq"], func=fx["func"], method=method, show_interpolated=True,
                            target_cols=["value_a", "value_b"]).df
        assert list(got.columns) == fx["expected_columns"], method
        assert got["value_a"].dtype == np.float64 and got["is_ts_interpolated"].dtype == bool
        assert got["event_ts"].tolist() == exp["event_ts"].tolist(), method   # rows come back in (key, ts) order
        assert_frames_equal(got, exp)
    # defaults: ts_col / partition_cols from the TSDF, every numeric column, flags hidden (:406-444)
    lin = build_df(fx["expected_columns"], fx["expected"]["linear"])
    got = t.interpolate(freq="30 seconds", func="mean", method="linear").df
    assert list(got.columns) == fx["columns"]
    assert_frames_equal(got, lin[fx["columns"]])
    # another ts_col name, one target (:446-495)
    t2 = TSDF(df.rename(columns={"event_ts": "other_ts_col"}), partition_cols=["partition_a", "partition_b"],
              ts_col="other_ts_col")
    got = t2.interpolate(ts_col="other_ts_col", show_interpolated=True, partition_cols=["partition_a", "partition_b"],
                         target_cols=["value_a"], freq="30 seconds", func="mean", method="linear").df
    exp = lin.rename(columns={"event_ts": "other_ts_col"})[
        ["partition_a", "partition_b", "other_ts_col", "value_a", "is_ts_interpolated", "is_interpolated_value_a"]]
    assert list(got.columns) == list(exp.columns)
    assert_frames_equal(got, exp)
    # constructor parameters of the result (:497-512)
    r = t.interpolate(ts_col="event_ts", show_interpolated=True, partition_cols=["partition_b"],
                      target_cols=["value_a"], freq="30 seconds", func="mean", method="linear")
    assert r.ts_col == "event_ts" and r.partitionCols == ["partition_b"]
    # resample(...).interpolate(...) (:514-595)
    got = t.resample(freq="30 seconds", func="mean", fill=None).interpolate(
        method="linear", target_cols=["value_a"], show_interpolated=True).df
    assert_frames_equal(got, lin[["partition_a", "partition_b", "event_ts", "value_a", "is_ts_interpolated",
                                  "is_interpolated_value_a"]])
    got = t.resample(freq="30 seconds", func="mean", fill=None).interpolate(method="ffill").df
    assert_frames_equal(got, build_df(fx["expected_columns"], fx["expected"]["ffill"])[fx["columns"]])
    # validation (:78-152)
    kw = dict(freq="30 seconds", func="mean", show_interpolated=True)
    with pytest.raises(ValueError, match="fill options"):
        t.interpolate(method="abcd", target_cols=["value_a", "value_b"], **kw)
    with pytest.raises(ValueError, match="Target Column needs to be one of"):
        t.interpolate(method="zero", target_cols=["partition_a", "value_b"], **kw)
    with pytest.raises(ValueError):
        t.interpolate(method="zero", partition_cols=["partition_c", "partition_b"],
                      target_cols=["value_a", "value_b"], **kw)
    with pytest.raises(ValueError):
        t.interpolate(method="zero", ts_col="value_a", target_cols=["value_a", "value_b"], **kw)


def case_resample(TSDF):
    fx = _golden()["resample"]
    df = build_df(fx["columns"], fx["data"], float32=fx["float32_columns"])
    t = TSDF(df, partition_cols=["symbol"])
    c = fx["floor_min"]
    res = t.resample(freq=c["freq"], func=c["func"], prefix=c["prefix"])
    assert type(res).__name__ == "_ResampledTSDF"
    assert list(res.df.columns) == ["symbol", "event_ts", "floor_date", "floor_trade_pr", "floor_trade_pr_2"]
    assert_frames_equal(res.df, build_df(c["expected_columns"], c["expected"], float32=["floor_trade_pr", "floor_trade_pr_2"]))
    c = fx["mean_5min"]
    out = t.resample(freq=c["freq"], func=c["func"]).df
    assert out["trade_pr"].dtype == np.float64 and out["date"].dtype == np.float64
    out["trade_pr"] = out["trade_pr"].round(2)
    assert_frames_equal(out, build_df(c["expected_columns"], c["expected"]), decimals=2)
    c = fx["calc_bars"]
    bars = t.calc_bars(freq=c["freq"], metricCols=c["metricCols"]).df
    assert list(bars.columns) == c["expected_columns"]
    assert_frames_equal(bars, build_df(c["expected_columns"], c["expected"]))


def case_resample_fill(TSDF):
    g = _golden()
    fx, c = g["resample"], g["resample_fill"]
    df = build_df(fx["columns"], fx["data"], float32=fx["float32_columns"])
    out = TSDF(df, partition_cols=["symbol"]).resample(freq=c["freq"], func=c["func"], fill=True).df
    out["trade_pr"] = out["trade_pr"].round(2)
    keep = out[out["event_ts"].isin(pd.to_datetime(c["filter_event_ts"]))]
    assert_frames_equal(keep, build_df(c["expected_columns"], c["expected"]), decimals=2)
    assert len(out) > 4  # the grid spans every 5-minute bucket between the first and the last


def case_ema(TSDF):
    fx = _golden()["ema_scala"]
    df = build_df(fx["columns"], fx["data"])
    out = TSDF(df, partition_cols=["symbol"]).EMA("trade_pr", window=fx["python_window"], exp_factor=fx["exp_factor"])
    assert out.df["EMA_trade_pr"].tolist() == fx["expected_ema"] and out.sequence_col == ''


def case_lookback(TSDF):
    df = pd.DataFrame({"symbol": ["a", "a", "a", "b", "b", "a"],
                       "event_ts": pd.to_datetime(["2020-01-01 00:00:01", "2020-01-01 00:00:02", "2020-01-01 00:00:03",
                                                   "2020-01-01 00:00:01", "2020-01-01 00:00:02", "2020-01-01 00:00:04"]),
                       "x": [1.0, 2.0, 3.0, 10.0, 20.0, 4.0], "y": [1, 2, 3, 10, 20, 4]})
    t = TSDF(df, partition_cols=["symbol"])
    exact = t.withLookbackFeatures(["x", "y"], 2)
    assert isinstance(exact, pd.DataFrame)  # tsdf.py:668-669 returns the bare DataFrame
    assert exact["x"].tolist() == [3.0, 4.0]
    assert [f.tolist() for f in exact["features"]] == [[[1.0, 1.0], [2.0, 2.0]], [[2.0, 2.0], [3.0, 3.0]]]
    loose = t.withLookbackFeatures(["x"], 2, exactSize=False, featureColName="hist")
    assert isinstance(loose, TSDF)
    assert [f.tolist() for f in loose.df["hist"]] == [[], [[1.0]], [[1.0], [2.0]], [[2.0], [3.0]], [], [[10.0]]]


def case_errors(TSDF):
    import pytest
    df = pd.DataFrame({"symbol": ["a"], "event_ts": pd.to_datetime(["2020-01-01"]), "x": [1.0]})
    with pytest.raises(TypeError):
        TSDF(df, ts_col=0)
    with pytest.raises(ValueError):
        TSDF(df, ts_col="nope")
    with pytest.raises(TypeError):
        TSDF(df, partition_cols=("symbol",))
    assert TSDF(df, partition_cols="symbol").partitionCols == ["symbol"]
    assert TSDF(df).sequence_col == '' and TSDF(df).partitionCols == []
    t = TSDF(df, partition_cols=["symbol"])
    other = TSDF(df.rename(columns={"symbol": "sym"}), partition_cols=["sym"])
    with pytest.raises(ValueError, match="partition columns"):
        t.asofJoin(other)
    ints = TSDF(df.assign(event_ts=[5]), partition_cols=["symbol"])
    with pytest.raises(ValueError, match="timestamp index"):
        t.asofJoin(ints)
    with pytest.raises(ValueError, match="null skipping"):
        t.asofJoin(t, tsPartitionVal=10, skipNulls=False)
    with pytest.raises(ValueError, match="Aggregate function missing"):
        t.resample(freq="min")
    with pytest.raises(ValueError, match="not in the valid list"):
        t.resample(freq="min", func="median")
    with pytest.raises(ValueError, match="Allowable grouping frequencies"):
        t.resample(freq="fortnight", func="mean")
    with pytest.raises(Exception, match="select statement"):
        t.select("x")
    assert t.select("symbol", "event_ts").df.columns.tolist() == ["symbol", "event_ts"]


def case_write_to_delta(TSDF):
    """tsdf_tests.py:746-787: the written table holds every row; derived columns, column rotation,
    date partitioning and the Delta log as io.py:29-35 produce them."""
    import json
    import os
    import tempfile

    from tempo_amd import io as tio
    fx = _golden()["write_to_delta"]
    df = build_df(fx["columns"], fx["data"], float32=fx["float32_columns"])
    # shuffle the rows: the writer has to restore (event_dt, key, ts) order on its own
    shuffled = df.iloc[[5, 2, 6, 0, 4, 1, 3]].reset_index(drop=True)
    with tempfile.TemporaryDirectory() as root:
        TSDF(shuffled, partition_cols=["symbol"]).write(root, "my_table")
        back = tio.read_table(root, "my_table")
        assert len(back) == fx["expected_count"]
        assert list(back.columns) == ["event_time"] + fx["columns"] + ["event_dt"]   # io.py:31-33
        assert back["event_ts"].tolist() == df["event_ts"].tolist()                  # clustered by (date, key, ts)
        assert back["event_time"].tolist() == fx["event_time"]
        assert [str(d)[:10] for d in back["event_dt"]] == fx["event_dt"]
        assert back["trade_pr"].dtype == np.float32 and back["trade_pr"].tolist() == df["trade_pr"].tolist()
        path = tio.table_path(root, "my_table")
        assert sorted(d for d in os.listdir(path) if d.startswith("event_dt=")) == \
            ["event_dt=2020-08-01", "event_dt=2020-09-01"]
        with open(os.path.join(path, "_delta_log", "00000000000000000000.json")) as f:
            actions = [json.loads(line) for line in f]
        meta = [a["metaData"] for a in actions if "metaData" in a][0]
        assert meta["partitionColumns"] == ["event_dt"]
        types = {fld["name"]: fld["type"] for fld in json.loads(meta["schemaString"])["fields"]}
        assert types == {"event_time": "double", "symbol": "string", "date": "string", "event_ts": "timestamp",
                         "trade_pr": "float", "trade_pr_2": "float", "event_dt": "date"}
        assert len([a for a in actions if "add" in a]) == 2
        # overwrite mode: a second write replaces the table
        TSDF(df.iloc[:3], partition_cols=["symbol"]).write(root, "my_table", optimizationCols=["trade_pr"])
        assert len(tio.read_table(root, "my_table")) == 3
        with pytest.raises(ValueError):
            TSDF(df, partition_cols=["symbol"]).write(root, "my_table", optimizationCols=["nope"])
        # only a Delta table is ever replaced, and only inside the warehouse
        os.makedirs(os.path.join(root, "plain_dir"))
        with open(os.path.join(root, "plain_dir", "keep.txt"), "w") as f:
            f.write("x")
        with pytest.raises(ValueError, match="not a Delta table"):
            TSDF(df, partition_cols=["symbol"]).write(root, "plain_dir")
        assert os.path.exists(os.path.join(root, "plain_dir", "keep.txt"))
        with pytest.raises(ValueError):
            TSDF(df, partition_cols=["symbol"]).write(root, "../escaped")
        assert sorted(os.listdir(root)) == ["my_table", "plain_dir"]     # no staging directory left behind


ALL_CASES = [case_asof_basic, case_asof_skip_nulls, case_asof_sequence, case_asof_skew, case_range_stats,
             case_grouped_stats, case_interpolate, case_write_to_delta, case_resample, case_resample_fill, case_ema, case_lookback, case_errors]
