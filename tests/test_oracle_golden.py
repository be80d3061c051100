"""The CPU oracle against every golden vector the reference's own tests hold for the hot path
(tests/golden/README.md lists the reference file:line of each fixture)."""
import numpy as np
import pytest

from conftest import ts_us


def _table(cols, rows):
    return {c: [r[i] for r in rows] for i, c in enumerate(cols)}


def _valid(vals):
    return np.array([v is not None for v in vals], np.uint8)


def _keys(vals):
    uniq = sorted(set(vals))
    return np.array([uniq.index(v) for v in vals], np.int32)


def _asof(orc, fx, expected, **kw):
    L = _table(fx["left_columns"], fx["left"])
    R = _table(fx["right_columns"], fx["right"])
    tl = np.array([ts_us(s) for s in L["event_ts"]])
    tr = np.array([ts_us(s) for s in R["event_ts"]])
    rcols = [c for c in fx["right_columns"] if c != "symbol"]
    valid = [_valid(R[c]) for c in rcols]
    seq = kw.pop("seq", None)
    idx, missing = orc.asof_join(_keys(L["symbol"]), tl, _keys(R["symbol"]), tr,
                                 seq_r=None if seq is None else np.array(R[seq]),
                                 valid_r=valid, **kw)
    ecols = fx["expected_columns"]
    for i, erow in enumerate(expected):
        for c, name in enumerate(rcols):
            pos = [k for k, e in enumerate(ecols) if e in ("right_" + name,)][0]
            j = idx[c, i]
            got = None if j < 0 else R[name][j]
            assert got == erow[pos], (i, name, got, erow[pos])
    return idx, missing


def test_asof_basic(orc, golden):
    fx = golden["asof_basic"]
    _asof(orc, fx, fx["expected"])


def test_asof_skip_nulls(orc, golden):
    fx = golden["asof_skip_nulls"]
    _asof(orc, fx, fx["expected_skip_nulls"], skip_nulls=True)
    _asof(orc, fx, fx["expected_skip_nulls_disabled"], skip_nulls=False)


def test_asof_sequence(orc, golden):
    fx = golden["asof_sequence"]
    _asof(orc, fx, fx["expected"], seq="seq_nb")


def test_asof_skew(orc, golden):
    fx = golden["asof_skew"]
    idx, missing = _asof(orc, fx, fx["expected"], ts_partition_val=fx["tsPartitionVal"],
                         fraction=fx["fraction"])
    assert missing.tolist() == [0, 0, 0]
    # scala AsOfJoinTests.scala:106-108: tsPartitionVal=1 still yields one output row per left row
    L = _table(fx["left_columns"], fx["left"])
    R = _table(fx["right_columns"], fx["right"])
    idx1, _ = orc.asof_join(_keys(L["symbol"]), np.array([ts_us(s) for s in L["event_ts"]]),
                            _keys(R["symbol"]), np.array([ts_us(s) for s in R["event_ts"]]),
                            valid_r=[None], ts_partition_val=1, fraction=0.1)
    assert idx1.shape == (1, 7)


def _dec2(v):
    return None if v is None else round(float(v) + 0.0, 2)


def test_range_stats(orc, golden):
    fx = golden["range_stats"]
    T = _table(fx["columns"], fx["data"])
    ts = np.array([ts_us(s) for s in T["event_ts"]])
    x = np.array(T["trade_pr"], np.float32).astype(np.float64)  # FloatType column widened
    out = orc.range_stats(_keys(T["symbol"]), ts, x, None, fx["rangeBackWindowSecs"])
    assert out["perm"].tolist() == [0, 1, 2, 3]
    for p, erow in enumerate(fx["expected"]):
        e = dict(zip(fx["expected_columns"], erow))
        assert out["count"][p] == e["count_trade_pr"]
        for name in ("mean", "min", "max", "sum"):
            assert abs(out[name][p] - e[name + "_trade_pr"]) < 0.00501, (p, name)
        if e["stddev_trade_pr"] is None:
            assert not out["stddev_valid"][p] and not out["zscore_valid"][p]
        else:
            assert out["stddev_valid"][p] and out["zscore_valid"][p]
            assert abs(out["stddev"][p] - e["stddev_trade_pr"]) < 0.00501
            assert abs(out["zscore"][p] - e["zscore_trade_pr"]) < 0.00501


def test_ema_scala_vector(orc, golden):
    fx = golden["ema_scala"]
    T = _table(fx["columns"], fx["data"])
    ts = np.array([ts_us(s) for s in T["event_ts"]])
    perm, out = orc.ema(_keys(T["symbol"]), ts, np.array(T["trade_pr"]), None,
                        window=fx["python_window"], exp_factor=fx["exp_factor"])
    assert perm.tolist() == list(range(6))
    assert out.tolist() == fx["expected_ema"]


def test_resample(orc, golden):
    fx = golden["resample"]
    T = _table(fx["columns"], fx["data"])
    ts = np.array([ts_us(s) for s in T["event_ts"]])
    key = _keys(T["symbol"])
    pr = np.array(T["trade_pr"], np.float32).astype(np.float64)
    pr2 = np.array(T["trade_pr_2"], np.float32).astype(np.float64)
    date_code = np.zeros(len(ts), np.int64)  # single distinct string -> code 0
    # floor, 1 minute: whole row with the smallest (ts, date, trade_pr, trade_pr_2) struct
    r = orc.resample(key, ts, 60_000_000, "floor", [date_code, pr, pr2], [None, None, None])
    e = fx["floor_min"]["expected"]
    assert r["n"] == len(e)
    for b, erow in enumerate(e):
        assert r["start"][b] == ts_us(erow[1])
        row = r["row"][b]
        assert np.float32(pr[row]) == np.float32(erow[2]) and pr2[row] == erow[4]
    # mean, 5 minutes
    r = orc.resample(key, ts, 300_000_000, "mean", [pr, pr2], [None, None])
    e = fx["mean_5min"]["expected"]
    assert r["n"] == len(e)
    for b, erow in enumerate(e):
        assert r["start"][b] == ts_us(erow[1])
        assert round(r["f64"][0][b], 2) == erow[3]
        assert r["f64"][1][b] == erow[4]
    # calc_bars = floor/min/max/ceil per minute (open/low/high/close)
    e = fx["calc_bars"]["expected"]
    ro = orc.resample(key, ts, 60_000_000, "floor", [pr, pr2], [None, None])
    rc = orc.resample(key, ts, 60_000_000, "ceil", [pr, pr2], [None, None])
    rl = orc.resample(key, ts, 60_000_000, "min", [pr, pr2], [None, None])
    rh = orc.resample(key, ts, 60_000_000, "max", [pr, pr2], [None, None])
    f32 = lambda v: float(np.float32(v))
    for b, erow in enumerate(e):
        d = dict(zip(fx["calc_bars"]["expected_columns"], erow))
        assert f32(pr[rc["row"][b]]) == f32(d["close_trade_pr"]) and pr2[rc["row"][b]] == d["close_trade_pr_2"]
        assert f32(pr[ro["row"][b]]) == f32(d["open_trade_pr"]) and pr2[ro["row"][b]] == d["open_trade_pr_2"]
        assert f32(rh["f64"][0][b]) == f32(d["high_trade_pr"]) and rh["f64"][1][b] == d["high_trade_pr_2"]
        assert f32(rl["f64"][0][b]) == f32(d["low_trade_pr"]) and rl["f64"][1][b] == d["low_trade_pr_2"]


def test_grouped_stats_vector_via_bucket_stats(orc, golden):
    """withGroupedStats golden (next-tier API) pins the bucket + mean path of the oracle."""
    fx = golden["grouped_stats"]
    T = _table(fx["columns"], fx["data"])
    ts = np.array([ts_us(s) for s in T["event_ts"]])
    pr = np.array(T["trade_pr"], np.float32).astype(np.float64)
    r = orc.resample(_keys(T["symbol"]), ts, 60_000_000, "mean", [pr], [None])
    for b, erow in enumerate(fx["expected"]):
        assert r["start"][b] == ts_us(erow[1])
        assert abs(r["f64"][0][b] - erow[2]) < 0.00501


@pytest.mark.parametrize("method", ["zero", "null", "bfill", "ffill", "linear"])
def test_interpolate_vectors(orc, golden, method):
    """interpol_tests.py:154-352 on the oracle itself: the sampled series (30 s means of
    simple_data, computed by the oracle's resample) upsampled and filled."""
    fx = golden["interpolate"]
    T = _table(fx["columns"], fx["data"])
    ts = np.array([ts_us(s) for s in T["event_ts"]])
    cols, valid = [], []
    for c in ("value_a", "value_b"):
        v = np.array([np.nan if x is None else x for x in T[c]], np.float32).astype(np.float64)
        valid.append((~np.isnan(v)).astype(np.uint8))
        cols.append(np.nan_to_num(v))
    key = np.zeros(len(ts), np.int32)
    rs = orc.resample(key, ts, 30_000_000, "mean", cols, valid)
    r = orc.interpolate(rs["key"], rs["start"], 30_000_000, method,
                        [rs["f64"][0], rs["f64"][1]], [rs["valid"][0], rs["valid"][1]])
    exp = fx["expected"][method]
    assert r["n"] == len(exp)
    for q, row in enumerate(exp):
        assert r["ts"][q] == ts_us(row[2])
        for c in range(2):
            if row[3 + c] is None:
                assert r["valid"][c][q] == 0, (q, c)
            else:
                assert r["valid"][c][q] == 1 and r["val"][c][q] == row[3 + c], (q, c)
        assert bool(r["ts_interp"][q]) == row[5]
        assert bool(r["interp"][0][q]) == row[6] and bool(r["interp"][1][q]) == row[7]


def test_write_layout_vectors(orc, golden):
    """io.py:29-39 on the oracle: derived columns of the reference's write test input
    (tsdf_tests.py:760-766) and the (event_dt, key, ts) order; fractional seconds print with
    their trailing zeros dropped, as Spark's timestamp-to-string cast does."""
    fx = golden["write_to_delta"]
    ts = np.array([ts_us(r[2]) for r in fx["data"]])
    order = np.array([5, 2, 6, 0, 4, 1, 3])
    dt, tm, perm = orc.write_layout(np.zeros(7, np.int32), ts[order])
    assert [str(np.datetime64(int(d), "D")) for d in dt[np.argsort(order)]] == fx["event_dt"]
    assert tm[np.argsort(order)].tolist() == fx["event_time"]
    assert order[perm].tolist() == list(range(7))
    # 23:59:59.5 -> "235959.5"; 00:00:00.000001 -> "000000.000001"; before the epoch dates floor
    odd = np.array([ts_us("2021-03-04 23:59:59") + 500_000, ts_us("2021-03-04 00:00:00") + 1,
                    ts_us("1969-12-31 12:30:00") + 250_000])
    dt, tm, perm = orc.write_layout(None, odd)
    assert tm.tolist() == [235959.5, 0.000001, 123000.25]
    assert dt.tolist() == [18690, 18690, -1] and perm.tolist() == [2, 1, 0]
