"""Arrow C data interface import (tmp_arrow_import) against pyarrow's own conversions: every
storage type, sliced arrays (offsets that are not multiples of 8), chunked arrays, dictionary
arrays, timestamps in all four units, NULLs; and the Spark-side glue on plain pyarrow batches."""
import zlib

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pa = pytest.importorskip("pyarrow")
pytestmark = pytest.mark.gpu


def _expect(arr, kind):
    arr = arr.combine_chunks() if isinstance(arr, pa.ChunkedArray) else arr
    if pa.types.is_dictionary(arr.type):
        arr = arr.indices
    ok = np.asarray(arr.is_valid())
    if pa.types.is_timestamp(arr.type):
        raw = np.asarray(arr.cast(pa.int64()).fill_null(0))
        if kind == "i64":
            unit = arr.type.unit
            raw = {"s": raw * 1_000_000, "ms": raw * 1000, "us": raw, "ns": np.floor_divide(raw, 1000)}[unit]
        vals = raw
    elif kind == "f64":
        vals = np.asarray(arr.cast(pa.float64()).fill_null(0.0))
    else:
        vals = np.asarray(arr.cast(pa.int64()).fill_null(0))
        if kind == "i32":
            vals = vals.astype(np.int32)
    return np.where(ok, vals, 0), ok


CASES = [
    (pa.int8(), "i64"), (pa.uint8(), "i32"), (pa.int16(), "f64"), (pa.uint16(), "i64"), (pa.int32(), "i32"),
    (pa.int32(), "i64"), (pa.uint32(), "i64"), (pa.int64(), "i64"), (pa.int64(), "f64"), (pa.float32(), "f64"),
    (pa.float64(), "f64"),
]


@pytest.mark.parametrize("typ,kind", CASES)
@pytest.mark.parametrize("with_nulls", [False, True])
def test_import_numeric(typ, kind, with_nulls):
    from tempo_amd import ops
    rng = np.random.default_rng(zlib.crc32(("%s/%s" % (typ, kind)).encode()))
    n = 70_003
    if pa.types.is_floating(typ):
        vals = rng.normal(0, 1e3, n)
    else:
        info = np.iinfo(typ.to_pandas_dtype())
        vals = rng.integers(max(info.min, -2**31), min(info.max, 2**31 - 1), n)
    mask = (rng.random(n) < 0.2) if with_nulls else None
    arr = pa.array(vals, type=typ, mask=mask)
    for piece in (arr, arr.slice(13, 50_001), arr.slice(8, 9), pa.chunked_array([arr.slice(0, 5), arr.slice(5, 0), arr.slice(5)])):
        got, valid, nulls = ops.arrow_import(piece, kind)
        want, ok = _expect(piece, kind)
        assert nulls == int((~ok).sum())
        assert (valid is None) == (piece.null_count == 0)
        if valid is not None:
            assert np.array_equal(valid.cpu().numpy().astype(bool), ok)
        assert np.array_equal(got.cpu().numpy(), want.astype(got.cpu().numpy().dtype))


@pytest.mark.parametrize("unit", ["s", "ms", "us", "ns"])
def test_import_timestamps(unit):
    from tempo_amd import ops
    rng = np.random.default_rng(5)
    raw = rng.integers(-10**12, 10**12, 20_000)            # before and after the epoch
    arr = pa.array(raw, type=pa.int64()).cast(pa.timestamp(unit, tz="UTC"))
    for kind in ("i64", "i64_raw"):
        got, valid, nulls = ops.arrow_import(arr.slice(3), kind)
        want, _ = _expect(arr.slice(3), kind)
        assert valid is None and nulls == 0 and np.array_equal(got.cpu().numpy(), want)
    with_null = pa.array([1, None, 3], type=pa.timestamp(unit))
    got, valid, nulls = ops.arrow_import(with_null, "i64")
    assert nulls == 1 and valid.cpu().tolist() == [1, 0, 1] and got.cpu().tolist()[1] == 0


def test_import_dictionary_and_errors():
    from tempo_amd import ops
    keys = pa.array(["b", "a", None, "b", "c", "a"] * 1000).dictionary_encode()
    got, valid, nulls = ops.arrow_import(keys, "i32")
    assert nulls == 1000 and got.dtype == torch.int32
    assert np.array_equal(got.cpu().numpy(), np.asarray(keys.indices.fill_null(0)))
    assert np.array_equal(valid.cpu().numpy().astype(bool), np.asarray(keys.is_valid()))
    empty, v, nulls = ops.arrow_import(pa.array([], type=pa.float64()), "f64")
    assert empty.numel() == 0 and v is None and nulls == 0
    with pytest.raises(NotImplementedError):
        ops.arrow_import(pa.array(["x", "y"]), "i64")            # strings are not imported as they are
    with pytest.raises(NotImplementedError):
        ops.arrow_import(pa.array([1.5, 2.5]), "i64")            # floats go to a double column only
    with pytest.raises(NotImplementedError):
        ops.arrow_import(pa.array([1, 2], type=pa.timestamp("us")), "i32")


def test_spark_glue_on_arrow_batches(orc):
    """What mapInArrow / cogrouped applyInArrow would call, fed with pyarrow batches: the
    results equal arrow_io on the whole table."""
    from tempo_amd import arrow_io, spark_glue
    rng = np.random.default_rng(11)
    n = 5000
    base = 1_596_240_000_000_000
    sym = rng.choice(["AAA", "BBB", "CC"], n)
    ts = pa.array(base + rng.integers(0, 600_000, n) * 1000, type=pa.int64()).cast(pa.timestamp("us"))
    price = pa.array(rng.normal(100, 5, n), mask=rng.random(n) < 0.1)
    qty = pa.array(rng.integers(1, 100, n), type=pa.int32())
    table = pa.table({"symbol": sym, "event_ts": ts, "price": price, "qty": qty})
    fn = spark_glue.range_stats_in_arrow("event_ts", ["symbol"], ["price", "qty"], 30, max_rows_per_batch=700)
    batches = list(fn(iter(table.to_batches(max_chunksize=999))))
    assert max(b.num_rows for b in batches) <= 700
    got = pa.Table.from_batches(batches)
    want = arrow_io.range_stats(table, "event_ts", ["symbol"], ["price", "qty"], 30)
    assert got.equals(want)
    fields = spark_glue.range_stats_fields(table.schema, ["price", "qty"])
    assert [f.name for f in fields] == got.column_names[4:]
    assert all(got.schema.field(f.name).type == f.type for f in fields)
    assert list(fn(iter([]))) == []
    quotes = pa.table({"symbol": rng.choice(["AAA", "BBB", "DD"], n),
                       "event_ts": pa.array(base + rng.integers(0, 600_000, n) * 1000, type=pa.int64()).cast(pa.timestamp("us")),
                       "bid": pa.array(rng.normal(99, 5, n), mask=rng.random(n) < 0.3)})
    join = spark_glue.asof_join_in_arrow("event_ts", ["symbol"])
    assert join(table, quotes).equals(arrow_io.asof_join(table, quotes, "event_ts", ["symbol"]))
    assert spark_glue.to_arrow(table.to_pandas()).num_rows == n and spark_glue.to_arrow(table) is table
