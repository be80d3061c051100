"""Arrow edge, host side only (no device): key dictionaries, timestamp units, payload encoding."""
import numpy as np
import pyarrow as pa
import pytest

from tempo_amd import arrow_io


def test_encode_keys_shared_dictionary_and_decode():
    a = pa.table({"sym": ["B", "A", None, "B"], "ven": ["x", "x", "y", "y"]})
    b = pa.table({"sym": ["A", "C"], "ven": ["x", "x"]})
    (ka, kb), n, decode = arrow_io.encode_keys([a, b], ["sym", "ven"])
    assert ka.dtype == np.int32 and n == 5            # (B,x) (A,x) (None,y) (B,y) (C,x)
    assert ka[1] == kb[0] and len({*ka.tolist(), *kb.tolist()}) == 5
    cols = decode(np.concatenate([ka, kb]))
    assert cols[0].to_pylist() == ["B", "A", None, "B", "A", "C"]
    assert cols[1].to_pylist() == ["x", "x", "y", "y", "x", "x"]
    (k1,), n1, _ = arrow_io.encode_keys([a], [])
    assert k1 is None and n1 == 1


@pytest.mark.parametrize("unit,scale", [("s", 1_000_000), ("ms", 1000), ("us", 1), ("ns", None)])
def test_timestamps_to_microseconds(unit, scale):
    raw = [1_596_240_000, 1_596_240_001]
    if unit == "ns":
        arr = pa.array([r * 10**9 + 1999 for r in raw], type=pa.timestamp("ns"))
        want = [r * 10**6 + 1 for r in raw]            # floor to microseconds
    else:
        arr = pa.array(raw, type=pa.timestamp(unit, tz="UTC" if unit == "ms" else None))
        want = [r * scale for r in raw]
    us, typ = arrow_io.timestamps_us(pa.chunked_array([arr.slice(0, 1), arr.slice(1)]))
    assert us.dtype == np.int64 and us.tolist() == want and typ == arr.type
    with pytest.raises(ValueError, match="NULL timestamps"):
        arrow_io.timestamps_us(pa.array([None, 1], type=pa.timestamp("us")))
    with pytest.raises(TypeError):
        arrow_io.timestamps_us(pa.array([1, 2]))


def test_many_wide_key_columns_do_not_collide():
    """Five partition columns of ~7000 distinct values each: the product of the cardinalities
    (1.7e19) exceeds int64, the combined key must still be collision free."""
    import pandas as pd

    from tempo_amd import columns
    rng = np.random.default_rng(0)
    n, ncol, card = 30_000, 5, 7000
    vals = rng.integers(0, card, (n, ncol))
    vals[n // 2:] = vals[: n - n // 2]                # every combination occurs twice
    names = [f"k{j}" for j in range(ncol)]
    df = pd.DataFrame({nm: vals[:, j] for j, nm in enumerate(names)})
    want = len(set(map(tuple, vals.tolist())))
    (codes,), n_keys = columns.encode_keys([df], names)
    assert n_keys == want and len(np.unique(codes)) == want
    assert np.array_equal(codes[: n - n // 2], codes[n // 2:])
    tbl = pa.table({nm: vals[:, j] for j, nm in enumerate(names)})
    (acodes,), an, decode = arrow_io.encode_keys([tbl], names)
    assert an == want and np.array_equal(acodes[: n - n // 2], acodes[n // 2:])
    back = decode(acodes)
    for j in range(ncol):
        assert back[j].to_pylist() == vals[:, j].tolist()


def test_arrow_import_refuses_values_that_would_wrap():
    """uint64 beyond int64, 64-bit indices beyond int32 and second / millisecond timestamps whose
    microseconds overflow are refused before anything is copied (no device needed to find out)."""
    from tempo_amd import ops
    with pytest.raises(OverflowError):
        ops.arrow_import(pa.array([1, 2 ** 63 + 1], type=pa.uint64()), "i64")
    with pytest.raises(OverflowError):
        ops.arrow_import(pa.array([0, 2 ** 31], type=pa.int64()), "i32")
    with pytest.raises(OverflowError):
        ops.arrow_import(pa.array([9_223_372_036_855], type=pa.timestamp("s")), "i64")   # x 10^6 > 2^63
    with pytest.raises(OverflowError):
        ops.arrow_import(pa.chunked_array([pa.array([1, None], type=pa.timestamp("ms")),
                                           pa.array([-9_223_372_036_854_776], type=pa.timestamp("ms"))]), "i64")


def test_threaded_dictionary_passes_equal_arrows_own(monkeypatch):
    """Long columns are dictionary-encoded and rebuilt from their codes slice by slice on several
    threads: dictionary, indices and the rebuilt column are those of one pyarrow call over the whole
    column -- NULL keys as a dictionary entry of their own or as NULL indices, slices without some of
    the values, a key that first appears in a late slice."""
    import numpy as np
    import pyarrow as pa
    import pyarrow.compute as pc
    from tempo_amd import arrow_io
    monkeypatch.setattr(arrow_io, "PARALLEL_MIN_ROWS", 1000)
    monkeypatch.setattr(arrow_io, "HOST_THREADS", 5)
    rng = np.random.default_rng(5)
    n = 12_345
    values = np.array(["k%02d" % i for i in range(40)] + [None], dtype=object)
    picks = rng.integers(0, 41, n)
    picks[: n // 2] = np.minimum(picks[: n // 2], 20)    # the upper keys appear in the later slices only
    picks[-7] = 39
    arr = pa.array(values[picks], type=pa.string())
    for mode in ("encode", "mask"):
        dictionary, indices = arrow_io.dictionary_encode(arr, null_encoding=mode)
        assert isinstance(indices, pa.ChunkedArray) and indices.num_chunks == 5
        want = pc.dictionary_encode(arr, null_encoding=mode)
        assert dictionary.equals(want.dictionary)
        assert indices.combine_chunks().equals(want.indices)
        back = arrow_io.dictionary_take(dictionary, indices.combine_chunks())
        assert isinstance(back, pa.ChunkedArray) and back.combine_chunks().equals(arr)
    # chunked input, and a take of a chunked column
    chunked = pa.chunked_array([arr.slice(0, 5000), arr.slice(5000)])
    dictionary, indices = arrow_io.dictionary_encode(chunked, null_encoding="encode")
    assert dictionary.equals(pc.dictionary_encode(arr, null_encoding="encode").dictionary)
    perm = pa.array(rng.permutation(n))
    assert arrow_io.dictionary_take(chunked, perm).combine_chunks().equals(arr.take(perm))
    # short columns: one call, plain arrays
    d, i = arrow_io.dictionary_encode(arr.slice(0, 500))
    assert isinstance(i, pa.Array) and isinstance(arrow_io.dictionary_take(d, i), pa.Array)
