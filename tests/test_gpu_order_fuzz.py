"""Randomised tables through tmp_sort_order against a host sort: every shortcut the ordering step
knows (rows in order, time-ordered feed, a few time-ordered runs, every key in order, bucketed
sort of large random tables, plain passes) on sizes either side of the thresholds that select
them, with and without a sequence column and the seconds cast."""
import zlib

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

BASE = 1_596_240_000_000_000


def _table(rng, n, n_keys, shape, step_us, span_steps):
    key = rng.integers(0, n_keys, n).astype(np.int32)
    ts = BASE + rng.integers(0, span_steps, n) * step_us
    if shape == "sorted":
        o = np.lexsort((ts, key)); key, ts = key[o], ts[o]
    elif shape == "time":
        ts = np.sort(ts)
    elif shape == "runs":
        k = int(rng.integers(2, 20))
        cuts = np.r_[0, np.sort(rng.choice(np.arange(1, n), k - 1, replace=False)), n]
        for a, b in zip(cuts[:-1], cuts[1:]):
            ts[a:b] = np.sort(ts[a:b])
    elif shape in ("per_key", "almost_per_key"):
        # every key's rows in time order, the keys interleaved at random
        where = np.argsort(key, kind="stable")          # positions of key 0's rows, then key 1's, ...
        ts_new = np.empty_like(ts)
        ts_new[where] = ts[np.lexsort((ts, key))]
        ts = ts_new
        if shape == "almost_per_key":
            ts[rng.choice(n, 3, replace=False)] = BASE
    return key, ts


CASES = []
_rng = np.random.default_rng(2024)
for shape in ("random", "sorted", "time", "runs", "per_key", "almost_per_key"):
    for n in (int(_rng.integers(2_000, 60_000)), int(_rng.integers(1_050_000, 1_400_000))):
        CASES.append((shape, n, int(_rng.choice([1, 3, 40, 5000])), bool(_rng.integers(0, 2)), bool(_rng.integers(0, 2)),
                      int(_rng.choice([1, 1000, 1_000_000]))))


@pytest.mark.parametrize("shape,n,n_keys,with_seq,floor_seconds,step_us", CASES)
def test_order_fuzz(orc, monkeypatch, shape, n, n_keys, with_seq, floor_seconds, step_us):
    from tempo_amd import ops
    monkeypatch.delenv("TEMPO_PLAIN_SORT", raising=False)
    rng = np.random.default_rng(zlib.crc32(repr((shape, n, n_keys)).encode()))
    span_steps = int(rng.choice([50, 5_000, 3_000_000]))
    key, ts = _table(rng, n, n_keys, shape, step_us, span_steps)
    seq = sv = None
    if with_seq:
        seq = rng.integers(-5, 50, n)
        sv = (rng.random(n) > 0.2).astype(np.uint8)
    d = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()
    perm = ops.sort_order(d(key), d(ts), n_keys, floor_seconds=floor_seconds, seq=d(seq), seq_valid=d(sv))
    prim = ts // 1_000_000 if floor_seconds else ts
    sfield = np.zeros(n, np.int64) if seq is None else np.where(sv != 0, seq - seq.min() + 1, 0)
    want = np.lexsort((np.arange(n), sfield, prim, key))
    assert np.array_equal(perm.cpu().numpy().astype(np.int64), want)


@pytest.mark.parametrize("n,n_keys,shape", [(5_000, 2, "time"), (70_001, 100, "time"), (1_200_000, 300, "time"),
                                            (300_000, 200, "per_key"), (1_100_000, 9, "per_key"),
                                            (8_193, 256, "time"), (16_384, 1, "time"),
                                            # 2^20 rows and more, at most 256 keys: the range scan leaves the
                                            # key histogram behind, the pass runs no histogram kernel of its own
                                            (1_300_003, 100, "time"), (1_048_576, 2, "time")])
def test_time_ordered_first_pass_builds_its_words(orc, monkeypatch, n, n_keys, shape):
    """A time-ordered (or per-key-ordered) feed takes one pass over the key bits whose scatter builds
    the sort words itself from the key / time / validity columns: the order, the ordered metric and the
    statistics are those of the pass that reads words laid out beforehand, and the oracle's."""
    from tempo_amd import _lib, ops
    rng = np.random.default_rng(n + n_keys)
    key, ts = _table(rng, n, n_keys, shape, 250_000, 4 * n)
    x = np.round(rng.normal(100, 3, n), 2)
    xv = (rng.random(n) > 0.1).astype(np.uint8)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    runs = {}
    for fused in (True, False):
        if fused:
            monkeypatch.delenv("TEMPO_NO_FUSED_WORDS", raising=False)
        else:
            monkeypatch.setenv("TEMPO_NO_FUSED_WORDS", "1")
        _lib.profile_enable(True)
        _lib.profile_reset()
        perm = ops.sort_order(d(key), d(ts), n_keys, floor_seconds=True)
        st = ops.range_stats(d(key), d(ts), d(x), d(xv), n_keys, 30)
        torch.cuda.synchronize()
        names = {name for name, _, _ in _lib.profile_read()}
        _lib.profile_enable(False)
        if n_keys > 1 and (shape == "time" or n >= 1 << 20):  # the path under test really ran (or really did not)
            assert fused == any("scatter_kernel+words" in name for name in names), names
        runs[fused] = (perm.cpu().numpy(), {k: v.cpu().numpy() for k, v in st.items() if isinstance(v, torch.Tensor)})
    assert np.array_equal(runs[True][0], runs[False][0])
    assert np.array_equal(runs[True][0], np.lexsort((np.arange(n), ts // 1_000_000, key)))
    for name, a in runs[True][1].items():
        b = runs[False][1][name]
        assert np.array_equal(a, b, equal_nan=a.dtype.kind == "f"), name
    e = orc.range_stats(key, ts, x, xv, 30)
    assert np.array_equal(runs[True][1]["perm"].astype(np.int64), e["perm"])
    assert np.array_equal(runs[True][1]["count"], e["count"])
    ok = e["stat_valid"].astype(bool)
    assert np.array_equal(runs[True][1]["min"][ok], e["min"][ok])
    np.testing.assert_allclose(runs[True][1]["mean"][ok], e["mean"][ok], rtol=1e-9)
