"""The cell paths against the ordered paths of the same library on random tables (tools/fuzz_cells.py):
resample mean / min / max / floor / ceil, withGroupedStats and withRangeStats(any_order) must agree
whichever way the library computes them -- keys, starts, counts, extremes and picked rows bit for
bit, means / sums / deviations within 1e-9."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.gpu
def test_cell_paths_agree_with_ordered_paths(monkeypatch):
    import fuzz_cells
    for k in fuzz_cells.OFF + ["TEMPO_DENSE_MIN_ROWS"]:
        monkeypatch.delenv(k, raising=False)
    fuzz_cells.TAKEN.clear()
    try:
        for seed in range(1000, 1030):
            fuzz_cells.one_case(seed)
    finally:
        for k in fuzz_cells.OFF + ["TEMPO_DENSE_MIN_ROWS"]:
            os.environ.pop(k, None)
    taken = fuzz_cells.TAKEN
    for kernel in ("dense_mean_kernel", "dense_stats_kernel<0>", "dense_pick_kernel<0>", "cell_rows_kernel"):
        assert taken.get(kernel, 0) > 0, (kernel, taken)


@pytest.mark.gpu
def test_cell_paths_twice_at_their_default_thresholds(orc):
    """The cell paths add with floating-point atomics, so the last bits of their sums depend on the
    order the hardware took the adds in.  At sizes that take each path by default (no threshold
    lowered): two runs agree exactly on everything that is not a sum (bucket membership, counts,
    minima, maxima, NULL flags), to 1e-12 on means / sums / standard deviations, and the result is
    within 1e-9 of the oracle's."""
    import numpy as np
    import torch
    from tempo_amd import ops
    rng = np.random.default_rng(8)
    base = 1_596_240_000_000_000
    d = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()

    # resample(mean) and withGroupedStats over (key, minute) cells: 2^21 rows in time order, 16 keys
    n, nk = 1 << 21, 16
    key = rng.integers(0, nk, n).astype(np.int32)
    ts = base + np.sort(rng.integers(0, 4 * 3600 * 1_000_000, n))
    x = 100 + rng.normal(0, 1, n)
    xv = (rng.random(n) >= 0.05).astype(np.uint8)
    runs = [ops.resample(d(key), d(ts), nk, 60_000_000, "mean", [d(x)], [d(xv)]) for _ in range(2)]
    a, b = runs
    assert a["n"] == b["n"] and torch.equal(a["key"], b["key"]) and torch.equal(a["start"], b["start"])
    assert torch.equal(a["valid"], b["valid"])
    np.testing.assert_allclose(a["f64"].cpu().numpy(), b["f64"].cpu().numpy(), rtol=1e-12)
    ref = orc.resample(key, ts, 60_000_000, "mean", [x], [xv])
    assert a["n"] == ref["n"] and np.array_equal(a["key"].cpu().numpy(), ref["key"])
    np.testing.assert_allclose(a["f64"].cpu().numpy()[0], ref["f64"][0], rtol=1e-9)
    g = [ops.grouped_stats(d(key), d(ts), nk, 60_000_000, [d(x)], [d(xv)]) for _ in range(2)]
    for name in ("count", "min", "max"):
        assert torch.equal(g[0][name][0], g[1][name][0]), name
    for name in ("mean", "sum", "stddev"):
        np.testing.assert_allclose(g[0][name][0].cpu().numpy(), g[1][name][0].cpu().numpy(), rtol=1e-12, atol=1e-12)
    rg = orc.grouped_stats(key, ts, 60_000_000, [x], [xv])
    assert np.array_equal(g[0]["count"][0].cpu().numpy(), rg["count"][0])
    np.testing.assert_allclose(g[0]["stddev"][0].cpu().numpy(), rg["stddev"][0], rtol=1e-9)
    np.testing.assert_allclose(g[0]["mean"][0].cpu().numpy(), rg["mean"][0], rtol=1e-9)

    # withRangeStats(any_order) over (key, second) cells: 2^22 rows, 50 rows per second and key
    n, nk = 1 << 22, 8
    key = rng.integers(0, nk, n).astype(np.int32)
    ts = base + np.sort(rng.integers(0, (n // (50 * nk)) * 1_000_000, n))
    x = 50 + rng.normal(0, 1, n)
    xv = (rng.random(n) >= 0.02).astype(np.uint8)
    r = [ops.range_stats(d(key), d(ts), d(x), d(xv), nk, 30, any_order=True) for _ in range(2)]
    assert torch.equal(r[0]["perm"], torch.arange(n, dtype=torch.int32, device="cuda")), "the cell path was not taken"
    for name in ("perm", "count", "min", "max", "valid"):
        assert torch.equal(r[0][name], r[1][name]), name
    for name in ("mean", "sum", "stddev"):
        np.testing.assert_allclose(r[0][name].cpu().numpy(), r[1][name].cpu().numpy(), rtol=1e-12, atol=1e-12)
    ref = orc.range_stats(key, ts, x, xv, 30)
    inv = np.empty(n, np.int64)
    inv[ref["perm"]] = np.arange(n)    # oracle rows are in (key, second) order: back to input order
    assert np.array_equal(r[0]["count"].cpu().numpy(), ref["count"][inv])
    assert np.array_equal(r[0]["min"].cpu().numpy(), ref["min"][inv])
    np.testing.assert_allclose(r[0]["mean"].cpu().numpy(), ref["mean"][inv], rtol=1e-9)
    np.testing.assert_allclose(r[0]["stddev"].cpu().numpy(), ref["stddev"][inv], rtol=1e-9, atol=1e-12)
