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
