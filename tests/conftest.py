import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def ts_us(s):
    """'YYYY-MM-DD HH:MM:SS' (UTC) -> int64 microseconds since the epoch."""
    return int(np.datetime64(s.replace(" ", "T"), "us").astype(np.int64))


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "reference_vectors.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.build()
    return oracle


# The ordering step has two forms: plain LSD passes, and -- for large tables in random order -- the
# bucketed form (partition passes over a bucket number + an in-LDS tile sort, with the plain passes
# as a device-side stand-by).  The library picks by table size; the parity tests of these modules run
# every case both ways: with the size threshold lowered so that test-sized tables take the bucketed
# form (their skewed, duplicated and bunched inputs also exercise its stand-by), and with it disabled.
_BOTH_SORT_PATHS = {"test_gpu_asof", "test_gpu_window_ops", "test_gpu_rangestats"}


def pytest_generate_tests(metafunc):
    if metafunc.module.__name__.split(".")[-1] in _BOTH_SORT_PATHS:
        metafunc.fixturenames.append("sort_path")
        metafunc.parametrize("sort_path", ["plain", "bucketed"])


@pytest.fixture(autouse=True)
def _apply_sort_path(request, monkeypatch):
    path = getattr(getattr(request.node, "callspec", None), "params", {}).get("sort_path")
    if path == "plain":
        monkeypatch.setenv("TEMPO_PLAIN_SORT", "1")
    elif path == "bucketed":
        monkeypatch.delenv("TEMPO_PLAIN_SORT", raising=False)
        monkeypatch.setenv("TEMPO_BUCKET_MIN_ROWS", "3000")
    yield
