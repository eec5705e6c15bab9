import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as d:
        return {k: d[k] for k in d.files}


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (checker only)."""
    from oracle import oracle as orc
    orc.build()
    return orc


def assert_llr_close(a, b, tol=1e-5):
    """LLR comparison of the north-star contract: |a-b| <= tol, equal infinities equal, NaN==NaN."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape
    fin = np.isfinite(a) & np.isfinite(b)
    assert np.array_equal(np.isnan(a), np.isnan(b))
    inf = np.isinf(a) | np.isinf(b)
    assert np.array_equal(a[inf], b[inf])
    if fin.any():
        assert np.max(np.abs(a[fin] - b[fin])) <= tol
