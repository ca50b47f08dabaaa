import sys

import numpy as np
import os

import pytest

# the XCD-resident chain kernel is used for batches >= 192 by default; the suites run it at every batch size
os.environ.setdefault("SBM_CHAIN_MIN_B", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name)))
        return cache[name]

    return load


def assert_close(got, want, rtol, atol, what=""):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, f"{what}: shape {got.shape} vs {want.shape}"
    err = np.abs(got - want)
    bound = atol + rtol * np.abs(want)
    bad = err > bound
    assert not bad.any(), (f"{what}: {int(bad.sum())}/{bad.size} outside rtol={rtol} atol={atol}; "
                           f"max abs err {err.max():.3e} at {np.unravel_index(err.argmax(), err.shape)}, "
                           f"max |want| {np.abs(want).max():.3e}")
