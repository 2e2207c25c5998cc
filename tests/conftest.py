import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    denom = max(float(np.max(np.abs(b))) if b.size else 0.0, 1e-300)
    return float(np.max(np.abs(a - b))) / denom if a.size else 0.0


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture
def deterministic(monkeypatch):
    """Contexts created inside the test use the fixed-order vertex sums (ms_set_deterministic):
    what the bitwise-equality tests need.  Everything else runs the default LDS-atomic mode."""
    monkeypatch.setenv("MS_DETERMINISTIC", "1")
