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


@pytest.fixture(scope="session")
def golden_book():
    return np.load(os.path.join(GOLDEN, "ref_bookkeeping.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def golden_num():
    return np.load(os.path.join(GOLDEN, "ref_numpy_restatements.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def golden_analytic():
    return np.load(os.path.join(GOLDEN, "ref_analytic_density.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def golden_bethe():
    return np.load(os.path.join(GOLDEN, "ref_bethe.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def engine():
    """The process-wide GPU engine (gpu tests only)."""
    from gaunegf_amd.engine import get_engine
    return get_engine()


def rel_fro(a, b):
    a = np.asarray(a); b = np.asarray(b)
    d = np.linalg.norm((a - b).ravel())
    n = np.linalg.norm(b.ravel())
    return d / n if n > 0 else d
