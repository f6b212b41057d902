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
def g1():
    return np.load(os.path.join(GOLDEN, "g1_blkconv.npz"))


@pytest.fixture(scope="session")
def g4():
    return np.load(os.path.join(GOLDEN, "g4_reference_test_vector.npz"))


@pytest.fixture(scope="session")
def g5():
    return np.load(os.path.join(GOLDEN, "g5_baseline_shapes.npz"))


@pytest.fixture(scope="session")
def g6():
    """Outputs of the reference's own blkconv class (tests/golden/make_golden_blkconv.py)."""
    return np.load(os.path.join(GOLDEN, "g6_blkconv_reference.npz"))


@pytest.fixture(scope="session")
def orc():
    from oracle import binding
    binding.lib()
    return binding


def has_gpu():
    return os.path.exists("/dev/kfd")
