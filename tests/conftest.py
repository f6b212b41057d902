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
def g7():
    """Outputs of the reference's blkconv class on the reference's OWN FFTW 3.3.5 binary
    (tests/golden/make_golden_fftw.py; oracle/pe/): the pin at the FFTW boundary."""
    return np.load(os.path.join(GOLDEN, "g7_blkconv_fftw.npz"))


G7_CASES = ["kat", "bpsk", "cfg1", "cfg2", "rrc551"]


@pytest.fixture(scope="session")
def orc():
    from oracle import binding
    binding.lib()
    return binding


def has_gpu():
    return os.path.exists("/dev/kfd")
