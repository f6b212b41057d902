"""The drop-in C++ class surface on the GPU: the reference's own test program and a caller
written like the reference's drivers, both linked against libsfe_dsp.so.  `-m gpu`."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "simplefe_amd")


@pytest.fixture(scope="module")
def dropin_exe(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("dropin") / "test_dropin")
    r = subprocess.run(["g++", "-O1", os.path.join(ROOT, "tests/host/test_dropin.cpp"), "-o", exe,
                        "-L" + LIBDIR, "-lsfe_dsp", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib",
                        "-Wl,--allow-shlib-undefined"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_reference_test_blkconv_program_unmodified(g1):
    """libdsp/test/test_blkconv.cxx compiled from the reference tree against include/blkconv.h
    (oracle/Makefile `dropin`), run here: it prints blksize and 2 x 28 values with %.2f."""
    exe = os.path.join(ROOT, "oracle/_ref/test_blkconv_dropin")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/test_blkconv_dropin not prebuilt")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.split("\n")
    assert lines[0].strip() == "blksize = 28"
    vals = [l.strip() for l in lines[1:] if l.strip()]
    want = ["%.2f" % v for v in list(g1["out1"]) + list(g1["out2"])]
    assert [v.lstrip("-") for v in vals] == want       # "-0.00" and "0.00" print alike in effect


def test_cxx_blkconv_class(dropin_exe, g1):
    r = subprocess.run([dropin_exe, "blkconv"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.split()
    assert lines[:3] == ["blksize", "=", "28"]
    v = np.array([float(t) for t in lines[3:]], dtype=np.float64)
    assert np.allclose(v[:28], g1["out1"], atol=2e-6) and np.allclose(v[28:], g1["out2"], atol=2e-6)


@pytest.mark.parametrize("cls", ["resample", "decimate"])
def test_cxx_resampler_classes_bit_exact(dropin_exe, g4, cls, tmp_path):
    """The C++ classes through the driver loop of test_decimate.py:22-25, rate 1.77."""
    g4["taps"].astype(np.float32).tofile(tmp_path / "taps.f32")
    g4["x"].astype(np.float32).tofile(tmp_path / "x.f32")
    B, U = int(g4["B"]), int(g4["U"])
    r = subprocess.run([dropin_exe, "rs", cls, str(tmp_path / "taps.f32"), str(tmp_path / "x.f32"), str(U), str(B),
                        str(4 * B), repr(float(g4["rate_1p77"])), str(tmp_path / "y.f32")],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert [int(t) for t in r.stdout.split()] == g4["n_1p77"].tolist()
    assert np.array_equal(np.fromfile(tmp_path / "y.f32", dtype=np.float32), g4["y_1p77"])
