"""Round 5: wide sweeps over (upsample, rate, taps per phase, blksize, stream type, input format, mode) -- what SURVEY.md 8(c) calls the reference's
parameter space: resample takes any rate >= 1 / upsample, decimate any rate >= 1 (libdsp/resample.cxx:91, decimate.cxx:75-78).  Three
questions, one script each under scripts/probes/: does the bulk call REFUSE anything (it did: decimate by 128, u8 input without a fused
kernel, tap tables beyond the LDS), does the bulk call give the oracle's bits (exact mode) / stay within 1e-5 (default mode), does the
drop-in class path give the oracle's bits call for call.  `-m gpu`."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, timeout):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "probes", script)], capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


def test_the_bulk_resampler_call_refuses_nothing():
    out = _run("find_refusals.py", 600)
    assert "2016 combinations tried, 0 refused" in out, out[-3000:]


def test_the_bulk_resampler_call_matches_the_oracle_over_the_matrix():
    out = _run("sweep_bulk_path.py", 900)
    assert "1152 combinations tried, 0 bad" in out, out[-3000:]


def test_the_class_path_matches_the_oracle_over_the_matrix():
    out = _run("sweep_class_path.py", 600)
    assert "816 combinations tried, 0 bad" in out, out[-3000:]


def test_the_bulk_fir_call_matches_float64_convolution_over_the_matrix():
    """blkconv's law (libdsp/blkconv.cxx:77-110) on 1 728 combinations of (taps 1 ... 20 001, real / complex taps, real / complex data, channels,
    stream length, chunking) against float64 convolution."""
    out = _run("sweep_fir_bulk.py", 900)
    assert "1728 combinations tried, 0 bad" in out, out[-3000:]


def test_the_bulk_resampler_call_with_several_channels_in_both_modes():
    """three channels at odd strides, on and off 16-byte boundaries, resample and decimate modes, two calls with carried state: every channel against
    the oracle (432 combinations)"""
    out = _run("sweep_bulk_channels.py", 900)
    assert "432 combinations tried, 0 bad" in out, out[-3000:]
