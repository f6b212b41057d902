"""oracle/pe/: the reference's blkconv.cxx on the reference's own FFTW 3.3.5 binary.

These tests need /root/reference (the DLL is read where it lies), so they run in the authoring
container and skip on the GPU box; what they protect -- tests/golden/g7_blkconv_fftw.npz -- is
checked everywhere by tests/test_oracle.py and the `-m gpu` parity tests.
"""
import os
import re
import subprocess

import numpy as np
import pytest

from simplefe_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DLL = "/root/reference/contrib/fftw-3.3.5-dll64/libfftw3f-3.dll"
needs_reference = pytest.mark.skipif(not os.path.exists(DLL), reason="the vendored FFTW DLL lives under /root/reference")


@pytest.fixture(scope="module")
def built():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref_fftw"])
    return os.path.join(ROOT, "oracle", "_ref")


@needs_reference
def test_reference_test_program_on_reference_fftw(built):
    """libdsp/test/test_blkconv.cxx + libdsp/blkconv.cxx + libfftw3f-3.dll, all unmodified: prints
    blksize 28, then 1 2 3 4 5 5 ... 5, then 4 3 2 1 0 ... 0."""
    out = subprocess.run([os.path.join(built, "test_blkconv_fftw")], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.split()
    assert lines[:3] == ["blksize", "=", "28"]
    vals = [float(v) for v in lines[3:]]
    assert vals == [1, 2, 3, 4] + [5] * 24 + [4, 3, 2, 1] + [0] * 24


@needs_reference
def test_loader_and_stubs_are_clean_under_asan_and_ubsan():
    """The same program with peload.c / win_stubs.c / fftwf_tramp.c built -fsanitize=address,undefined: mapping, relocating,
    binding, the start-up code's calls into the stubs and the six forwarded FFTW calls raise no report."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref_fftw_asan"])
    out = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "test_blkconv_fftw_asan")], capture_output=True, text=True, timeout=120,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="halt_on_error=1"))
    assert out.returncode == 0, out.stderr[-2000:]
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr[-2000:]
    vals = [float(v) for v in out.stdout.split()[3:]]
    assert vals == [1, 2, 3, 4] + [5] * 24 + [4, 3, 2, 1] + [0] * 24


@needs_reference
def test_live_reference_reproduces_the_committed_fixture(built, orc, g7):
    """The fixture is what the reference computes here, today: every case regenerated and
    compared.  Bit-equal on the CPU that wrote it (FFTW chooses codelets by CPU features, so another
    x86-64 may differ in rounding: the hard bound is 1e-6)."""
    assert orc.RefBlkconvFFTW.fftw_version().startswith("fftw-3.3.5")
    from tests.conftest import G7_CASES
    for name in G7_CASES:
        taps, fft_len, x, want = g7[f"{name}_taps"], int(g7[f"{name}_fft_len"]), g7[f"{name}_x"], g7[f"{name}_y"]
        live = orc.RefBlkconv.stream(orc.RefBlkconvFFTW(taps, fft_len), x)        # block by block
        assert synth.rel_rms(live, want) < 1e-6, name
        bulk = orc.RefBlkconvFFTW(taps, fft_len).stream(x)
        assert np.array_equal(live, bulk), name


@needs_reference
def test_loader_and_stubs_hold_no_floating_point_arithmetic(built):
    """'the stubs contain no floating-point code'.  Disassemble the three objects: SSE
    register moves are allowed (ms_abi prologues save xmm6-15, memcpy is inlined through xmm), any
    arithmetic, conversion, comparison or x87 instruction is not."""
    arith = re.compile(r"\b(v?(add|sub|mul|div|sqrt|max|min|rcp|rsqrt|cmp|comi|ucomi|cvt\w*|fmadd\w*|fmsub\w*|fnmadd\w*|hadd|dp)"
                       r"(ss|sd|ps|pd)\w*|f(ld|st|add|sub|mul|div|sin|cos|sqrt|ild|ist)\w*)\b")
    objs = sorted(os.listdir(os.path.join(built, "pe_obj")))
    assert objs == ["fftwf_tramp.o", "peload.o", "win_stubs.o"]
    for o in objs:
        dis = subprocess.run(["objdump", "-d", "--no-show-raw-insn", os.path.join(built, "pe_obj", o)],
                             capture_output=True, text=True, check=True).stdout
        hits = [l for l in dis.splitlines() if arith.search(l.split(":", 1)[-1])]
        assert not hits, (o, hits[:5])
    for src in ("peload.c", "win_stubs.c", "fftwf_tramp.c"):
        text = open(os.path.join(ROOT, "oracle", "pe", src)).read()
        code = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        code = re.sub(r"\b(float \*|const float \*)", "", code)      # pointers are passed through, never read
        assert not re.search(r"\b(float|double)\b", code), src


@needs_reference
def test_every_import_of_the_dll_has_a_stub(built):
    """objdump -p lists 29 KERNEL32 + 31 msvcrt imports; the loader refuses to run with any unbound."""
    p = subprocess.run(["objdump", "-p", DLL], capture_output=True, text=True, check=True).stdout
    sec = p.split("The Import Tables", 1)[1].split("The Export Tables", 1)[0]
    names = re.findall(r"^\s+[0-9a-f]+\s+\d+\s+(\w+)\s*$", sec, flags=re.M)
    assert len(names) == 60
    src = open(os.path.join(ROOT, "oracle", "pe", "win_stubs.c")).read()
    table = src.split("g_stubs[] = {", 1)[1]
    bound = set(re.findall(r"[KC]\((\w+)\)", table))
    assert set(names) == bound
