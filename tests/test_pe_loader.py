"""oracle/pe/: the reference's blkconv.cxx on the reference's own FFTW 3.3.5 binary.

These tests need /root/reference (the DLL is read where it lies) AND an explicit opt-in -- they execute a binary from the
reference tree --

    SFE_ORACLE_RUN_FFTW_DLL=1 python -m pytest tests/test_pe_loader.py -q

so they skip in the default CPU suite and on the GPU box; what they protect -- tests/golden/g7_blkconv_fftw.npz -- is checked
everywhere by tests/test_oracle.py and the `-m gpu` parity tests, and the two checks that execute nothing of the DLL (the stubs
hold no arithmetic; every import has a stub) run whenever /root/reference is there.  The DLL is mapped only in child processes
(the reference's own test program; oracle/ref_fftw_child.py), only if its SHA-256 is the pinned one (oracle/pe/peload.c), and the
last run's output is kept in profiles/r05/pe_loader_opt_in.txt.
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
OPT_IN = dict(os.environ, SFE_ORACLE_RUN_FFTW_DLL="1")
runs_the_dll = pytest.mark.skipif(os.environ.get("SFE_ORACLE_RUN_FFTW_DLL") != "1" or not os.path.exists(DLL),
                                  reason="executes a binary from the reference tree: opt in with SFE_ORACLE_RUN_FFTW_DLL=1")


@pytest.fixture(scope="module")
def built():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref_fftw"])
    return os.path.join(ROOT, "oracle", "_ref")


@runs_the_dll
def test_reference_test_program_on_reference_fftw(built):
    """libdsp/test/test_blkconv.cxx + libdsp/blkconv.cxx + libfftw3f-3.dll, all unmodified: prints
    blksize 28, then 1 2 3 4 5 5 ... 5, then 4 3 2 1 0 ... 0."""
    out = subprocess.run([os.path.join(built, "test_blkconv_fftw")], capture_output=True, text=True, timeout=60, env=OPT_IN)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.split()
    assert lines[:3] == ["blksize", "=", "28"]
    vals = [float(v) for v in lines[3:]]
    assert vals == [1, 2, 3, 4] + [5] * 24 + [4, 3, 2, 1] + [0] * 24


@needs_reference
def test_loader_refuses_without_the_opt_in_and_with_another_file(built, tmp_path):
    """No SFE_ORACLE_RUN_FFTW_DLL=1: nothing is mapped, the program aborts with the reason.  With the opt-in but a file whose
    SHA-256 is not the pinned one (the DLL with one byte changed): refused as well.  Neither run executes a byte of the DLL."""
    env = {k: v for k, v in os.environ.items() if k != "SFE_ORACLE_RUN_FFTW_DLL"}
    out = subprocess.run([os.path.join(built, "test_blkconv_fftw")], capture_output=True, text=True, timeout=60, env=env)
    assert out.returncode != 0 and "SFE_ORACLE_RUN_FFTW_DLL=1" in out.stderr
    blob = bytearray(open(DLL, "rb").read())
    blob[len(blob) // 2] ^= 1
    other = tmp_path / "other.dll"
    other.write_bytes(bytes(blob))
    out = subprocess.run([os.path.join(built, "test_blkconv_fftw")], capture_output=True, text=True, timeout=60,
                         env=dict(OPT_IN, SFE_FFTW_DLL=str(other)))
    assert out.returncode != 0 and "not the pinned library" in out.stderr


@runs_the_dll
def test_loader_and_stubs_are_clean_under_asan_and_ubsan():
    """The same program with peload.c / win_stubs.c / fftwf_tramp.c built -fsanitize=address,undefined: mapping, relocating,
    binding, the start-up code's calls into the stubs and the six forwarded FFTW calls raise no report."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref_fftw_asan"])
    out = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "test_blkconv_fftw_asan")], capture_output=True, text=True, timeout=120,
                         env=dict(OPT_IN, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="halt_on_error=1"))
    assert out.returncode == 0, out.stderr[-2000:]
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr[-2000:]
    vals = [float(v) for v in out.stdout.split()[3:]]
    assert vals == [1, 2, 3, 4] + [5] * 24 + [4, 3, 2, 1] + [0] * 24


@runs_the_dll
def test_live_reference_reproduces_the_committed_fixture(built, orc, g7):
    """The fixture is what the reference computes here, today: every case regenerated and
    compared.  Bit-equal on the CPU that wrote it (FFTW chooses codelets by CPU features, so another
    x86-64 may differ in rounding: the hard bound is 1e-6)."""
    assert orc.RefBlkconvFFTW.fftw_version().startswith("fftw-3.3.5")
    from tests.conftest import G7_CASES
    for name in G7_CASES:
        taps, fft_len, x, want = g7[f"{name}_taps"], int(g7[f"{name}_fft_len"]), g7[f"{name}_x"], g7[f"{name}_y"]
        live = orc.RefBlkconvFFTW(taps, fft_len).stream_blocks(x)                 # block by block, in a child process
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
