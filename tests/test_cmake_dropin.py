"""The committed CMake drop-in (CMakeLists.txt at the repo root): a scratch project whose ONLY
source is the reference's own libdsp/test/test_blkconv.cxx (used in place, never copied) and whose
only link line is the reference's -- `target_link_libraries(test_blkconv LINK_PUBLIC Libdsp)`
(libdsp/test/CMakeLists.txt:1-2) -- configures and builds against this repo's `Libdsp` target
(libdsp/CMakeLists.txt:15-21: one library, one public include dir).  The build tree lives under
oracle/_ref/ (git-ignored, travels to the GPU box), where tests/test_gpu_dropin.py runs the binary."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_TEST = "/root/reference/libdsp/test/test_blkconv.cxx"
WORK = os.path.join(ROOT, "oracle", "_ref", "cmake_dropin")

PROJECT = """cmake_minimum_required(VERSION 3.16)
project(dropin_check CXX)
add_subdirectory({root} sfe_dsp)
add_executable(test_blkconv {src})
target_link_libraries(test_blkconv LINK_PUBLIC Libdsp)
"""


@pytest.mark.skipif(not os.path.exists(REF_TEST), reason="reference tree absent")
@pytest.mark.skipif(shutil.which("cmake") is None or not os.path.exists("/opt/rocm/bin/hipcc"), reason="cmake or hipcc absent")
def test_cmake_libdsp_target_builds_the_reference_test_program():
    proj, build = os.path.join(WORK, "proj"), os.path.join(WORK, "build")
    os.makedirs(proj, exist_ok=True)
    with open(os.path.join(proj, "CMakeLists.txt"), "w") as f:
        f.write(PROJECT.format(root=ROOT, src=REF_TEST))
    gen = ["-G", "Ninja"] if shutil.which("ninja") else []
    r = subprocess.run(["cmake", "-S", proj, "-B", build, *gen], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run(["cmake", "--build", build, "-j", "8"], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-4000:]
    exe, lib = os.path.join(build, "test_blkconv"), os.path.join(build, "sfe_dsp", "libsfe_dsp.so")
    assert os.path.exists(exe) and os.path.exists(lib)
    # the program links the C-ABI library and nothing of FFTW; the library CMake built exports the whole C ABI
    dyn = subprocess.run(["readelf", "-d", exe], capture_output=True, text=True).stdout
    assert "libsfe_dsp.so" in dyn and "fftw" not in dyn.lower()
    hdr = open(os.path.join(ROOT, "include", "sfe_dsp.h")).read()
    declared = set(re.findall(r"\b(sfe_dsp_[a-z0-9_]+)\s*\(", hdr))
    exported = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True).stdout
    missing = [s for s in sorted(declared) if not re.search(r"\bT %s\b" % s, exported)]
    assert not missing, missing
    # without a GPU the program must fail loudly (no CPU fallback), with one it prints the known answer
    if not os.path.exists("/dev/kfd"):
        r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        assert r.returncode != 0 and "blksize" not in r.stdout
