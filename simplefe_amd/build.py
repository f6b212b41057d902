"""Build libsfe_dsp.so (the HIP extension) in-tree with hipcc for gfx950.

    python -m simplefe_amd.build          # rebuild if sources are newer than the .so

hipcc cross-compiles without a GPU; the built .so sits next to this file so that it travels
with the repo snapshot to the GPU box and is the library the tests are seen to load.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsfe_dsp.so")
# diagnostic flavour (-DSFE_DIAG): A/B kernel variants, bare access-pattern kernels, load/store
# suppression switches, all behind environment variables.  Used by scripts/ only (ab_fir.py,
# ablate.py); never loaded by simplefe_amd.lib, tests or bench.py, and not built by default.
LIB_DIAG = os.path.join(HERE, "libsfe_dsp_diag.so")
ARCH = "gfx950"
# bit-exact restatements of the reference arithmetic: no implicit FMA contraction
EXACT_SOURCES = ("polyphase.hip", "util.hip")
TICKET_SOURCES = ("fir_fft.hip", "poly_fft.hip")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _deps():
    return sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(
        os.path.join(os.path.dirname(HERE), "include", "*.h"))


def csrc_hash():
    """sha256 over the kernel sources (csrc/*.hip, csrc/*.h: names and contents).  Stamped into
    profiles/pmc_*.json when counters are collected; bench.py reports roofline.traffic only while
    the stamp matches the tree (otherwise traffic: null, traffic_stale: true)."""
    import hashlib
    h = hashlib.sha256()
    for p in sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h"))):
        h.update(os.path.basename(p).encode() + b"\0")
        h.update(open(p, "rb").read())
        h.update(b"\0")
    return h.hexdigest()


def needs_build(lib=LIB):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(p) > t for p in _deps())


def build_lib(force=False, verbose=False, extra=(), diag=False):
    LIB = LIB_DIAG if diag else globals()["LIB"]
    if diag:
        extra = (*extra, "-DSFE_DIAG")
    if not force and not needs_build(LIB):
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    odir = os.path.join(HERE, "build_diag" if diag else "build")
    os.makedirs(odir, exist_ok=True)
    procs = []
    for src in sources():
        obj = os.path.join(odir, os.path.basename(src) + ".o")
        objs.append(obj)
        if (not force and os.path.exists(obj) and
                os.path.getmtime(obj) > max(os.path.getmtime(p) for p in [src] + glob.glob(os.path.join(CSRC, "*.h"))
                                            + glob.glob(os.path.join(os.path.dirname(HERE), "include", "*.h")))):
            continue
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", obj,
               "-Wall", "-Wno-unused-function", *extra]
        if os.path.basename(src) in EXACT_SOURCES:
            cmd.append("-ffp-contract=off")
        if os.path.basename(src) in TICKET_SOURCES:
            # the work-counter draw is issued early and looked at late (fir_fft.hip: draw_issue / draw_finish);
            # the wave-aggregated lowering of the atomic optimizer would read its result back on the spot
            cmd += ["-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB, *objs]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    if "--hash" in sys.argv:
        print(csrc_hash())
        raise SystemExit(0)
    print(build_lib(force="--force" in sys.argv, verbose=True, diag="--diag" in sys.argv))
