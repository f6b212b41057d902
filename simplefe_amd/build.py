"""Build libsfe_dsp.so (the HIP extension) in-tree with hipcc for gfx950.

    python -m simplefe_amd.build          # rebuild if sources are newer than the .so

hipcc cross-compiles without a GPU; the built .so sits next to this file so that it travels
with the repo snapshot to the GPU box and is the library the tests are seen to load.
"""
import glob
import json
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
# host side only (handles, plans, launch choices, device groups):
# not part of the kernel-source hash
HOST_SOURCES = ("api.hip", "api_plans.hip", "api_fir.hip", "api_rs.hip", "api_pipe.hip", "group.hip", "host.h")


def sources(diag=False):
    # the diagnostic flavour also holds whole kernels that were measured and not kept (csrc/diag/*.hip)
    return sorted(glob.glob(os.path.join(CSRC, "*.hip"))) + (sorted(glob.glob(os.path.join(CSRC, "diag", "*.hip"))) if diag else [])


def _deps():
    return sources(diag=True) + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "diag", "*.inc")) + glob.glob(
        os.path.join(os.path.dirname(HERE), "include", "*.h"))


# the files that define one workload's kernel (round 5): a counter pass is stamped with the hash of THESE, so that a change to another
# kernel's file does not orphan it.  None = every kernel source.
KERNEL_FILES = {"fir": ("fir_fft.hip", "fft16.h", "common.h"), "resample": ("poly_fft.hip", "fft16.h", "common.h"),
                "decimate": ("polyphase.hip", "common.h")}


def csrc_hash(kind=None):
    """sha256 over the KERNEL sources -- csrc/*.hip and csrc/*.h except HOST_SOURCES (the host side: handles, plans, launch
    choices; what the counters count is the kernels' traffic), or, with `kind` in KERNEL_FILES, the files of that workload's
    kernel only.  File names and CODE: `//` comments, blank lines and indentation are left out, and so is everything between
    `#ifdef SFE_DIAG` and its `#else` / `#endif` (the diagnostic library's switches: not in the product kernels), so that
    rewording a comment or adding a diagnostic variant does not orphan a counter pass.
    Stamped into profiles/pmc_*.json when counters are collected; bench.py reports roofline.traffic
    only while the stamp matches the tree (otherwise traffic: null, traffic_stale: true)."""
    import hashlib
    import re
    h = hashlib.sha256()
    paths = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")))
    if kind is not None:
        paths = [p for p in paths if os.path.basename(p) in KERNEL_FILES[kind]]
    for p in paths:
        if os.path.basename(p) in HOST_SOURCES:
            continue
        h.update(os.path.basename(p).encode() + b"\0")
        depth, skip_at = 0, None          # preprocessor nesting; the depth at which a diagnostic block opened
        for line in open(p, errors="replace"):
            code = re.sub(r"\s+", " ", re.sub(r"//.*$", "", line)).strip()
            if code.startswith("#if"):
                depth += 1
                if skip_at is None and re.match(r"#ifdef SFE_DIAG\b", code):
                    skip_at = depth
                    continue
            elif code.startswith("#else") and skip_at == depth:
                skip_at = -depth           # the product's branch of a diagnostic conditional: hashed
                continue
            elif code.startswith("#endif"):
                closing = depth
                depth -= 1
                if skip_at is not None and abs(skip_at) == closing:
                    skip_at = None
                    continue
            if skip_at is not None and skip_at > 0:
                continue
            if code:
                h.update(code.encode() + b"\n")
        h.update(b"\0")
    return h.hexdigest()


def needs_build(lib=LIB):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(p) > t for p in _deps())


def parse_resources(text):
    """hipcc -Rpass-analysis=kernel-resource-usage remarks -> {demangled kernel name: {field: value}}."""
    import re
    out, cur = {}, None
    for line in text.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|"
                      r"LDS Size \[bytes/block\]): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).split(" [")[0]] = int(m.group(2))
    if out:
        names = list(out)
        try:
            dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
        except OSError:         # no binutils: keep the mangled names (check_resources then sees no FIR kernel names)
            dem = []
        if len(dem) == len(names):
            out = {d: out[n] for d, n in zip(dem, names)}
    return out


FIR_TEMPLATE_ARGS = "IN_C OUT_C IN_U8 PAIR OUT_TX10 DMA DIAG ACC WP HCH".split()


def fir_kernel_flags(name):
    """Template arguments of a demangled fir_fft4096_kernel<...> name -> {flag: value} (None for other kernels)."""
    import re
    m = re.search(r"fir_fft4096_kernel<(.*?)>\(", name)
    if not m:
        return None
    vals = [a.strip() for a in m.group(1).split(",")]
    return {k: (v == "true" if v in ("true", "false") else int(v)) for k, v in zip(FIR_TEMPLATE_ARGS, vals)}


def check_resources(res):
    """the LDS-DMA FIR kernels wait with a COUNTED s_waitcnt vmcnt(15) -- the eight DMA
    pieces are older than the 15 row stores issued after them.  A scratch spill or reload between
    the request and the wait is one MORE vector-memory operation there, which only makes that wait
    retire some stores too (conservative, slower) -- it cannot make F1 read early; what would is
    FEWER than 15 operations, which the kernel's own `counted` condition rules out.  So scratch in
    these kernels is a performance bug, not a correctness one: the build refuses it for the default
    path (DMA, padded layout: the shared-filter and the per-channel kernel) and records every
    kernel's registers / scratch / occupancy in build/*.resources.json for the tests to read."""
    bad = []
    for k, r in res.get("fir_fft.hip", {}).items():
        fl = fir_kernel_flags(k)
        if fl and fl["DMA"] and not fl["WP"] and not fl["DIAG"] and (r.get("ScratchSize", 0) or r.get("VGPRs Spill", 0)):
            bad.append("%s: %s" % (k[:200], r))
        if fl and not fl["DIAG"] and r.get("Occupancy", 4) < 4:
            bad.append("%s: fewer than 4 workgroups per CU: %s" % (k[:200], r))
    if bad:
        raise RuntimeError("FIR kernels of the default path must not touch scratch and must keep 4 workgroups per CU:\n  " + "\n  ".join(bad))


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
    for src in sources(diag):
        obj = os.path.join(odir, os.path.basename(src) + ".o")
        objs.append(obj)
        if (not force and os.path.exists(obj) and
                os.path.getmtime(obj) > max(os.path.getmtime(p) for p in [src] + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "diag", "*.inc"))
                                            + glob.glob(os.path.join(os.path.dirname(HERE), "include", "*.h")))):
            continue
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", obj,
               "-Wall", "-Wno-unused-function", "-Rpass-analysis=kernel-resource-usage", *extra]
        if os.path.basename(src) in EXACT_SOURCES:
            cmd.append("-ffp-contract=off")
        if os.path.basename(src) in TICKET_SOURCES:
            # the work-counter draw is issued early and looked at late (fir_fft.hip: draw_issue / draw_finish);
            # the wave-aggregated lowering of the atomic optimizer would read its result back on the spot
            cmd += ["-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]
        if verbose:
            print(" ".join(cmd))
        log = open(obj + ".log", "w")
        procs.append((cmd, subprocess.Popen(cmd, stderr=log), log, obj))
    for cmd, p, log, obj in procs:
        rc = p.wait()
        log.close()
        text = open(obj + ".log", errors="replace").read()
        if rc != 0:
            sys.stderr.write("\n".join(l for l in text.splitlines() if "-Rpass-analysis" not in l)[-8000:])
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
        # -Wall diagnostics must not vanish into the log on a successful compile
        warn = [l for l in text.splitlines() if "warning:" in l and "-Rpass-analysis" not in l]
        if warn:
            sys.stderr.write("\n".join(warn[:40]) + "\n")
        # registers / scratch / occupancy of every kernel in the file, kept beside the object
        with open(obj[:-2] + ".resources.json", "w") as o:
            json.dump(parse_resources(text), o, indent=0, sort_keys=True)
    res = {}
    for obj in objs:
        rj = obj[:-2] + ".resources.json"
        if os.path.exists(rj):
            res[os.path.basename(obj)[:-2]] = json.load(open(rj))
    if not diag:          # the diagnostic flavour holds measured-and-rejected variants that do spill
        check_resources(res)
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB, *objs]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    if "--hash" in sys.argv:          # --hash [fir|resample|decimate]
        k = sys.argv[sys.argv.index("--hash") + 1] if len(sys.argv) > sys.argv.index("--hash") + 1 else None
        print(csrc_hash(k if k in KERNEL_FILES else None))
        raise SystemExit(0)
    print(build_lib(force="--force" in sys.argv, verbose=True, diag="--diag" in sys.argv))
