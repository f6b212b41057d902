#!/usr/bin/env python3
"""Ablation tables of the two transform kernels, regenerated from the tree:
the DIAGNOSTIC library (libsfe_dsp_diag.so, -DSFE_DIAG) can suppress a kernel's input loads,
output stores and (resample) spectrum stage; everything else -- instruction stream, barriers,
LDS traffic -- stays.  Interleaved rounds in one process on one device, medians.

    python scripts/ablate.py fir      > profiles/<tag>/fir_variants_ab.txt
    python scripts/ablate.py resample > profiles/<tag>/resample_fft_ablation.txt
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplefe_amd import build, lib  # noqa: E402
lib.LIB_PATH = build.build_lib(diag=True)
from simplefe_amd import api, synth  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "fir"
rounds = int(os.environ.get("ROUNDS", "8"))
n = 1 << 28
x = api.DeviceArray(2 * n)
x.fill_synth(synth.SEED)
t = api.Timer()


def run(cases, step, setenv):
    res = {name: [] for name, _ in cases}
    for r in range(rounds + 1):
        for name, env in cases:
            setenv(env)
            t.start()
            for _ in range(5):
                step()
            t.stop()
            if r:
                res[name].append(t.elapsed_ms() / 5)
    return {k: np.array(v) for k, v in res.items()}


if which == "fir":
    y = api.DeviceArray(2 * n)
    f = api.Fir(synth.taps_cfg2(), data_complex=True, algo=lib.FIR_ALGO_FFT)
    cases = [("product kernel (work counters + LDS-DMA, X)", ("X", "0")), ("  X: output stores suppressed", ("X", "2")),
             ("  X: input loads replaced", ("X", "1")), ("  X: neither (on-chip work only)", ("X", "3")),
             ("product kernel (work counters + register loads, T)", ("T", "0")), ("  T: output stores suppressed", ("T", "2")),
             ("  T: input loads replaced", ("T", "1")), ("  T: neither (on-chip work only)", ("T", "3")),
             ("access pattern of the LDS-DMA variant alone (g)", ("g", "0")),
             ("access pattern alone, 8 B nt lanes (e)", ("e", "0")), ("access pattern, 8 B plain (c)", ("c", "0")),
             ("access pattern, 16 B plain (d)", ("d", "0"))]

    def setenv(e):
        if e[0]:
            os.environ["SFE_FIR_VARIANT"] = e[0]
        else:
            os.environ.pop("SFE_FIR_VARIANT", None)
        os.environ["SFE_FIR_DIAG"] = e[1]
    res = run(cases, lambda: f.process_stream(x, y, n), setenv)
    print(f"# scripts/ablate.py fir   (ROUNDS={rounds} x 5 launches interleaved in one process on one MI355X; 2^28 cf32, 256 taps)")
    alg = 16.0 * n
else:
    cap = int(n * 3 / 5) + 8
    y = api.DeviceArray(2 * cap)
    r = api.Rs(synth.taps_cfg3(), 3, 4096, mode=lib.RS_RESAMPLE, data_complex=True)
    cases = [("transform-domain, full", "0"), ("no output stores", "2"), ("no input loads", "1"),
             ("no spectrum stage", "4"), ("no loads, no stores", "3"), ("no loads, stores, spectrum stage", "7")]

    def setenv(e):
        os.environ["SFE_RS_DIAG"] = e
    res = run(cases, lambda: r.process_stream(x, n, y, cap, 5.0 / 3.0), setenv)
    print(f"# scripts/ablate.py resample   (ROUNDS={rounds} x 5 launches interleaved; poly_fft256<5,3,2>, 2^28 cf32 in, 381 taps)")
    alg = 8.0 * n + 8.0 * (3 * n // 5)
for name, _ in cases:
    a = res[name]
    print(f"{name:42s} median {np.median(a):.4f} ms  min {a.min():.4f}  max {a.max():.4f}   ({alg / np.median(a) / 1e6 / 80:.1f}% of 8 TB/s if it were the kernel)")
