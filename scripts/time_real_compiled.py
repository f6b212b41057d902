#!/usr/bin/env python3
"""The ratios that HAVE compile-time tiled kernels (polyphase.hip: poly_tiled_compiled), on a REAL float32 stream and on the complex stream of
the same bytes: the default dispatch (the compile-time kernel) and, with SFE_RT_DMA_FORCE=1 in the diagnostic library, poly_rt_dma_kernel /
poly_int4_dma_kernel forced onto the same shape.  Taps per output phase chosen so that the compile-time form applies (a multiple of 2 SP).
2^29 real / 2^28 complex samples, HIP events, median of 7 x 3 launches.   DIAG=1 [SFE_RT_DMA_FORCE=1] python scripts/time_real_compiled.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplefe_amd import build, lib  # noqa: E402
if os.environ.get("DIAG") == "1":
    lib.LIB_PATH = build.build_lib(diag=True)
from simplefe_amd import api, synth  # noqa: E402

nbytes = 1 << 31
# name, U, step, taps per output phase
SHAPES = [("rate 1 (plain FIR)", 1, 1, 32), ("decimate by 2", 1, 2, 32), ("decimate by 3", 1, 3, 36), ("decimate by 4", 1, 4, 32), ("decimate by 5", 1, 5, 30),
          ("decimate by 8", 1, 8, 32), ("decimate by 10", 1, 10, 40), ("5/3", 3, 5, 30), ("3/2", 2, 3, 36), ("5/2", 2, 5, 30), ("5/4", 4, 5, 30),
          ("2/3", 3, 2, 32), ("3/4", 4, 3, 36), ("4/3", 3, 4, 32)]
if os.environ.get("SHAPES"):
    SHAPES = [s for s in SHAPES if any(w == s[0] or w in s[0] for w in os.environ["SHAPES"].split(","))]
x = api.DeviceArray(nbytes // 4)
x.fill_synth(synth.SEED)
t = api.Timer()
warm = False
print(f"# 2^31 bytes in: 2^29 real or 2^28 complex float32 samples; SFE_RT_DMA_FORCE={os.environ.get('SFE_RT_DMA_FORCE', '0')}")
print(f"{'shape':22s} {'U':>2s} {'step':>4s} {'taps':>5s} {'real ms':>9s} {'frac':>6s} {'cplx ms':>9s} {'frac':>6s}")
for name, U, step, lp in SHAPES:
    rate = float(np.float32(step) / np.float32(U))
    taps = synth.lowpass_taps(lp * U, 0.9 * min(1.0 / U, 1.0 / step), gain=float(U))
    res = []
    for cplx in (False, True):
        w = 2 if cplx else 1
        n = nbytes // 4 // w
        cap = n * U // step + 64
        y = api.DeviceArray(w * cap)
        r = api.Rs(taps, U, 4096, mode=lib.RS_RESAMPLE, data_complex=cplx)
        for _ in range(4 if warm else 150):
            k = r.process_stream(x, n, y, cap, rate)
        warm = True
        v = []
        for _ in range(7):
            t.start()
            for _ in range(3):
                r.process_stream(x, n, y, cap, rate)
            t.stop()
            v.append(t.elapsed_ms() / 3)
        ms = float(np.median(v))
        res += [ms, 4.0 * w * (n + k) / 1e9 / ms / 8.0]
        r.close()
        y.free()
    print(f"{name:22s} {U:2d} {step:4d} {lp * U:5d} {res[0]:9.4f} {res[1]:6.3f} {res[2]:9.4f} {res[3]:6.3f}", flush=True)
