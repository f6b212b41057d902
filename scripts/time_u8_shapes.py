#!/usr/bin/env python3
"""The receive wire format (u8 offset-binary I/Q pairs, gr-simplefe/lib/source_c_impl.cc) straight into decimators and resamplers: 2^28 complex samples as
2^29 bytes in, float32 out, beside the same shape fed float32 samples (what the u8 path would cost if conversion and the narrower loads were free).
HIP events, 150 warm-up launches at the start, median of 7 x 3.  frac = algorithmic bytes of the path AS RUN (2 B or 8 B per input, 8 B per output) / time / 8 TB/s.
    python scripts/time_u8_shapes.py > profiles/r05/shapes_u8.txt"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplefe_amd import build, lib  # noqa: E402
if os.environ.get("DIAG") == "1":          # the diagnostic library: SFE_RT_DMA_U8=0 keeps wire-format input on the kernels it had before round 5's LDS-DMA path
    lib.LIB_PATH = build.build_lib(diag=True)
from simplefe_amd import api, synth  # noqa: E402

n = 1 << int(os.environ.get("LOG2N", "28"))
REAL = os.environ.get("REAL") == "1"          # real u8 streams (source_f_impl.cc: one byte per sample): 2^29 samples from the same bytes
if REAL:
    n *= 2
w = 1 if REAL else 2
SHAPES = [("decimate by 2", 1, 2, 32), ("decimate by 4", 1, 4, 32), ("decimate by 8 (64 taps)", 1, 8, 64), ("decimate by 7", 1, 7, 32), ("decimate by 16", 1, 16, 32),
          ("5/3", 3, 5, 30), ("7/4", 4, 7, 32), ("3/2", 2, 3, 36), ("decimate by 6", 1, 6, 32), ("decimate by 12", 1, 12, 32), ("7/3", 3, 7, 32), ("4/5", 5, 4, 32),
          ("interpolate x2", 2, 1, 32), ("interpolate x4", 4, 1, 32), ("decimate by 3", 1, 3, 36), ("decimate by 5", 1, 5, 30), ("2/3", 3, 2, 32), ("3/4", 4, 3, 36)]
x = api.DeviceArray(w * n)            # float32 stream; its first 2^29 bytes double as the u8 stream
x.fill_synth(synth.SEED)
t = api.Timer()
warm = False
print(f"# 2^{n.bit_length() - 1} {'real' if REAL else 'complex'} samples in; SFE_RT_DMA_U8={os.environ.get('SFE_RT_DMA_U8', '1')}")
print(f"{'shape':26s} {'U':>2s} {'step':>4s} {'u8 ms':>8s} {'frac':>6s} {'f32 ms':>8s} {'frac':>6s}")
for name, U, step, lp in SHAPES:
    rate = float(np.float32(step) / np.float32(U))
    taps = synth.lowpass_taps(lp * U, 0.9 * min(1.0 / U, 1.0 / step), gain=float(U))
    cap = n * U // step + 64
    y = api.DeviceArray(w * cap)
    res = []
    for u8 in (True, False):
        r = api.Rs(taps, U, 4096, mode=lib.RS_RESAMPLE, data_complex=not REAL)
        if u8:
            r.set_input_format(lib.FMT_U8)
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
        res += [ms, ((0.5 if u8 else 2.0) * w * 2 * n + 4.0 * w * k) / 1e9 / ms / 8.0]
        r.close()
    y.free()
    print(f"{name:26s} {U:2d} {step:4d} {res[0]:8.4f} {res[1]:6.3f} {res[2]:8.4f} {res[3]:6.3f}", flush=True)
