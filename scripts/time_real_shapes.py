#!/usr/bin/env python3
"""Integer-step shapes outside the compiled tables on a REAL float32 stream (libdsp's native type: the reference's classes take float*): 2^29 real
samples (the bytes of 2^28 cf32), 32 taps per polyphase arm, the default dispatch; HIP events, median of 7 x 3 launches.
frac = (4 B per input + 4 B per output) / time / 8 TB/s.    python scripts/time_real_shapes.py > profiles/r05/shapes_real.txt"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplefe_amd import build, lib  # noqa: E402
if os.environ.get("DIAG") == "1":          # the diagnostic library: honours SFE_RT_DMA / SFE_RT_DMA_SP1 / SFE_RT_DMA_FORCE
    lib.LIB_PATH = build.build_lib(diag=True)
from simplefe_amd import api, synth  # noqa: E402

n = 1 << int(os.environ.get("LOG2N", "29"))
SHAPES = [("decimate by 6", 1, 6), ("decimate by 7", 1, 7), ("decimate by 8", 1, 8), ("decimate by 12", 1, 12), ("decimate by 16", 1, 16), ("7/4", 4, 7), ("7/3", 3, 7),
          ("4/5", 5, 4), ("2/5", 5, 2), ("3/5", 5, 3), ("interpolate x2", 2, 1), ("interpolate x3", 3, 1), ("interpolate x4", 4, 1), ("interpolate x5", 5, 1), ("interpolate x6", 6, 1), ("interpolate x7", 7, 1), ("interpolate x8", 8, 1)]
if os.environ.get("SHAPES"):
    SHAPES = [s for s in SHAPES if any(w in s[0] for w in os.environ["SHAPES"].split(","))]
x = api.DeviceArray(n)
x.fill_synth(synth.SEED)
t = api.Timer()
warm = False
print(f"# 2^{n.bit_length() - 1} REAL float32 samples in, 32 taps per polyphase arm, default dispatch")
print(f"{'shape':24s} {'U':>2s} {'step':>4s} {'ms':>8s} {'GB':>6s} {'frac':>6s}")
for name, U, step in SHAPES:
    rate = float(np.float32(step) / np.float32(U))
    taps = synth.lowpass_taps(32 * U, 0.9 * min(1.0 / U, 1.0 / step), gain=float(U))
    cap = n * U // step + 64
    y = api.DeviceArray(cap)
    r = api.Rs(taps, U, 4096, mode=lib.RS_RESAMPLE, data_complex=False)
    for _ in range(4 if warm else 150):            # the first shape also carries the chip past its first ~100 ms after idling (5-6 % slow)
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
    gb = 4.0 * (n + k) / 1e9
    print(f"{name:24s} {U:2d} {step:4d} {ms:8.4f} {gb:6.2f} {gb / ms / 8.0:6.3f}", flush=True)
    r.close()
    y.free()
