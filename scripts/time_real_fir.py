#!/usr/bin/env python3
"""Real float32 stream through the 256-tap FIR (libdsp's native case): throughput of the PAIR mode."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplefe_amd import api, lib, synth
n = 1 << 29          # real samples (2 GiB in, 2 GiB out: same bytes as 2^28 cf32)
x = api.DeviceArray(n); x.fill_synth(synth.SEED)
y = api.DeviceArray(n)
f = api.Fir(synth.taps_cfg2(), data_complex=False, algo=lib.FIR_ALGO_FFT)
t = api.Timer()
for _ in range(200): f.process_stream(x, y, n)          # past the chip's first ~100 ms after idling (DVFS) and the variant-free PAIR kernel's first launches
t.start()
for _ in range(100): f.process_stream(x, y, n)
t.stop()
ms = t.elapsed_ms() / 100
print(f"real f32 2^29 samples: {ms:.4f} ms  {n / ms / 1e3:.0f} real MS/s  {8.0 * n / ms / 1e6:.0f} GB/s alg = {8.0 * n / ms / 1e6 / 80:.1f}% of 8 TB/s")
# the same bytes as a complex stream, same process, for scale
xc = api.DeviceArray(n); xc.fill_synth(synth.SEED)
fc = api.Fir(synth.taps_cfg2(), data_complex=True, algo=lib.FIR_ALGO_FFT)
for _ in range(100): fc.process_stream(xc, y, n // 2)
t.start()
for _ in range(100): fc.process_stream(xc, y, n // 2)
t.stop()
ms = t.elapsed_ms() / 100
print(f"cf32 2^28 samples (same bytes): {ms:.4f} ms  {8.0 * n / ms / 1e6:.0f} GB/s alg = {8.0 * n / ms / 1e6 / 80:.1f}% of 8 TB/s")
