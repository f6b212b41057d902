#!/usr/bin/env python3
"""One shape through the direct general-rate kernel (poly_seg_kernel), for counters: 2^27 real samples, 3 phases, rate 1.77, TAPS per phase (default 8)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from simplefe_amd import api, lib, synth  # noqa: E402

n = 1 << 27
plen = int(os.environ.get("TAPS", "8"))
x = api.DeviceArray(n)
x.fill_synth(synth.SEED)
taps = synth.lowpass_taps(plen * 3, 0.1, gain=3.0)
cap = int(n / 1.77) + 4 * (n // 4096) + 4096
y = api.DeviceArray(cap)
r = api.Rs(taps, 3, 4096, mode=lib.RS_RESAMPLE, data_complex=False)
r.set_algo(lib.RS_ALGO_DIRECT)
t = api.Timer()
for _ in range(3):
    r.process_stream(x, n, y, cap, float(np.float32(1.77)))
t.start()
r.process_stream(x, n, y, cap, float(np.float32(1.77)))
t.stop()
print(f"{t.elapsed_ms():.4f} ms")
