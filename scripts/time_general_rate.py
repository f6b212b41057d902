#!/usr/bin/env python3
"""General (non-integer-step) bulk resample: where does the time go (host replay vs kernel)?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplefe_amd import api, lib, synth
log2n = int(os.environ.get("LOG2N", "24"))
n = 1 << log2n
taps = synth.lowpass_taps(31, 0.18, gain=4.0)
x = api.DeviceArray(2 * n); x.fill_synth(synth.SEED)
cap = int(n / 1.77) + 16
y = api.DeviceArray(2 * cap)
r = api.Rs(taps, 4, 4096, mode=lib.RS_RESAMPLE, data_complex=True)
for rate in (1.77, 2.0):
    r.reset()
    t0 = time.perf_counter()
    r.process_stream(x, n, y, cap, rate); api.sync()
    print(f"rate {rate}: first call (builds the plan memo) {(time.perf_counter() - t0) * 1e3:.2f} ms")
    t0 = time.perf_counter()
    for _ in range(3):
        k = r.process_stream(x, n, y, cap, rate)
    api.sync()
    dt = (time.perf_counter() - t0) / 3
    print(f"rate {rate}: 2^{log2n} cf32 in -> {k} out: {dt * 1e3:.2f} ms per call ({n / dt / 1e6:.0f} MS/s)")
