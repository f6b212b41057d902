#!/usr/bin/env python3
"""Multi-channel resample 5/3 (381 taps) through the transform-domain kernel: ms per launch for
C channels x 2^28/C samples (the same bytes at every C) -- the work-counter form serves every C
(round 2: only C = 1, the fixed-stride walk otherwise)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplefe_amd import api, lib, synth  # noqa: E402

log2n = int(os.environ.get("LOG2N", "28"))
taps = synth.taps_cfg3()
rate = float(np.float32(5) / np.float32(3))
x = api.DeviceArray(2 << log2n)
x.fill_synth(synth.SEED)
y = api.DeviceArray(2 * ((3 << log2n) // 5 + 64 * 16))
for C in (1, 2, 8, 64):
    n = (1 << log2n) // C
    cap = n * 3 // 5 + 8
    r = api.Rs(taps, 3, 4096, mode=lib.RS_RESAMPLE, data_complex=True, n_channels=C)
    ts = []
    for it in range(60):
        t = api.Timer()
        t.start()
        k = r.process_stream(x, n, y, cap, rate, in_stride=n, out_stride=cap)
        t.stop()
        ts.append(t.elapsed_ms())
    print(f"{C:3d} channel(s) x 2^{log2n}/{C}: {k} out per channel, mean[20:] {np.mean(ts[20:]):.4f} ms  min {min(ts):.4f}", flush=True)
    r.close()
