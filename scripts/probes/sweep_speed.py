#!/usr/bin/env python3
"""Where are the performance CLIFFS?  The bulk resampler call over (upsample, rate, taps per phase, stream type) at 2^26 samples in, default dispatch:
time, fraction of the 8 TB/s roofline by algorithmic bytes (all input + output samples), sorted worst first."""
import itertools
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from simplefe_amd import api, lib, synth  # noqa: E402

n0 = 1 << int(os.environ.get("LOG2N", "26"))
x = api.DeviceArray(2 * n0)
x.fill_synth(synth.SEED)
t = api.Timer()
RATES = [(1, 1.0), (1, 2.0), (1, 3.0), (1, 7.0), (1, 8.0), (1, 64.0), (1, 100.0), (1, 1000.0), (1, 2.5), (1, 1.77), (1, 100.3), (2, 0.5), (2, 1.5), (3, 5.0 / 3.0), (3, 1.77), (3, 0.77),
         (3, 1.0 / 3.0), (4, 1.25), (5, 0.8), (8, 1.0 / 8.0), (8, 0.3), (8, 1.77), (9, 10.0 / 9.0), (16, 17.0 / 16.0), (16, 1.77), (24, 25.0 / 24.0), (32, 1.0 / 32.0), (32, 1.77),
         (160, 147.0 / 160.0), (160, 1.77)]
rows = []
for (U, rate), plen, cplx in itertools.product(RATES, (8, 32, 127), (True, False)):
    rate = float(np.float32(rate))
    w = 2 if cplx else 1
    n = n0 * (2 // w)
    if n / rate > 3 * n0 * 2 // w:
        n //= 8                                        # strong interpolation: keep the output in memory
    taps = synth.lowpass_taps(plen * U, 0.9 * min(1.0 / U, 1.0 / max(rate * U, 1.0)), gain=float(U))
    cap = int(n / rate) + 4 * (n // 4096) + 4096
    try:
        y = api.DeviceArray(w * cap)
        r = api.Rs(taps, U, 4096, mode=lib.RS_RESAMPLE, data_complex=cplx)
        r.set_exact(os.environ.get("EXACT") == "1")          # EXACT=1: the bit-exact kernels
        for _ in range(int(os.environ.get("WARM", "12"))):      # (an integer-step stream cut into equal calls cycles through up to `step` phase offsets, each with a plan
            k = r.process_stream(x, n, y, cap, rate)               # of its own built on first use -- ~1 ms of host work for a transform-domain plan: not what this sweep is after)
        v = []
        for _ in range(3):
            t.start()
            r.process_stream(x, n, y, cap, rate)
            t.stop()
            v.append(t.elapsed_ms())
        ms = float(np.median(v))
        rows.append((4.0 * w * (n + k) / 1e9 / ms / 8.0, ms, U, round(rate, 4), plen, "cplx" if cplx else "real", n, k))
        r.close()
        y.free()
    except Exception as e:                   # noqa: BLE001
        rows.append((-1.0, 0.0, U, round(rate, 4), plen, "cplx" if cplx else "real", n, str(e)[:60]))
rows.sort()
print(f"{'frac':>6s} {'ms':>9s} {'U':>4s} {'rate':>9s} {'taps/ph':>7s} {'type':>5s} {'n_in':>10s} n_out")
for f, ms, U, rate, plen, ty, n, k in rows:
    print(f"{f:6.3f} {ms:9.4f} {U:4d} {rate:9.4f} {plen:7d} {ty:>5s} {n:10d} {k}")
