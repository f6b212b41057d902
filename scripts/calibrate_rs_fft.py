#!/usr/bin/env python3
"""Direct (tiled) vs transform-domain kernel over a grid of shapes: the data behind the selection
rule in api_plans.hip:get_fft_plan.  2^26 cf32 input samples per shape."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplefe_amd import api, lib, synth  # noqa: E402

n = 1 << 26
x = api.DeviceArray(2 * n)
x.fill_synth(synth.SEED)
y = api.DeviceArray(2 * (3 * n // 2 + 64))
rng = np.random.default_rng(0)
print("U S taps   SP UP  flops/sample   direct ms   fft ms   ratio")
for U, S, taps_list in ((1, 2, (64, 128, 256, 512)), (1, 3, (100, 200, 400)), (1, 4, (128, 256, 512, 1024)),
                        (1, 5, (160, 320, 640)), (1, 8, (256, 512, 1024)), (3, 5, (95, 190, 381, 760)),
                        (2, 3, (100, 200, 400)), (2, 5, (160, 320, 640)), (3, 4, (150, 300, 600)),
                        (4, 5, (200, 400, 800)), (2, 7, (280, 560))):
    g = int(np.gcd(U, S))
    SP, UP = S // g, U // g
    for nt in taps_list:
        taps = (rng.standard_normal(nt) / np.sqrt(nt)).astype(np.float32)
        res = []
        for algo in (lib.RS_ALGO_DIRECT, lib.RS_ALGO_FFT):
            r = api.Rs(taps, U, 4096, mode=lib.RS_RESAMPLE, data_complex=True, algo=algo)
            cap = int(n * U / S) + 16
            ts = []
            for it in range(12):
                r.reset()
                t = api.Timer()
                t.start()
                r.process_stream(x, n, y, cap, float(np.float32(S) / np.float32(U)))
                t.stop()
                ts.append(t.elapsed_ms())
            res.append(sum(ts[4:]) / 8)
            r.close()
        plen = (nt + U - 1) // U
        flops = 4.0 * plen * UP / SP
        print(f"{U} {S} {nt:5d}   {SP}  {UP}   {flops:8.0f}     {res[0]:8.3f}  {res[1]:8.3f}   {res[0] / res[1]:5.2f}", flush=True)
