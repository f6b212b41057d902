"""A REAL float32 stream at the general rate (2^29 samples = the bytes of 2^28 cf32): the transform-domain kernel with two
blocks per transform (poly_gen.hip: REAL) against the direct kernel."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplefe_amd import api, lib, synth
n = 1 << 29
x = api.DeviceArray(n); x.fill_synth(synth.SEED)
cap = int(n / 1.77) + 131072
y = api.DeviceArray(cap)
t = api.Timer()
r = api.Rs(synth.taps_cfg3(), 3, 4096, mode=lib.RS_RESAMPLE, data_complex=False)
for rate in (1.77,):
    for algo, name in ((lib.RS_ALGO_AUTO, "default dispatch"), (lib.RS_ALGO_DIRECT, "direct form (poly_seg_kernel)")):
        r.set_algo(algo)
        r.reset()
        for _ in range(2): k = r.process_stream(x, n, y, cap, rate)
        v = []
        for _ in range(5):
            t.start(); r.process_stream(x, n, y, cap, rate); t.stop(); v.append(t.elapsed_ms())
        print(f"real stream 2^29 samples, rate {rate}, 381 taps in 3 phases, {name}: {k} out, median {np.median(v):.4f} ms  (frac {4.0 * (n + k) / np.median(v) / 1e-3 / 8e12:.3f})")
