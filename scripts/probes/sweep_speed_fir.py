#!/usr/bin/env python3
"""Performance cliffs of the FIR bulk call: (taps count, real / complex taps, real / complex data, channels) at 2^26 samples, sorted worst first.
frac = algorithmic bytes (in + out) / time / 8 TB/s; beyond one transform's reach (3841 taps) a filter costs one pass per 3584-tap partition."""
import itertools
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from simplefe_amd import api, synth  # noqa: E402

n0 = 1 << 26
x = api.DeviceArray(2 * n0)
x.fill_synth(synth.SEED)
y = api.DeviceArray(2 * n0 + 64)
t = api.Timer()
rows = []
for n_taps, ctaps, cplx, nch in itertools.product((1, 16, 63, 256, 257, 512, 1000, 2049, 3841, 3842, 5000, 20000, 70000), (False, True), (True, False), (1, 16)):
    rng = np.random.default_rng(n_taps)
    taps = (rng.standard_normal(n_taps) / np.sqrt(n_taps)).astype(np.float32)
    if ctaps:
        taps = (taps + 1j * taps[::-1]).astype(np.complex64)
    w = 2 if cplx else 1
    n = (n0 * (2 // w)) // nch // (2 if (ctaps and not cplx) else 1)
    try:
        f = api.Fir(taps, data_complex=cplx, n_channels=nch)
        for _ in range(3):
            f.process_stream(x, y, n)
        v = []
        for _ in range(3):
            t.start()
            f.process_stream(x, y, n)
            t.stop()
            v.append(t.elapsed_ms())
        ms = float(np.median(v))
        wo = 2 if (cplx or ctaps) else 1
        rows.append((4.0 * (w + wo) * n * nch / 1e9 / ms / 8.0, ms, n_taps, "ctaps" if ctaps else "rtaps", "cplx" if cplx else "real", nch, n))
        f.close()
    except Exception as e:                   # noqa: BLE001
        rows.append((-1.0, 0.0, n_taps, "ctaps" if ctaps else "rtaps", "cplx" if cplx else "real", nch, str(e)[:70]))
rows.sort()
print(f"{'frac':>6s} {'ms':>9s} {'taps':>6s} {'':>5s} {'data':>5s} {'ch':>3s} n per channel")
for r in rows:
    print(f"{r[0]:6.3f} {r[1]:9.4f} {r[2]:6d} {r[3]:>5s} {r[4]:>5s} {r[5]:3d} {r[6]}")
