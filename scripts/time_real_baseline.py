#!/usr/bin/env python3
"""The three BASELINE workloads on a REAL float32 stream (libdsp's native type: blkconv, resample and decimate take float*; a cf32 caller of the
reference runs I and Q as two real passes): 2^29 real samples = the bytes of 2^28 cf32; HIP events, 200 warm-up launches (the chip runs 5-6 % slow in its first ~100 ms after idling:
profiles/r03/fir_real_data.txt), then the median of 9 x 5 launches, default dispatch; the FIR on the complex stream of the same bytes beside it.
frac = (4 B per input + 4 B per output) / time / 8 TB/s.       python scripts/time_real_baseline.py > profiles/r05/baseline_real.txt"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplefe_amd import api, lib, synth  # noqa: E402

n = 1 << int(os.environ.get("LOG2N", "29"))
x = api.DeviceArray(n)
x.fill_synth(synth.SEED)
t = api.Timer()


def timed(call):
    for _ in range(200):
        k = call()
    v = []
    for _ in range(9):
        t.start()
        for _ in range(5):
            call()
        t.stop()
        v.append(t.elapsed_ms() / 5)
    return float(np.median(v)), k


print(f"# 2^{n.bit_length() - 1} REAL float32 samples in, default dispatch")
print(f"{'workload':44s} {'ms':>8s} {'GB':>6s} {'frac':>6s}")
y = api.DeviceArray(n)
f = api.Fir(synth.taps_cfg2(), data_complex=False)
ms, _ = timed(lambda: f.process_stream(x, y, n) or n)
print(f"{'blkconv, 256 taps (configs[1] on float*)':44s} {ms:8.4f} {8.0 * n / 1e9:6.2f} {8.0 * n / 1e9 / ms / 8.0:6.3f}", flush=True)
fc = api.Fir(synth.taps_cfg2(), data_complex=True)
ms, _ = timed(lambda: fc.process_stream(x, y, n // 2) or n)
print(f"{'  (the same bytes as 2^28 cf32, configs[1])':44s} {ms:8.4f} {8.0 * n / 1e9:6.2f} {8.0 * n / 1e9 / ms / 8.0:6.3f}", flush=True)
for name, taps, U, mode, rate in (("resample 5/3, 381 taps (configs[2])", synth.taps_cfg3(), 3, lib.RS_RESAMPLE, 5.0 / 3.0),
                                  ("decimate by 8, 64 taps (configs[3])", synth.taps_cfg4(), 1, lib.RS_DECIMATE, 8.0)):
    cap = int(n / rate) + 64
    r = api.Rs(taps, U, 4096, mode=mode, data_complex=False)
    ms, k = timed(lambda: r.process_stream(x, n, y, cap, rate))
    gb = 4.0 * (n + k) / 1e9
    print(f"{name:44s} {ms:8.4f} {gb:6.2f} {gb / ms / 8.0:6.3f}", flush=True)
    r.close()
