#!/usr/bin/env python3
"""Which bulk calls does the library REFUSE?  A sweep over (upsample, rate, taps per phase, stream type, input format, exact mode, blksize)
through sfe_dsp_rs_process_stream on a small stream; prints every combination that returns an error, with the message.  (Round 5: the
reference's classes take any rate >= 1 / upsample resp. >= 1 -- a refusal here is a gap, not a speed.)"""
import itertools
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from simplefe_amd import api, lib, synth  # noqa: E402

n = 64 * 4096
x = api.DeviceArray(2 * n)
x.fill_synth(synth.SEED)
bad, tried = {}, 0
RATES = [(1, 1.0), (1, 2.0), (1, 7.0), (1, 64.0), (1, 100.0), (1, 128.0), (1, 1000.0), (1, 2.5), (1, 1.77), (1, 100.3), (3, 5.0 / 3.0), (3, 1.77), (3, 0.77), (3, 1.0 / 3.0), (3, 40.0),
         (8, 1.0 / 8.0), (8, 0.3), (9, 10.0 / 9.0), (24, 25.0 / 24.0), (160, 147.0 / 160.0), (160, 0.5), (160, 1.77), (12, 1.0 / 12.0), (12, 1.003)]
for (U, rate), plen, fmt, exact, B in itertools.product(RATES, (1, 7, 32, 127, 400), ("cf32", "f32", "u8c", "u8r"), (False, True), (4096, 1000, 256)):
    if (plen + 0) > B or (fmt.startswith("u8") and exact):
        continue
    taps = synth.lowpass_taps(max(U, plen * U - (U > 1)), 0.9 * min(1.0 / U, 1.0 / max(rate * U, 1.0)), gain=float(U))
    cplx = fmt in ("cf32", "u8c")
    w = 2 if cplx else 1
    m = (n // B) * B // 4
    cap = int(m / rate) + 4 * (m // B) + 4096
    tried += 1
    try:
        r = api.Rs(taps, U, B, mode=lib.RS_RESAMPLE, data_complex=cplx)
        r.set_exact(exact)
        if fmt.startswith("u8"):
            r.set_input_format(lib.FMT_U8)
        y = api.DeviceArray(w * cap)
        r.process_stream(x, m, y, cap, float(np.float32(rate)))
        api.sync()
        y.free()
        r.close()
    except Exception as e:                   # noqa: BLE001
        key = str(e).split(": ", 1)[-1][:110]
        bad.setdefault(key, []).append((U, round(rate, 4), plen, fmt, exact, B))
print(f"{tried} combinations tried, {sum(len(v) for v in bad.values())} refused")
for k, v in bad.items():
    print(f"-- {len(v):4d} x  {k}")
    for c in v[:12]:
        print("        ", c)
