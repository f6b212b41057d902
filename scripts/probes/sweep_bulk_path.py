#!/usr/bin/env python3
"""The bulk call (sfe_dsp_rs_process_stream through api.Rs.resample_array) over the refusal sweep's matrix, this time CHECKED: exact mode against
the oracle's bits and per-stream output count, the default (fused / transform-domain) mode within 1e-5 rel-RMS of it; real and complex streams,
the stream cut into two bulk calls.  Prints the combinations that differ or raise."""
import itertools
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from simplefe_amd import api, lib, synth  # noqa: E402
from oracle import binding as orc  # noqa: E402

RATES = [(1, 1.0), (1, 2.0), (1, 7.0), (1, 64.0), (1, 100.0), (1, 128.0), (1, 1000.0), (1, 2.5), (1, 1.77), (1, 100.3), (3, 5.0 / 3.0), (3, 1.77), (3, 0.77), (3, 1.0 / 3.0), (3, 40.0),
         (8, 1.0 / 8.0), (8, 0.3), (9, 10.0 / 9.0), (24, 25.0 / 24.0), (160, 147.0 / 160.0), (160, 0.5), (160, 1.77), (12, 1.0 / 12.0), (12, 1.003)]
bad, tried, worst = {}, 0, 0.0
for (U, rate), plen, B, cplx, exact in itertools.product(RATES, (1, 7, 32, 127), (4096, 1000, 256), (True, False), (True, False)):
    if plen > B:
        continue
    rate = float(np.float32(rate))
    taps = synth.lowpass_taps(max(U, plen * U - (U > 1)), 0.9 * min(1.0 / U, 1.0 / max(rate * U, 1.0)), gain=float(U))
    w = 2 if cplx else 1
    n = 48 * B if rate < 50 else 400 * B
    x = synth.synth_f32(w * n, ch=tried % 50)
    tried += 1
    try:
        r = api.Rs(taps, U, B, mode=lib.RS_RESAMPLE, data_complex=cplx)
        r.set_exact(exact)
        y = r.resample_array(x, rate, chunk=(n // B // 3) * B)[0]
        r.close()
        for part in range(w):
            ref, _ = orc.Resample(taps, U, B).stream(np.ascontiguousarray(x[part::w]), rate)
            got = y[part::w]
            if len(ref) - len(got) not in (0, 1):
                bad.setdefault("output count differs", []).append((U, round(rate, 4), plen, B, cplx, exact, len(got), len(ref)))
                break
            if exact:
                if not np.array_equal(got.view(np.uint32), ref[: len(got)].view(np.uint32)):
                    bad.setdefault("exact mode: bits differ", []).append((U, round(rate, 4), plen, B, cplx))
                    break
            elif len(got):
                e = synth.rel_rms(got, ref[: len(got)])
                worst = max(worst, e)
                if not e <= 1e-5:
                    bad.setdefault("default mode: beyond 1e-5", []).append((U, round(rate, 4), plen, B, cplx, e))
                    break
    except Exception as e:                   # noqa: BLE001
        bad.setdefault(str(e).split(": ", 1)[-1][:110], []).append((U, round(rate, 4), plen, B, cplx, exact))
print(f"{tried} combinations tried, {sum(len(v) for v in bad.values())} bad; worst default-mode rel-RMS {worst:.2e}")
for k, v in bad.items():
    print(f"-- {len(v):4d} x  {k}")
    for c in v[:12]:
        print("        ", c)
