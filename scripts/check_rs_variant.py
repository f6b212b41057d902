#!/usr/bin/env python3
"""A diagnostic variant of the transform-domain resample kernel must produce the product kernel's bits: same
arithmetic, different data movement.  Usage: check_rs_variant.py A [t]   (through libsfe_dsp_diag.so)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplefe_amd import build, lib  # noqa: E402
lib.LIB_PATH = build.build_lib(diag=True)
from simplefe_amd import api, synth  # noqa: E402

a, b = (sys.argv[1:] + ["t"])[:2]
os.environ["SFE_RS_WG_FACTOR"] = "1"
for n in (1 << 24, 2310 * 7 + 5, 1 << 16, 5 * 100000):
    x = api.DeviceArray(2 * n)
    x.fill_synth(synth.SEED, channel=3)
    cap = int(n * 3 / 5) + 8
    out = {}
    for v in (a, b):
        os.environ["SFE_RS_VARIANT"] = v
        r = api.Rs(synth.taps_cfg3(), 3, 4096, mode=lib.RS_RESAMPLE, data_complex=True, algo=lib.RS_ALGO_FFT)
        y = api.DeviceArray(2 * cap)
        k1 = r.process_stream(x, n, y, cap, 5.0 / 3.0)
        k2 = r.process_stream(x, n, y, cap, 5.0 / 3.0)          # second call: carried history
        out[v] = (k1, k2, y.to_numpy(2 * k2))
    same = out[a][:2] == out[b][:2] and np.array_equal(out[a][2], out[b][2])
    print(f"n={n}: {a} vs {b}: n_out {out[a][:2]} {'identical' if same else 'DIFFERENT'}")
    if not same:
        sys.exit(1)
