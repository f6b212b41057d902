#!/usr/bin/env python3
"""A diagnostic kernel variant must produce the product kernel's bits: same arithmetic, different
data movement.  Usage: check_variant.py W [X]   (through libsfe_dsp_diag.so)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplefe_amd import build, lib  # noqa: E402
lib.LIB_PATH = build.build_lib(diag=True)
from simplefe_amd import api, synth  # noqa: E402

a, b = (sys.argv[1:] + ["X"])[:2] if len(sys.argv) > 1 else ("W", "X")
out = {}
for n in (1 << 22, 3840 * 5 + 17, 100, 1 << 16):
    x = api.DeviceArray(2 * n)
    x.fill_synth(synth.SEED, channel=3)
    for v in (a, b):
        os.environ["SFE_FIR_VARIANT"] = v
        f = api.Fir(synth.taps_cfg2(), data_complex=True, algo=lib.FIR_ALGO_FFT)
        y = api.DeviceArray(2 * n)
        f.process_stream(x, y, n)
        f.process_stream(x, y, n)          # second call: carried history
        out[v] = y.to_numpy()
    same = np.array_equal(out[a], out[b])
    print(f"n={n}: {a} vs {b}: {'identical' if same else 'DIFFERENT max|d|=%g' % np.abs(out[a] - out[b]).max()}")
    if not same:
        sys.exit(1)
