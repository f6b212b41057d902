#!/usr/bin/env python3
"""The same question for the FIR: which bulk calls does sfe_dsp_fir_process_stream REFUSE?  (taps count, real / complex taps, real / complex data,
u8 input, 10-bit output, channels, call length)"""
import itertools
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from simplefe_amd import api, lib, synth  # noqa: E402

x = api.DeviceArray(1 << 24)
x.fill_synth(synth.SEED)
y = api.DeviceArray(1 << 25)
bad, tried = {}, 0
for n_taps, ctaps, cplx, fin, fout, nch, n in itertools.product((1, 2, 63, 256, 257, 1000, 2049, 3841, 5000, 20000, 70000), (False, True), (True, False),
                                                                 ("f32", "u8"), ("f32", "tx10"), (1, 3), (1, 100, 4096, 3840 * 5 + 7, 1 << 20)):
    if ctaps and (fout == "tx10" and not cplx):
        pass
    rng = np.random.default_rng(n_taps)
    taps = (rng.standard_normal(n_taps) / np.sqrt(n_taps)).astype(np.float32)
    if ctaps:
        taps = (taps + 1j * taps[::-1]).astype(np.complex64)
    tried += 1
    try:
        f = api.Fir(taps, data_complex=cplx, n_channels=nch)
        if fin == "u8":
            f.set_input_format(lib.FMT_U8)
        if fout == "tx10":
            f.set_output_format(lib.FMT_TX10)
        f.process_stream(x, y, n, in_stride=n + 16, out_stride=n + 16)
        api.sync()
        f.close()
    except Exception as e:                   # noqa: BLE001
        key = str(e).split(": ", 1)[-1][:120]
        bad.setdefault(key, []).append((n_taps, ctaps, cplx, fin, fout, nch, n))
print(f"{tried} combinations tried, {sum(len(v) for v in bad.values())} refused")
for k, v in bad.items():
    print(f"-- {len(v):4d} x  {k}")
    for c in v[:10]:
        print("        ", c)
