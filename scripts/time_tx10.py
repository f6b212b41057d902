#!/usr/bin/env python3
"""256-tap FIR on 2^28 cf32 samples with the 10-bit transmit packing fused into the store, beside
float32 output; and the real-stream (bpsk chain) form on 2^29 samples."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplefe_amd import api, lib, synth  # noqa: E402

L = lib.load()
n = 1 << 28
x = api.DeviceArray(2 * n)
x.fill_synth(synth.SEED)
y = api.DeviceArray(2 * n)
for cplx, fmt in ((True, lib.FMT_F32), (True, lib.FMT_TX10), (False, lib.FMT_F32), (False, lib.FMT_TX10)):
    f = api.Fir(synth.taps_cfg2(), data_complex=cplx, algo=lib.FIR_ALGO_FFT)
    if fmt != lib.FMT_F32:
        f.set_output_format(fmt)
    m = n if cplx else 2 * n
    ts = []
    for it in range(60):
        t = api.Timer()
        t.start()
        f.process_stream(x, y, m)
        t.stop()
        ts.append(t.elapsed_ms())
    print(f"complex={cplx} out={'tx10' if fmt != lib.FMT_F32 else 'f32'}: mean[10:] {sum(ts[10:]) / len(ts[10:]):.4f} ms  min {min(ts):.4f}", flush=True)
