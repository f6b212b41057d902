#!/usr/bin/env python3
"""Is a kernel's time DATA dependent?  Round 4 found the FIR 10 % faster on an input of zeros than on the synthetic stream
(the chip's power limit beside the HBM roofline).  The same question for every bulk kernel of the path:
each timed on one pair of plain allocations, the input holding zeros, then the synthetic stream, then zeros again (HIP events on
the launch stream, median of 7 x 5 launches after 20).

    python scripts/time_zeros_vs_data.py > profiles/r04/zeros_vs_data.txt
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplefe_amd import lib  # noqa: E402
from simplefe_amd import api, synth  # noqa: E402

t = api.Timer()


def timed(call):
    for _ in range(20):
        call()
    v = []
    for _ in range(7):
        t.start()
        for _ in range(5):
            call()
        t.stop()
        v.append(t.elapsed_ms() / 5)
    return float(np.median(v))


def three(x, call):
    out = []
    for what in ("zeros", "data", "zeros"):
        if what == "zeros":
            x.zero()
        else:
            x.fill_synth(synth.SEED)
        api.sync()
        out.append(timed(call))
    return out


print("# kernel, size: ms on an input of zeros | on the synthetic stream | on zeros again   (data / zeros)")
n = 1 << 28
x, y = api.DeviceArray(2 * n), api.DeviceArray(2 * n)
f = api.Fir(synth.taps_cfg2(), data_complex=True)
a = three(x, lambda: f.process_stream(x, y, n))
print("256-tap FIR (fir_fft4096_kernel), 2^28 cf32:            %.4f | %.4f | %.4f   (%.3f)" % (a[0], a[1], a[2], a[1] / (0.5 * (a[0] + a[2]))), flush=True)
# what gr-simplefe's source hands on: 8-bit samples as floats, (b - 128) / 127 (source_c_impl.cc:134-153) -- 256 distinct values
import ctypes as C
blk = 1 << 24
q = (np.round(synth.synth_f32(2 * blk, synth.SEED, 0) * 127.0) / 127.0).astype(np.float32)
for k in range(n // blk):
    api.check(lib.load().sfe_dsp_memcpy_h2d(C.c_void_p(x.ptr + 8 * blk * k), q.ctypes.data, q.nbytes, None))
api.sync()
t8 = timed(lambda: f.process_stream(x, y, n))
x.fill_synth(synth.SEED)
api.sync()
print("   the same FIR on 8-bit samples as floats ((b - 128) / 127, a 2^24-sample block repeated): %.4f ms, on the synthetic stream right after: %.4f" % (t8, timed(lambda: f.process_stream(x, y, n))), flush=True)
r = api.Rs(synth.taps_cfg3(), 3, 4096, mode=lib.RS_RESAMPLE, data_complex=True)
cap = n * 3 // 5 + 64
a = three(x, lambda: r.process_stream(x, n, y, cap, 5.0 / 3.0))
print("resample 5/3, 381 taps (poly_fft256_kernel), 2^28 cf32:  %.4f | %.4f | %.4f   (%.3f)" % (a[0], a[1], a[2], a[1] / (0.5 * (a[0] + a[2]))), flush=True)
r.close()
r = api.Rs(synth.lowpass_taps(64, 0.45, gain=2.0), 2, 4096, mode=lib.RS_RESAMPLE, data_complex=True)
m = n // 2 - 1024
a = three(x, lambda: r.process_stream(x, m, y, n, 0.5))
print("interpolate x2, 64 taps (poly_rt1_kernel), 2^27 cf32 in:  %.4f | %.4f | %.4f   (%.3f)" % (a[0], a[1], a[2], a[1] / (0.5 * (a[0] + a[2]))), flush=True)
r.close()
r = api.Rs(synth.taps_cfg3(), 3, 4096, mode=lib.RS_RESAMPLE, data_complex=True)
capg = int(n / 1.77) + 4096
a = three(x, lambda: r.process_stream(x, n, y, capg, 1.77))
print("general rate 1.77, 381 taps (poly_gen4096_kernel), 2^28:  %.4f | %.4f | %.4f   (%.3f)" % (a[0], a[1], a[2], a[1] / (0.5 * (a[0] + a[2]))), flush=True)
r.close()
x.free()
y.free()
n8 = 1 << 30
x, y = api.DeviceArray(2 * n8), api.DeviceArray(2 * (n8 // 8 + 64))
r = api.Rs(synth.taps_cfg4(), 1, 4096, mode=lib.RS_DECIMATE, data_complex=True)
a = three(x, lambda: r.process_stream(x, n8, y, n8 // 8 + 64, 8.0))
print("decimate by 8, 64 taps (poly_tiled_kernel), 2^30 cf32:    %.4f | %.4f | %.4f   (%.3f)" % (a[0], a[1], a[2], a[1] / (0.5 * (a[0] + a[2]))), flush=True)
