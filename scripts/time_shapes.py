#!/usr/bin/env python3
"""Integer-step resampling shapes OUTSIDE the compiled (SP, UP) tables (VERDICT r3 missing 4): every
interpolating ratio (UP > SP -- what `resample` can do and `decimate` cannot, libdsp/resample.cxx:91 against
libdsp/decimate.cxx:75-78) and decimations the tables skip.  2^28 cf32 samples in (LOG2N), prototypes of
32 * U taps (32 per polyphase arm), the product library's default dispatch; HIP events on the launch stream,
median of 9 x 3 launches, plain allocations.  BARE=1 (round 5; loads the DIAGNOSTIC library, which holds sfe_dsp_probe_pair): beside
each row the bare read : write mix of the shape -- 32 KiB read per workgroup, the output written in proportion, nothing
computed -- as time, as `frac`, and the kernel's time as a share of it.  frac = algorithmic bytes (8 B per input + 8 B per output) / time / 8 TB/s; the bar
north_star sets is 0.40.  Also checked here: the fused result against the float64 definition on a window.

    python scripts/time_shapes.py > profiles/r04/shapes.txt
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplefe_amd import build, lib  # noqa: E402
if os.environ.get("SFE_LIB"):
    lib.LIB_PATH = os.environ["SFE_LIB"]
BARE = os.environ.get("BARE") == "1"
if BARE:
    lib.LIB_PATH = build.build_lib(diag=True)      # the same kernels; the diagnostic library adds the bare-mix probe
from simplefe_amd import api, synth  # noqa: E402

log2n = int(os.environ.get("LOG2N", "28"))
n = 1 << log2n
# name, U, step (= rate * U): rate = step / U
SHAPES = [("interpolate x2 (rate 1/2)", 2, 1), ("3/4 (4 out per 3 in)", 4, 3), ("4/3 (3 out per 4 in)", 3, 4),
          ("decimate by 6", 1, 6), ("decimate by 7", 1, 7), ("decimate by 16", 1, 16),
          ("interpolate x4 (rate 1/4)", 4, 1), ("7/4", 4, 7), ("2/3 (3 out per 2 in)", 3, 2), ("3/2 (2 out per 3 in)", 2, 3), ("7/3 (3 out per 7 in)", 3, 7), ("4/5 (5 out per 4 in)", 5, 4),
          ("interpolate x3 (rate 1/3)", 3, 1), ("interpolate x8 (rate 1/8)", 8, 1)]
if os.environ.get("EXTRA") == "1":          # round 5: long decimations, for the tile-size rule of poly_rt_kernel
    SHAPES += [("decimate by 12", 1, 12), ("decimate by 13", 1, 13), ("decimate by 24", 1, 24), ("decimate by 32", 1, 32), ("decimate by 48", 1, 48)]
if os.environ.get("EXTRA") == "2":          # round 5: odd input steps, for the LDS-DMA form (poly_rt_dma.hip)
    SHAPES += [("decimate by 3", 1, 3), ("decimate by 5", 1, 5), ("decimate by 9", 1, 9), ("decimate by 13", 1, 13), ("decimate by 15", 1, 15),
               ("5/2 (2 out per 5 in)", 2, 5), ("9/4 (4 out per 9 in)", 4, 9), ("9/2 (2 out per 9 in)", 2, 9), ("9/5 (5 out per 9 in)", 5, 9),
               ("11/8 (8 out per 11 in)", 8, 11), ("decimate by 10", 1, 10), ("decimate by 12", 1, 12), ("decimate by 24", 1, 24),
               ("decimate by 32", 1, 32), ("decimate by 48", 1, 48), ("6/5 (5 out per 6 in)", 5, 6), ("10/3 (3 out per 10 in)", 3, 10),
               ("5/6 (6 out per 5 in)", 6, 5), ("7/8 (8 out per 7 in)", 8, 7), ("3/5 (5 out per 3 in)", 5, 3), ("2/5 (5 out per 2 in)", 5, 2)]
if os.environ.get("EXTRA") == "3":          # round 5: the interpolators between x4 and x8 (where does the LDS-DMA form overtake poly_rt1_kernel?)
    SHAPES += [("interpolate x5 (rate 1/5)", 5, 1), ("interpolate x6 (rate 1/6)", 6, 1), ("interpolate x7 (rate 1/7)", 7, 1)]
if os.environ.get("SHAPES"):
    want = os.environ["SHAPES"].split(",")
    SHAPES = [s for s in SHAPES if any(w in s[0] for w in want)]
x = api.DeviceArray(2 * n)
x.fill_synth(synth.SEED)
t = api.Timer()
warm = False
print(f"# 2^{log2n} cf32 in, 32 taps per polyphase arm; kernel = what the default dispatch ran; exact = the bit-exact mode's time")
print(f"{'shape':28s} {'U':>2s} {'step':>4s} {'out/in':>7s} {'ms':>8s} {'GB':>6s} {'frac':>6s} {'exact ms':>9s} {'rel-RMS vs f64':>15s}" + ("   bare mix ms / frac / kernel as a share of it" if BARE else ""))
for name, U, step in SHAPES:
    rate = float(np.float32(step) / np.float32(U))
    assert float(np.float32(rate) * np.float32(U)) == float(step), (name, "step must be exact in float32")
    taps = synth.lowpass_taps(32 * U, 0.9 * min(1.0 / U, 1.0 / step), gain=float(U))
    cap = n * U // step + 64
    import ctypes as C
    y = api.DeviceArray(2 * cap)
    bare = None
    if BARE and step <= 8 * U:              # (the probe writes at least one 4 KiB row per 32 KiB read: ratios beyond 8 : 1 are not its to give)
        ms = C.c_float(0.0)
        api.check(lib.load().sfe_dsp_probe_pair(x.ptr, 8 * n, y.ptr, 8 * (n * U // step), C.byref(ms)))
        bare = float(ms.value)
    res = {}
    for exact in (False, True, "direct", "fft"):
        r = api.Rs(taps, U, 4096, mode=lib.RS_RESAMPLE, data_complex=True)
        r.set_exact(exact is True)
        if exact in ("direct", "fft"):
            r.set_algo(lib.RS_ALGO_DIRECT if exact == "direct" else lib.RS_ALGO_FFT)
        for _ in range(4 if warm else 150):        # the first shape also carries the chip past its first ~100 ms after idling (5-6 % slow)
            k = r.process_stream(x, n, y, cap, rate)
        warm = True
        v = []
        for _ in range(9 if exact is False else 3):
            t.start()
            for _ in range(3):
                r.process_stream(x, n, y, cap, rate)
            t.stop()
            v.append(t.elapsed_ms() / 3)
        res[exact] = float(np.median(v))
        if exact is False:
            # a window in the middle against the float64 definition: s(p) = sum_j taps[p % U + j U] x[p / U - j]
            r.reset()                                                     # a fresh stream: output k sits at p = k * step
            k = r.process_stream(x, n, y, cap, rate)
            W = 4096
            k0 = (k // 2) - (k // 2) % U
            got = y.to_numpy(2 * W, offset=2 * k0)
            p = np.arange(k0, k0 + W, dtype=np.int64) * step
            n_lo = int(p[0] // U) - 32 - 1
            xs = x.to_numpy(2 * (int(p[-1] // U) - n_lo + 2), offset=2 * n_lo).astype(np.float64)
            xc = xs[0::2] + 1j * xs[1::2]
            tp = np.zeros(U * 32)
            tp[:len(taps)] = taps
            ref = np.zeros(W, complex)
            for j in range(32):
                ref += tp[(p % U) + j * U] * xc[(p // U) - j - n_lo]
            err = float(np.sqrt(np.sum(np.abs((got[0::2] + 1j * got[1::2]) - ref) ** 2) / np.sum(np.abs(ref) ** 2)))
        r.close()
    gb = 8.0 * (n + k) / 1e9
    tail = f"   bare {bare:.4f} / {gb / bare / 8.0:.3f} / {100.0 * bare / res[False]:.0f} %" if bare else ""
    print(f"{name:28s} {U:2d} {step:4d} {k / n:7.4f} {res[False]:8.4f} {gb:6.2f} {gb / res[False] / 8.0:6.3f} {res[True]:9.4f} {err:15.2e}   direct {res['direct']:.4f}  transform (where instantiated) {res['fft']:.4f}{tail}", flush=True)
    y.free()
