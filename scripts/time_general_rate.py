#!/usr/bin/env python3
"""General (non-integer-step) bulk resample: where does the time go (host replay vs kernel)?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplefe_amd import lib
if os.environ.get("SFE_LIB"):                      # a saved build to compare against
    lib.LIB_PATH = os.environ["SFE_LIB"]
from simplefe_amd import api, synth
log2n = int(os.environ.get("LOG2N", "24"))
n = 1 << log2n
x = api.DeviceArray(2 * n); x.fill_synth(synth.SEED)
cap = int(n / 0.77) + 131072         # room for the interpolating rate too (0.77: the rate of libdsp/test/test_resample.py:24)
y = api.DeviceArray(2 * cap)
# a short filter (31 taps in 4 phases: 8 per dot product) and BASELINE cfg3's (381 taps in 3 phases: 127 per dot product)
shapes = (("31 taps, U = 4", synth.lowpass_taps(31, 0.18, gain=4.0), 4, (1.77, 2.0)),
          ("381 taps, U = 3", synth.taps_cfg3(), 3, (1.77, 0.77)))
if os.environ.get("GENERAL_ONLY"):         # counter passes: the long filter alone (the transform-domain kernel's shape)
    shapes = shapes[1:]
if os.environ.get("RATES"):                # RATES=1.77: ONE rate per process, so that a counter pass averages launches of one shape
    rr = tuple(float(v) for v in os.environ["RATES"].split(","))     # (round 4's counters mixed 1.77 and 0.77)
    shapes = tuple((nm, tp, U, rr) for nm, tp, U, _ in shapes)
algos = ((lib.RS_ALGO_AUTO, "default dispatch"), (lib.RS_ALGO_DIRECT, "direct form (poly_seg_kernel)"))
if os.environ.get("DEFAULT_ONLY"):         # counter passes: the default dispatch alone
    algos = algos[:1]
timer = api.Timer()
for name, taps, U, rates in shapes:
  r = api.Rs(taps, U, 4096, mode=lib.RS_RESAMPLE, data_complex=True)
  print(f"-- {name}")
  for rate in rates:
    r.reset()
    t0 = time.perf_counter()
    r.process_stream(x, n, y, cap, rate); api.sync()
    print(f"rate {rate}: first call (builds the plan memo) {(time.perf_counter() - t0) * 1e3:.2f} ms")
    t0 = time.perf_counter()
    for _ in range(3):
        k = r.process_stream(x, n, y, cap, rate)
    api.sync()
    dt = (time.perf_counter() - t0) / 3
    print(f"rate {rate}: 2^{log2n} cf32 in -> {k} out: {dt * 1e3:.2f} ms per call ({n / dt / 1e6:.0f} MS/s)")
    # ... and what of that is the kernel (HIP events on the launch stream) with each of the two kernels
    for algo, label in algos:
        r.set_algo(algo)
        r.process_stream(x, n, y, cap, rate)
        v = []
        for _ in range(7):
            timer.start()
            r.process_stream(x, n, y, cap, rate)
            timer.stop()
            v.append(timer.elapsed_ms())
        print(f"rate {rate}: {label}: HIP events around one call: median {np.median(v):.4f} ms  min {min(v):.4f}  max {max(v):.4f}")
    r.set_algo(lib.RS_ALGO_AUTO)
