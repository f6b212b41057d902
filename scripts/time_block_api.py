#!/usr/bin/env python3
"""Latency of the reference-shaped per-block calls (one GPU round trip each): blkconv.process() on its
own pinned buffer (libdsp/blkconv.h:44-47), sfe_dsp_fir_process_host on caller memory, and the
resamplers' process() (libdsp/resample.h:52), at scheduler-sized blocks."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplefe_amd import api, lib, synth  # noqa: E402

L = lib.load()
taps = synth.taps_cfg2()
reps = int(os.environ.get("REPS", "2000"))


def per_call(fn):
    for _ in range(50):
        fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e6


# the floor under every per-block call: one trivial kernel launched and waited for (a 4-byte memset + a synchronise)
_d = api.DeviceArray(16)
us = per_call(lambda: (api.check(L.sfe_dsp_memset(_d.ptr, 0, 4, None)), api.check(L.sfe_dsp_sync(None))))
print(f"floor: a 4-byte device memset + synchronise: {us:6.1f} us per call")

b = api.blkconv(np.real(taps), 4096)
blk = b.get_blksize()
b.get_process_buf()[:blk] = np.random.default_rng(1).standard_normal(blk).astype(np.float32)
us = per_call(b.process)
print(f"blkconv(256 taps, 4096).process(): {us:6.1f} us per block of {blk} real samples = {blk / us:6.1f} MS/s")

for call in (1024, 4096, 16384, 65536):
    f = api.Fir(taps, data_complex=True)
    x = synth.synth_cf32(call)
    y = np.empty_like(x)
    us = per_call(lambda: api.check(L.sfe_dsp_fir_process_host(f._h, x.ctypes.data, y.ctypes.data, call)))
    print(f"fir_process_host, {call:6d} cf32 per call: {us:6.1f} us = {call / us:6.1f} MS/s")

for name, cls, U, rate in (("resample 5/3", api.resample, 3, 5.0 / 3.0), ("decimate 8", api.decimate, 1, 8.0)):
    B = 4096
    r = cls(synth.taps_cfg3() if U == 3 else synth.taps_cfg4(), U, B)
    x = np.random.default_rng(2).standard_normal(B).astype(np.float32)
    us = per_call(lambda: r.process(x, B * U, rate))
    print(f"{name}.process(), {B} real samples per call: {us:6.1f} us = {B / us:6.1f} MS/s")
