#!/usr/bin/env python3
"""The reference's own blkconv class on the GPU box (blkconv.cxx on ROCm's libhipfftw: host
buffers, hipFFT plans executed per process() call) beside this repo's drop-in class on the same
blocks.  A curiosity for DESIGN.md section 5, not a bench line."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import binding as orc  # noqa: E402
from simplefe_amd import api, synth  # noqa: E402

taps = synth.taps_cfg2()
for fft_len in (4096, 1 << 16, 1 << 20):
    r = orc.RefBlkconv(taps, fft_len)
    c = api.blkconv(taps, fft_len)
    blk = r.blk
    x = synth.synth_f32(blk)
    for name, obj, buf, proc in (("reference/hipfftw", r, r.buf, r.process), ("drop-in", c, c.get_process_buf(), c.process)):
        buf[:blk] = x
        for _ in range(3):
            proc()
        reps = max(5, min(500, int(5e7 // blk)))
        t0 = time.perf_counter()
        for _ in range(reps):
            proc()
        dt = (time.perf_counter() - t0) / reps
        print(f"fft_len {fft_len:8d} {name:18s} {dt * 1e6:10.1f} us/block  {blk / dt / 1e6:9.1f} real MS/s", flush=True)
