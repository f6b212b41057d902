#!/usr/bin/env python3
"""Host-buffer (class-compatible) path: blkconv::process() on the pinned object buffer, i.e.
H2D + kernel + D2H per block.  Reports the PCIe-inclusive rate DESIGN.md quotes (never the
bench `value`)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplefe_amd import api, synth  # noqa: E402

taps = synth.taps_cfg2()
for fft_len in (4096, 1 << 16, 1 << 20, 1 << 24):
    c = api.blkconv(taps, fft_len)
    blk = c.get_blksize()
    buf = c.get_process_buf()
    buf[:blk] = synth.synth_f32(blk)
    for _ in range(3):
        c.process()
    reps = max(3, int(2e8 // blk))
    reps = min(reps, 2000)
    t0 = time.perf_counter()
    for _ in range(reps):
        c.process()
    dt = time.perf_counter() - t0
    print(f"blkconv fft_len {fft_len:9d} blk {blk:9d}: {dt / reps * 1e6:10.1f} us/block  "
          f"{reps * blk / dt / 1e6:9.1f} real MS/s  ({reps * blk * 8 / dt / 1e9:6.2f} GB/s over PCIe, both directions)")
