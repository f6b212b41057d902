#!/usr/bin/env python3
"""BASELINE configs[3] (decimate by 8, 64 taps, 2^30 cf32) through its compile-time kernel poly_tiled_kernel<8, 1> and, forced, through the
runtime-shape LDS-DMA kernel (poly_rt_dma.hip; SFE_RT_DMA_FORCE=1, DIAGNOSTIC library), interleaved in one process; also resample-like
compiled shapes on request.   python scripts/ab_dec8_dma.py > profiles/r05/decimate8_rt_dma_ab.txt"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplefe_amd import build, lib  # noqa: E402
lib.LIB_PATH = build.build_lib(diag=True)
from simplefe_amd import api, synth  # noqa: E402

n = 1 << 30
x = api.DeviceArray(2 * n)
x.fill_synth(synth.SEED)
cap = n // 8 + 64
y = api.DeviceArray(2 * cap)
r = api.Rs(synth.taps_cfg4(), 1, 4096, mode=lib.RS_DECIMATE, data_complex=True)
t = api.Timer()
res = {"poly_tiled_kernel<8,1>": [], "poly_rt_dma_kernel (forced)": []}
outs = {}
for rnd in range(int(os.environ.get("ROUNDS", "8")) + 1):
    for name, env in (("poly_tiled_kernel<8,1>", "0"), ("poly_rt_dma_kernel (forced)", "1")):
        os.environ["SFE_RT_DMA_FORCE"] = env
        r.reset()
        t.start()
        for _ in range(3):
            k = r.process_stream(x, n, y, cap, 8.0)
        t.stop()
        if rnd:
            res[name].append(t.elapsed_ms() / 3)
        else:
            outs[name] = y.to_numpy(1 << 16, offset=2 * (k // 2))
print("# scripts/ab_dec8_dma.py: decimate by 8, 64 taps, 2^30 cf32 -> 2^27; HIP events around 3 calls, interleaved; algorithmic 9.66 GB")
for kname, v in res.items():
    print(f"{kname:30s} median {np.median(v):.4f} ms  min {min(v):.4f}  max {max(v):.4f}  -> frac {9.0 * n / np.median(v) / 1e6 / 8000:.3f}")
a, b = outs.values()
print("same bits in a window of the output:", bool(np.array_equal(a, b)), " max |difference|", float(np.abs(a - b).max()))
