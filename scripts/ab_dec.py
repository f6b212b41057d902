#!/usr/bin/env python3
"""Interleaved ablation of the tiled decimator (decimate by 8, 64 taps, 2^30 cf32) through the DIAGNOSTIC
library: 0 = the one-tile kernel, 1 = no dot products, 2 = no LDS staging, 3 = neither (loads and stores only), 4 = its taps read by scalar loads;
tN = the streamed kernel with N consecutive tiles per workgroup (t1 = the one-tile kernel, P = the product's own choice),
t-N = N tiles per workgroup at the stride of the grid.
The first round also checks that every tN / P output equals the one-tile kernel's bit for bit."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplefe_amd import build, lib  # noqa: E402
lib.LIB_PATH = build.build_lib(diag=True)
from simplefe_amd import api, synth  # noqa: E402

variants = sys.argv[1:] or ["0", "1", "2", "3"]
n = 1 << 30
x = api.DeviceArray(2 * n)
x.fill_synth(synth.SEED)
cap = n // 8 + 8
y = api.DeviceArray(2 * cap)
r = api.Rs(synth.taps_cfg4(), 1, 4096, mode=lib.RS_DECIMATE, data_complex=True)
t = api.Timer()
res = {v: [] for v in variants}
ref = None
for k in range(int(os.environ.get("ROUNDS", "6")) + 1):
    for v in variants:
        os.environ.pop("SFE_TILED_TPW", None)
        os.environ["SFE_TILED_DIAG"] = "0"
        if v[0] == "t":
            os.environ["SFE_TILED_TPW"] = v[1:]
        elif v != "P":
            os.environ["SFE_TILED_TPW"] = "1"
            os.environ["SFE_TILED_DIAG"] = v
        if k == 0 and (v[0] == "t" or v in ("P", "0", "4")):
            y.zero()
            r.reset()
            r.process_stream(x, n, y, cap, 8.0)
            r.process_stream(x, n, y, cap, 8.0)      # second call: carried history
            got = y.to_numpy(2 * (n // 8))
            if ref is None:
                os.environ["SFE_TILED_TPW"] = "1"
                y.zero()
                r.reset()
                r.process_stream(x, n, y, cap, 8.0)
                r.process_stream(x, n, y, cap, 8.0)
                ref = y.to_numpy(2 * (n // 8))
                if v[0] == "t":
                    os.environ["SFE_TILED_TPW"] = v[1:]
                elif v == "P":
                    os.environ.pop("SFE_TILED_TPW", None)
            print(f"# {v}: output {'identical to' if np.array_equal(got, ref) else 'DIFFERENT from'} the one-tile kernel's")
        t.start()
        for _ in range(3):
            r.process_stream(x, n, y, cap, 8.0)
        t.stop()
        if k:
            res[v].append(t.elapsed_ms() / 3)
alg = 9.0 * n
for v in variants:
    a = np.array(res[v])
    print(f"diag {v}: median {np.median(a):.4f} ms  min {a.min():.4f}  max {a.max():.4f}  -> {alg / np.median(a) / 1e6 / 80:.1f}% of 8 TB/s")
