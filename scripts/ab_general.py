#!/usr/bin/env python3
"""The general-rate transform kernel with and without its whole prologue (round 5, DIAGNOSTIC library): SFE_GEN_NOPRO=1 replaces step 0 -- call
records, runs, bound searches, the table of (position, mu) -- by a synthetic table (WRONG results on purpose): the most a table that arrived
ready-made (built by a separate streaming kernel, landing by LDS-DMA) could save.  One process, interleaved rounds, HIP events.
    python scripts/ab_general.py > profiles/r05/general_rate_prologue_ablation.txt"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplefe_amd import build, lib  # noqa: E402
lib.LIB_PATH = build.build_lib(diag=True)
from simplefe_amd import api, synth  # noqa: E402

n = 1 << 28
rate = float(np.float32(1.77))
x = api.DeviceArray(2 * n)
x.fill_synth(synth.SEED)
cap = int(n / rate) + 131072
y = api.DeviceArray(2 * cap)
r = api.Rs(synth.taps_cfg3(), 3, 4096, mode=lib.RS_RESAMPLE, data_complex=True)
t = api.Timer()
res = {"product": [], "no prologue": []}
for rnd in range(int(os.environ.get("ROUNDS", "8")) + 1):
    for name, env in (("product", "0"), ("no prologue", "1")):
        os.environ["SFE_GEN_NOPRO"] = env
        t.start()
        for _ in range(3):
            r.process_stream(x, n, y, cap, rate)
        t.stop()
        if rnd:
            res[name].append(t.elapsed_ms() / 3)
print("# scripts/ab_general.py: 2^28 cf32, rate 1.77, 381 taps in 3 phases; HIP events around 3 back-to-back calls (host planning overlapped), interleaved")
for k, v in res.items():
    print(f"{k:12s} median {np.median(v):.4f} ms  min {min(v):.4f}  max {max(v):.4f}")
