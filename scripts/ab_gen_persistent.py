#!/usr/bin/env python3
"""The general-rate transform kernel (2^28 cf32, 381 taps in 3 phases, rate 1.77): one workgroup per block (the product's
launch) against the persistent grid with the next block fetched ahead (csrc/diag/poly_gen_persistent.hip), its blocks dealt
statically (SFE_GEN_PERSISTENT=1) or by a device counter (=2).  DIAGNOSTIC library, one process, interleaved rounds, HIP events
over 5 back-to-back calls; the three must produce the same bits (they run the same arithmetic in the same order).

    python scripts/ab_gen_persistent.py > profiles/r04/general_persistent_ab.txt
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplefe_amd import build, lib  # noqa: E402
lib.LIB_PATH = build.build_lib(diag=True)
from simplefe_amd import api, synth  # noqa: E402

n = 1 << int(os.environ.get("LOG2N", "28"))
rate = 1.77
x = api.DeviceArray(2 * n)
x.fill_synth(synth.SEED)
cap = int(n / rate) + 8192
modes = (("one workgroup per block (product)", None), ("persistent, static deal", "1"), ("persistent, tickets", "2"))
ys = [api.DeviceArray(2 * cap) for _ in modes]
r = api.Rs(synth.taps_cfg3(), 3, 4096, mode=lib.RS_RESAMPLE, data_complex=True)
t = api.Timer()


def select(v):
    if v is None:
        os.environ.pop("SFE_GEN_PERSISTENT", None)
    else:
        os.environ["SFE_GEN_PERSISTENT"] = v


res = [[] for _ in modes]
k = 0
for rnd in range(int(os.environ.get("ROUNDS", "10")) + 1):
    for i, (name, v) in enumerate(modes):
        select(v)
        r.reset()
        t.start()
        for _ in range(5):
            k = r.process_stream(x, n, ys[i], cap, rate)
        t.stop()
        if rnd:
            res[i].append(t.elapsed_ms() / 5)
api.sync()
ref = ys[0].to_numpy(2 * k)
print(f"# 2^{int(np.log2(n))} cf32 at rate {rate}, 381 taps in 3 phases -> {k} outputs per call; ms per call, {len(res[0])} interleaved rounds of 5 calls")
for (name, v), a, y in zip(modes, res, ys):
    same = bool(np.array_equal(ref, y.to_numpy(2 * k)))
    print(f"{name:36s} median {np.median(a):.4f} ms  min {min(a):.4f}  max {max(a):.4f}   bit-identical to the product's launch: {same}")
