#!/usr/bin/env python3
"""The drop-in classes the reference's way -- resample / decimate ::process() on host pointers, block by block -- over a wide matrix of
(upsample, rate, taps per phase, blksize, block length): every call's n_out and every output BIT against the oracle (the class path is
always the exact kernels).  Prints the combinations that differ or raise."""
import itertools
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from simplefe_amd import api, synth  # noqa: E402
from oracle import binding as orc  # noqa: E402

RATES = [(1, 1.0), (1, 2.0), (1, 7.0), (1, 64.0), (1, 100.0), (1, 128.0), (1, 1000.0), (1, 2.5), (1, 1.77), (1, 100.3), (3, 5.0 / 3.0), (3, 1.77), (3, 0.77), (3, 1.0 / 3.0), (3, 40.0),
         (8, 1.0 / 8.0), (8, 0.3), (9, 10.0 / 9.0), (24, 25.0 / 24.0), (160, 147.0 / 160.0), (160, 0.5), (160, 1.77), (12, 1.0 / 12.0), (12, 1.003)]
bad, tried = {}, 0
for (U, rate), plen, B, frac_blk, mode in itertools.product(RATES, (1, 7, 32, 127), (4096, 1000, 256), (1.0, 0.37), ("resample", "decimate")):
    if plen > B or (mode == "decimate" and (rate < 1.0 or U != 1)):
        continue
    rate = float(np.float32(rate))
    taps = synth.lowpass_taps(max(U, plen * U - (U > 1)), 0.9 * min(1.0 / U, 1.0 / max(rate * U, 1.0)), gain=float(U))
    nb = max(1, int(B * frac_blk))
    x = synth.synth_f32(6 * nb, ch=tried % 50)
    tried += 1
    try:
        dut = (api.resample if mode == "resample" else api.decimate)(taps, U, B)
        ref = (orc.Resample if mode == "resample" else orc.Decimate)(taps, U, B)
        for off in range(0, len(x), nb):
            seg = x[off: off + nb]
            ol = int(np.ceil(len(seg) / rate)) + 2
            n1, o1 = dut.process(seg, ol, rate)
            n2, o2 = ref.process(seg, ol, rate)
            if n1 != n2 or not np.array_equal(o1[:n1].view(np.uint32), o2[:n2].view(np.uint32)):
                bad.setdefault("differs from the oracle", []).append((mode, U, round(rate, 4), plen, B, nb, off // nb, n1, n2))
                break
    except Exception as e:                   # noqa: BLE001
        bad.setdefault(str(e).split(": ", 1)[-1][:110], []).append((mode, U, round(rate, 4), plen, B, nb))
print(f"{tried} combinations tried, {sum(len(v) for v in bad.values())} bad")
for k, v in bad.items():
    print(f"-- {len(v):4d} x  {k}")
    for c in v[:12]:
        print("        ", c)
