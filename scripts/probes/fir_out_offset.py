#!/usr/bin/env python3
"""Does the relative placement of the input and output streams matter?  The product FIR kernel on the
headline shape with the output buffer shifted by 0 .. 1 MiB against its allocation (the physical page
mapping keeps offsets below the page size, so this moves the output's channel phase against the
input's).  Interleaved rounds, medians."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from simplefe_amd import api, lib, synth  # noqa: E402

n = 1 << 28
x = api.DeviceArray(2 * n)
x.fill_synth(synth.SEED)
y = api.DeviceArray(2 * n + (1 << 19))
f = api.Fir(synth.taps_cfg2(), data_complex=True)
t = api.Timer()
offs = [0, 256, 1024, 4096, 16384, 65536, 131072, 262144, 524288, 1048576]
res = {o: [] for o in offs}
for r in range(7):
    for o in offs:
        t.start()
        for _ in range(5):
            f.process_stream(x, y.ptr + o, n)
        t.stop()
        if r:
            res[o].append(t.elapsed_ms() / 5)
for o in offs:
    a = np.array(res[o])
    print(f"output shifted by {o:8d} B: median {np.median(a):.4f} ms  min {a.min():.4f}  max {a.max():.4f}")
