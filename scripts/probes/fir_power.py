#!/usr/bin/env python3
"""The FIR on zeros and on data, with the chip's own telemetry beside it: ~2.5 s of back-to-back launches queued, rocm-smi's shader
clock and package power read a few times while they run.  (Is the 14 % the clock?)"""
import os
import re
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from simplefe_amd import api, synth  # noqa: E402

n = 1 << 28
x, y = api.DeviceArray(2 * n), api.DeviceArray(2 * n)
f = api.Fir(synth.taps_cfg2(), data_complex=True)
t = api.Timer()


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=6).stdout
    except Exception as e:                     # noqa: BLE001
        return "rocm-smi: %s" % e
    keep = [re.sub(r"\s+", " ", l.strip()) for l in out.splitlines() if re.search(r"sclk|mclk|fclk|Power", l)]
    return " ; ".join(keep[:6])


for what in ("zeros", "data", "zeros", "data"):
    if what == "zeros":
        x.zero()
    else:
        x.fill_synth(synth.SEED)
    for _ in range(50):
        f.process_stream(x, y, n)
    api.sync()
    t.start()
    for _ in range(3000):
        f.process_stream(x, y, n)
    t.stop()
    reads = []
    for _ in range(3):
        time.sleep(0.25)
        reads.append(smi())
    ms = t.elapsed_ms() / 3000
    print("%-5s  %.4f ms per launch" % (what, ms))
    for r in reads:
        print("        " + r)
    sys.stdout.flush()
