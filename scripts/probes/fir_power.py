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

from simplefe_amd import lib  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "fir"
n = 1 << (30 if which == "decimate" else 28)
x = api.DeviceArray(2 * n)
t = api.Timer()
if which == "fir":
    y = api.DeviceArray(2 * n)
    f = api.Fir(synth.taps_cfg2(), data_complex=True)
    call, count = (lambda: f.process_stream(x, y, n)), 3000
elif which == "decimate":
    y = api.DeviceArray(2 * (n // 8 + 64))
    r = api.Rs(synth.taps_cfg4(), 1, 4096, mode=lib.RS_DECIMATE, data_complex=True)
    call, count = (lambda: r.process_stream(x, n, y, n // 8 + 64, 8.0)), 1600
else:
    y = api.DeviceArray(2 * (n * 3 // 5 + 64))
    r = api.Rs(synth.taps_cfg3(), 3, 4096, mode=lib.RS_RESAMPLE, data_complex=True)
    call, count = (lambda: r.process_stream(x, n, y, n * 3 // 5 + 64, 5.0 / 3.0)), 3500
print("# %s" % which)


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=6).stdout
    except Exception as e:                     # noqa: BLE001
        return "rocm-smi: %s" % e
    keep = [re.sub(r"\s+", " ", l.strip()) for l in out.splitlines() if re.search(r"sclk|mclk|fclk|Power", l)]
    return " ; ".join(keep[:6])


for what in (("zeros", "data", "zeros", "data") if which == "fir" else ("zeros", "data")):
    if what == "zeros":
        x.zero()
    else:
        x.fill_synth(synth.SEED)
    for _ in range(50):
        call()
    api.sync()
    t.start()
    for _ in range(count):
        call()
    t.stop()
    reads = []
    for _ in range(3):
        time.sleep(0.25)
        reads.append(smi())
    ms = t.elapsed_ms() / count
    print("%-5s  %.4f ms per launch" % (what, ms))
    for line_ in reads:
        print("        " + line_)
    sys.stdout.flush()
