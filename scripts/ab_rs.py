#!/usr/bin/env python3
"""Interleaved A/B of the transform-domain resample kernel's variants (5/3, 381 taps, 2^28 cf32) in
ONE process through the DIAGNOSTIC library.  Usage: ab_rs.py s t l L
  s = fixed-stride walk (round 1)   t = passes drawn from work counters (product, single channel)
  l = t + next pass requested after S3   L = s + late request
  e = t without the in-wave exchange, z = t without the S0 scatter writes, Z = neither (ablations: WRONG results), w = t at three
  workgroups per CU, y = z at three per CU, b = round 2's exec-masked epilogue (poly_fft.hip has the full list)
ZEROS=1: the input holds zeros.  WATTS=1 (round 5): after the timing ~1.5 s of each variant's launches with rocm-smi's shader clock and
package power read twice while they run."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplefe_amd import build, lib  # noqa: E402
lib.LIB_PATH = os.environ.get("SFE_DIAG_LIB") or build.build_lib(diag=True)      # SFE_DIAG_LIB: a saved build to compare against
from simplefe_amd import api, synth  # noqa: E402

variants = sys.argv[1:] or ["s", "t"]
rounds = int(os.environ.get("ROUNDS", "8"))
n = 1 << 28
x = api.DeviceArray(2 * n)
x.fill_synth(synth.SEED)
if os.environ.get("ZEROS") == "1":
    x.zero()
cap = int(n * 3 / 5) + 8
y = api.DeviceArray(2 * cap)
r = api.Rs(synth.taps_cfg3(), 3, 4096, mode=lib.RS_RESAMPLE, data_complex=True)
t = api.Timer()
res = {v: [] for v in variants}


def select(v):
    """the environment switches of one variant string (read by the diagnostic library at every launch)"""
    vk = v.split("!")                                  # t^3!1 = first / last staged rows without the nontemporal hint
    os.environ["SFE_RS_HALO_KEEP"] = vk[1] if len(vk) > 1 else "1"
    vr = vk[0].split("^")                              # t^3 = a counter deals runs of 2^3 consecutive passes
    os.environ["SFE_RS_TQS"] = vr[1] if len(vr) > 1 else "3"
    vv = vr[0].split(":")                              # t:8 = grid of 8 x the resident workgroups
    os.environ["SFE_RS_VARIANT"] = vv[0]
    os.environ["SFE_RS_WG_FACTOR"] = vv[1] if len(vv) > 1 else ("1" if vv[0] in ("t", "x", "l", "h", "w", "e", "z", "Z", "y", "b", "p", "q") else "2")


for k in range(rounds + 1):
    for v in variants:
        select(v)
        t.start()
        for _ in range(5):
            r.process_stream(x, n, y, cap, 5.0 / 3.0)
        t.stop()
        if k:
            res[v].append(t.elapsed_ms() / 5)
alg = 8.0 * n + 8.0 * (3 * n // 5)
for v in variants:
    a = np.array(res[v])
    print(f"{v:4s} median {np.median(a):.4f} ms  min {a.min():.4f}  max {a.max():.4f}  -> {alg / np.median(a) / 1e6 / 80:.1f}% of 8 TB/s")

if os.environ.get("WATTS") == "1":
    import re
    import subprocess
    import time
    for v in variants:
        select(v)
        ms = float(np.median(res[v]))
        for _ in range(max(100, int(1500.0 / ms))):
            r.process_stream(x, n, y, cap, 5.0 / 3.0)
        reads = []
        for _ in range(2):
            time.sleep(0.3)
            try:
                txt = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=6).stdout
            except Exception as e:                  # noqa: BLE001
                txt = str(e)
            sclk = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", txt)
            watt = re.search(r"Package Power \(W\): ([0-9.]+)", txt)
            reads.append("%s MHz %s W" % (sclk.group(1) if sclk else "?", watt.group(1) if watt else "?"))
        api.sync()
        print(f"{v:4s} while running: " + " | ".join(reads))
