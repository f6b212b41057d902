#!/usr/bin/env python3
"""Interleaved A/B timing of FIR kernel variants in ONE process on ONE device
(cdna_hip_programming.md rule 24), through the DIAGNOSTIC library libsfe_dsp_diag.so (-DSFE_DIAG:
built on demand; the product library has none of these switches).
L / M / N = T without the two middle LDS exchanges / with one multiply per twiddle / both; l / n the same of X (round 5: energy
ablations, WRONG results on purpose -- never compared for parity).  WATTS=1 also prints the in-kernel clock of each variant.
Usage: ab_fir.py T X W "T:4" "X+1" "e:8" ...   variant[:wg_per_cu][+diag bits: 1 no loads, 2 no stores]
  variant = T / X / W = the product's three data-movement variants (register loads / LDS-DMA / LDS-DMA into the wave-private layout),
            t / x / w = the same with round 2's guarded store loop, u / y = stores interleaved with the last DFT16,
            c / d / e / E / b / g / G / q / Q / p = the bare access patterns (fir_fft_diag.inc), P = the product's own dispatch.
  (Round 5 removed the round-1 kernels "4n.h", "3p", D ... with their template switches.)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplefe_amd import build, lib  # noqa: E402
lib.LIB_PATH = os.environ.get("SFE_DIAG_LIB") or build.build_lib(diag=True)      # SFE_DIAG_LIB: a saved build to compare against          # before anything loads the product library
from simplefe_amd import api, synth  # noqa: E402

variants = sys.argv[1:] or ["T", "X", "W", "e:8"]
log2n = int(os.environ.get("LOG2N", "28"))
rounds = int(os.environ.get("ROUNDS", "6"))
# ZEROS=1: the input holds zeros (what the kernel costs without the data's toggling: DESIGN.md 9); WATTS=1: after the timing, ~1.5 s of
# each variant's launches with rocm-smi's shader clock and package power read twice while they run
n = 1 << log2n
x = api.DeviceArray(2 * n)
x.fill_synth(synth.SEED)
if os.environ.get("ZEROS") == "1":
    x.zero()
y = api.DeviceArray(2 * n)
f = api.Fir(synth.taps_cfg2(), data_complex=True, algo=lib.FIR_ALGO_FFT)
t = api.Timer()
res = {v: [] for v in variants}


def select(v):
    """the environment switches of one variant string (read by the diagnostic library at every launch)"""
    vk = v.split("!")                            # X!1 = the shared rows loaded without the nontemporal hint
    os.environ["SFE_FIR_HALO_KEEP"] = vk[1] if len(vk) > 1 else "0x8001"
    vr = vk[0].split("^")                            # X^2 = each counter deals runs of 2^2 consecutive transforms
    os.environ["SFE_FIR_TQS"] = vr[1] if len(vr) > 1 else "3"
    vs = vr[0].split("~")                            # Q:4~8 = eight idle steps (~0.5 us each) between pick-up and stores
    os.environ["SFE_FIR_DELAY"] = vs[1] if len(vs) > 1 else "0"
    v_ = vs[0]
    vq = v_.split("%")                            # e:8%7 = loads/stores gated on bit 7 of the device clock
    os.environ.pop("SFE_FIR_GATE", None)
    if len(vq) > 1:
        os.environ["SFE_FIR_GATE"] = vq[1]
    vd = vq[0].split("+")
    os.environ["SFE_FIR_DIAG"] = vd[1] if len(vd) > 1 else "0"
    vg = vd[0].split("/")                        # X/16 = 16 ticket groups
    os.environ["SFE_FIR_TGROUPS"] = vg[1] if len(vg) > 1 else "8"
    vv = vg[0].split(":")
    if vv[0] == "P":                              # P = the product's own dispatch (no variant switch)
        os.environ.pop("SFE_FIR_VARIANT", None)
    else:
        os.environ["SFE_FIR_VARIANT"] = vv[0]
    if len(vv) > 1:
        os.environ["SFE_FIR_WG_PER_CU"] = vv[1]
    else:
        os.environ.pop("SFE_FIR_WG_PER_CU", None)


for r in range(rounds + 1):
    for v in variants:
        select(v)
        t.start()
        for _ in range(5):
            f.process_stream(x, y, n)
        t.stop()
        ms = t.elapsed_ms() / 5
        if r:
            res[v].append(ms)
if "P" in variants:
    v_, cal, ms = f.get_variant()
    print(f"# P = the product's dispatch: measured on this device at its first call, ran '{lib.FIR_VARIANT_NAMES.get(v_, v_)}' "
          f"(calibration medians, ms: " + ", ".join(f"{lib.FIR_VARIANT_NAMES[i]} {m:.4f}" for i, m in enumerate(ms)) + ")")
for v in variants:
    a = np.array(res[v])
    print(f"{v:8s} median {np.median(a):.4f} ms  min {a.min():.4f}  max {a.max():.4f}  -> {16.0 * n / np.median(a) / 1e6:.0f} GB/s alg, "
          f"{16.0 * n / np.median(a) / 1e6 / 80:.1f}% of 8 TB/s")

if os.environ.get("WATTS") == "1":
    import re
    import subprocess
    import time
    for v in variants:
        select(v)
        ms = float(np.median(res[v]))
        for _ in range(max(100, int(1500.0 / ms))):
            f.process_stream(x, y, n)
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
        # the clock workgroup 0 of the last of those launches saw (s_memtime / s_memrealtime stamps, diagnostic build only)
        import ctypes as C
        mhz, span = C.c_double(), C.c_double()
        clk = ""
        fn = getattr(lib.load(), "sfe_dsp_diag_fir_clock", None)
        if fn is not None and fn(C.byref(mhz), C.byref(span)) == 0:
            clk = f"   in-kernel clock {mhz.value:.0f} MHz over {span.value:.4f} ms"
        print(f"{v:8s} while running: " + " | ".join(reads) + clk)
