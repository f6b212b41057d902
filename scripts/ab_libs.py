#!/usr/bin/env python3
"""Interleaved A/B of the SAME call through two builds of the library in one process (each loaded
privately): ab_libs.py <libA.so> <libB.so> [real|cf32|tx10|wire|rtx10|rwire|decimate|resample|general]   -- 256-tap FIR, 2^29 real / 2^28 cf32 samples;
tx10 = cf32 in, 10-bit packed out; wire = u8 (I,Q) bytes in, 10-bit packed out; rtx10 / rwire = the same for a real stream; decimate = by 8, 64 taps, 2^30 cf32; resample = 5/3, 381 taps, 2^28 cf32; general = the same filter at rate 1.77."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplefe_amd import lib as L, synth  # noqa: E402

paths = [a for a in sys.argv[1:] if a.endswith(".so")]          # two or more builds
mode = ([a for a in sys.argv[1:] if not a.endswith(".so")] or ["real"])[0]
cplx = mode in ("cf32", "tx10", "wire")
n = 1 << (28 if cplx else 29)
libs = []
for p in paths:
    h = C.CDLL(os.path.abspath(p), mode=os.RTLD_LOCAL)
    for name, (res, args) in L.SIGNATURES.items():
        if hasattr(h, name):
            getattr(h, name).restype, getattr(h, name).argtypes = res, args
    libs.append(h)
taps = synth.taps_cfg2()
rs = mode in ("decimate", "resample", "general")
if rs:
    taps, U, rate, n = (synth.taps_cfg4(), 1, 8.0, 1 << 30) if mode == "decimate" else (synth.taps_cfg3(), 3, 1.77 if mode == "general" else float(np.float32(5) / np.float32(3)), 1 << 28)
    cap = int(n / rate) + 8192
st = []
for h in libs:
    x, y, f, t = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
    nx = (8 << 30) if mode == "decimate" else (8 << 28)
    assert h.sfe_dsp_malloc(C.byref(x), nx) == 0 and h.sfe_dsp_malloc(C.byref(y), 8 << 28) == 0
    assert h.sfe_dsp_synth_fill(x, nx // 4, synth.SEED, 0, 0, None) == 0
    if rs:
        assert h.sfe_dsp_rs_create(taps.ctypes.data, len(taps), U, 4096, 1, 1, 0, 1 if mode == "decimate" else 0, C.byref(f)) == 0
    else:
        assert h.sfe_dsp_fir_create(taps.ctypes.data, len(taps), 0, int(cplx), 1, 0, 0, C.byref(f)) == 0
    if mode in ("wire", "rwire"):
        assert h.sfe_dsp_fir_set_input_format(f, 1) == 0
    if mode in ("tx10", "wire", "rtx10", "rwire"):
        assert h.sfe_dsp_fir_set_output_format(f, 2) == 0
    assert h.sfe_dsp_timer_create(C.byref(t)) == 0
    st.append((h, x, y, f, t))
res = [[] for _ in paths]
for r in range(int(os.environ.get("ROUNDS", "10")) + 1):
    for i, (h, x, y, f, t) in enumerate(st):
        h.sfe_dsp_timer_start(t, None)
        for _ in range(5):
            if rs:
                k = C.c_size_t(0)
                assert h.sfe_dsp_rs_process_stream(f, x, n, n, y, cap, cap, rate, C.byref(k), None) == 0
            else:
                assert h.sfe_dsp_fir_process_stream(f, x, y, n, n, n, None) == 0
        h.sfe_dsp_timer_stop(t, None)
        ms = C.c_float(0)
        h.sfe_dsp_timer_elapsed_ms(t, C.byref(ms))
        if r:
            res[i].append(ms.value / 5)
# the builds must agree: three windows of the output (first, middle, last written), against the first build's
wins = []
for h, x, y, f, t in st:
    h.sfe_dsp_sync(None)
    n_out_f = 2 * (int(k.value) if rs else n) if mode not in ("tx10", "wire", "rtx10", "rwire") else 0
    got = []
    for off in ((0, n_out_f // 2 - (1 << 17), n_out_f - (1 << 18)) if n_out_f else ()):
        off -= off % 2
        buf = np.empty(1 << 18, np.float32)
        assert h.sfe_dsp_memcpy_d2h(buf.ctypes.data, C.c_void_p(y.value + 4 * off), buf.nbytes, None) == 0
        h.sfe_dsp_sync(None)
        got.append(buf)
    wins.append(got)
for p, a, g in zip(paths, res, wins):
    dev = max([float(np.max(np.abs(u - v))) for u, v in zip(g, wins[0])] or [0.0])
    print(f"{os.path.basename(p):32s} median {np.median(a):.4f} ms  min {min(a):.4f}  max {max(a):.4f}   max |y - y(first build)| {dev:.2e}")
