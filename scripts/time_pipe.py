#!/usr/bin/env python3
"""Where a pipelined host call spends its time: per-call cost of push / pull at scheduler-sized
calls, and the rate of the whole pipe (sfe_dsp_fir_pipe_*), beside one round trip per call."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplefe_amd import api, lib, synth  # noqa: E402

L = lib.load()
taps = synth.taps_cfg2()
n = 1 << 22
x = synth.synth_cf32(n)
y = np.empty_like(x)
call = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for batch in (1 << 14, 1 << 16, 1 << 18):
    f = api.Fir(taps, data_complex=True)
    p = C.c_void_p()
    api.check(L.sfe_dsp_fir_pipe_create(f._h, batch, C.byref(p)))
    taken, got = C.c_size_t(0), C.c_size_t(0)
    t_push = t_pull = 0.0
    off = prod = 0
    t0 = time.perf_counter()
    while prod < n:
        m = min(call, n - off)
        a = time.perf_counter()
        if m:
            api.check(L.sfe_dsp_pipe_push(p, x.ctypes.data + 8 * off, m, C.byref(taken)))
            off += taken.value
        b = time.perf_counter()
        api.check(L.sfe_dsp_pipe_pull(p, y.ctypes.data + 8 * prod, min(call, n - prod), 0, C.byref(got)))
        if (m == 0 or taken.value == 0) and got.value == 0:
            api.check(L.sfe_dsp_pipe_pull(p, y.ctypes.data + 8 * prod, min(call, n - prod), 1 if m else 2, C.byref(got)))
        c = time.perf_counter()
        prod += got.value
        t_push += b - a
        t_pull += c - b
    dt = time.perf_counter() - t0
    print(f"batch {batch:7d}: {n / dt / 1e6:8.1f} MS/s   push {t_push * 1e3:7.2f} ms  pull {t_pull * 1e3:7.2f} ms  of {dt * 1e3:7.2f} ms "
          f"({n // call} calls of {call} items)")
    L.sfe_dsp_pipe_destroy(p)
# the same pipe without the two host copies: items are "produced" in place in the pinned input batch (nothing is written:
# the producer's own cost is not the pipe's) and finished items are consumed in place -- what the three streams move
f = api.Fir(taps, data_complex=True)
p = C.c_void_p()
api.check(L.sfe_dsp_fir_pipe_create(f._h, 1 << 18, C.byref(p)))
buf, src, room, got = C.c_void_p(), C.c_void_p(), C.c_size_t(0), C.c_size_t(0)
n2 = 1 << 26
off = prod = 0
t0 = time.perf_counter()
while prod < n2:
    moved = 0
    if off < n2:
        api.check(L.sfe_dsp_pipe_acquire(p, C.byref(buf), C.byref(room)))
        moved = min(room.value, n2 - off)
        if moved:
            api.check(L.sfe_dsp_pipe_commit(p, moved))
            off += moved
    api.check(L.sfe_dsp_pipe_peek(p, C.byref(src), C.byref(got), 0 if moved else (1 if off < n2 else 2)))
    if got.value:
        api.check(L.sfe_dsp_pipe_release(p, got.value))
        prod += got.value
dt = time.perf_counter() - t0
print(f"acquire / commit / peek / release (no host copies), batch 262144: {n2 / dt / 1e6:8.1f} MS/s = {16 * n2 / dt / 1e9:.1f} GB/s over PCIe, both directions")
L.sfe_dsp_pipe_destroy(p)
f = api.Fir(taps, data_complex=True)
t0 = time.perf_counter()
for off in range(0, n, call):
    api.check(L.sfe_dsp_fir_process_host(f._h, x.ctypes.data + 8 * off, y.ctypes.data + 8 * off, call))
dt = time.perf_counter() - t0
print(f"one round trip per call: {n / dt / 1e6:8.1f} MS/s ({dt / (n // call) * 1e6:.1f} us per call)")
