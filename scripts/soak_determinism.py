#!/usr/bin/env python3
"""Run each bulk kernel many times on the same full-size input and check that every run writes
the same bits (a missing barrier or an LDS hazard would show up as run-to-run differences)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplefe_amd import api, lib, synth  # noqa: E402

dev = torch.device("cuda:0")
L = lib.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100


def digest(t):
    v = t.view(torch.int32).to(torch.int64)
    return int(v.sum().item()), int((v * (torch.arange(v.numel(), device=dev) % 1021 + 1)).sum().item())


n = 1 << 26
x = torch.empty(2 * n, dtype=torch.float32, device=dev)
lib.check(L.sfe_dsp_synth_fill(x.data_ptr(), 2 * n, synth.SEED, 0, 0, None))
bad = 0
for name, make, out_n in (
        ("fir 256 taps", lambda: api.Fir(synth.taps_cfg2(), data_complex=True), n),
        ("resample 5/3 fft", lambda: api.Rs(synth.taps_cfg3(), 3, 4096, mode=lib.RS_RESAMPLE, data_complex=True), n * 3 // 5 + 8),
        ("resample 5/3 real fft", lambda: api.Rs(synth.taps_cfg3(), 3, 4096, mode=lib.RS_RESAMPLE, data_complex=False), n * 3 // 5 + 8),
        ("decimate 8", lambda: api.Rs(synth.taps_cfg4(), 1, 4096, mode=lib.RS_DECIMATE, data_complex=True), n // 8 + 8)):
    obj = make()
    y = torch.zeros(2 * out_n, dtype=torch.float32, device=dev)
    first = None
    for r in range(reps):
        obj.reset()
        y.zero_()
        if name.startswith("fir"):
            obj.process_stream(x.data_ptr(), y.data_ptr(), n)
        else:
            rate = 8.0 if "decimate" in name else 5.0 / 3.0
            obj.process_stream(x.data_ptr(), n, y.data_ptr(), out_n, rate)
        torch.cuda.synchronize()
        d = digest(y)
        if first is None:
            first = d
        elif d != first:
            bad += 1
            print(f"{name}: run {r} differs: {d} vs {first}", flush=True)
            break
    print(f"{name}: {reps} runs, digest {first}", flush=True)
sys.exit(1 if bad else 0)
