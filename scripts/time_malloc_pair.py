#!/usr/bin/env python3
"""sfe_dsp_malloc_pair against two plain allocations, in fresh processes (run several): the decimator (decimate by 8, 64 taps,
2^30 cf32 -> 2^27) and the bare 8 : 1 mix on each pair.   python scripts/time_malloc_pair.py [processes]"""
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplefe_amd import api, lib, synth  # noqa: E402

N = 1 << 30
CAP = N // 8 + 8


def one():
    L = lib.load()
    r = api.Rs(synth.taps_cfg4(), 1, 4096, mode=lib.RS_DECIMATE, data_complex=True)
    t = api.Timer()

    def dec(pin, pout):
        for _ in range(6):
            r.process_stream(pin, N, pout, CAP, 8.0)
        v = []
        for _ in range(7):
            t.start()
            for _ in range(3):
                r.process_stream(pin, N, pout, CAP, 8.0)
            t.stop()
            v.append(t.elapsed_ms() / 3)
        return float(np.median(v))

    a, b = C.c_void_p(), C.c_void_p()
    api.check(L.sfe_dsp_malloc(C.byref(a), 8 * N))
    api.check(L.sfe_dsp_malloc(C.byref(b), 8 * CAP))
    api.check(L.sfe_dsp_synth_fill(a, 2 * N, synth.SEED, 0, 0, None))
    for _ in range(40):
        r.process_stream(a.value, N, b.value, CAP, 8.0)
    ms = C.c_float()
    api.check(L.sfe_dsp_probe_pair(a, 8 * N, b, 8 * CAP, C.byref(ms)))
    plain = (dec(a.value, b.value), ms.value)
    api.check(L.sfe_dsp_free(a))
    api.check(L.sfe_dsp_free(b))
    kept, worst = C.c_float(), C.c_float()
    t0 = time.perf_counter()
    api.check(L.sfe_dsp_malloc_pair(8 * N, 8 * CAP, 4, C.byref(a), C.byref(b), C.byref(kept), C.byref(worst)))
    took = time.perf_counter() - t0
    api.check(L.sfe_dsp_synth_fill(a, 2 * N, synth.SEED, 0, 0, None))
    built = dec(a.value, b.value)
    y = np.empty(4096, np.float32)
    api.check(L.sfe_dsp_memcpy_d2h(y.ctypes.data, b, y.nbytes, None))
    api.sync()
    api.check(L.sfe_dsp_free(a))
    api.check(L.sfe_dsp_free(b))
    print(f"two plain allocations: decimate {plain[0]:.4f} ms (bare mix {plain[1]:.4f})   sfe_dsp_malloc_pair ({took:.2f} s): decimate {built:.4f} ms "
          f"(bare mix {kept.value:.4f}, a pair of one class ~{worst.value:.4f})   output finite: {bool(np.isfinite(y).all())}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        one()
    else:
        for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "one"], capture_output=True, text=True, timeout=300)
            print(f"process {i}: " + ((out.stdout.strip().splitlines() or ["(no output) " + out.stderr[-300:]])[-1]), flush=True)
