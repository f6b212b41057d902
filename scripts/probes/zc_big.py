import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from simplefe_amd import api, lib, synth
L = lib.load()
taps = synth.taps_cfg2()
for call in (65536, 1 << 18, 1 << 20, 1 << 22):
    f = api.Fir(taps, data_complex=True)
    if os.environ.get("ZC_MAX"):
        f.set_zero_copy_max(int(os.environ["ZC_MAX"]))
    x = synth.synth_cf32(call); y = np.empty_like(x)
    fn = lambda: api.check(L.sfe_dsp_fir_process_host(f._h, x.ctypes.data, y.ctypes.data, call))
    for _ in range(5): fn()
    reps = max(5, int(2e7 // call))
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    us = (time.perf_counter() - t0) / reps * 1e6
    print(f"zc_max={os.environ.get('ZC_MAX')}: {call:8d} cf32 per call: {us:9.1f} us = {call/us:7.1f} MS/s")
