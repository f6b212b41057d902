import os, sys
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from simplefe_amd import api, lib, synth
t = api.Timer()
x = api.DeviceArray(2 << 28); x.fill_synth(synth.SEED)
for name, taps in (("cfg3", synth.taps_cfg3()), ("lowpass381", synth.lowpass_taps(381, 0.18, gain=3.0))):
    for log2n in (26, 28):
        for cplx in (True, False):
            n = (1 << log2n) * (1 if cplx else 2)
            w = 2 if cplx else 1
            cap = n * 3 // 5 + 4096
            y = api.DeviceArray(w * cap)
            r = api.Rs(taps, 3, 4096, mode=lib.RS_RESAMPLE, data_complex=cplx)
            for rate in (5.0 / 3.0, float(np.float32(5.0 / 3.0))):
                for _ in range(3): k = r.process_stream(x, n, y, cap, rate)
                t.start(); r.process_stream(x, n, y, cap, rate); t.stop()
                print(name, log2n, "cplx" if cplx else "real", repr(rate), k, f"{t.elapsed_ms():.4f} ms", flush=True)
            r.close(); y.free()
