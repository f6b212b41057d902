import os, sys
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from simplefe_amd import api, lib, synth
n = 1 << 24
x = api.DeviceArray(2 * n); x.fill_synth(synth.SEED)
t = api.Timer()
for B in (1000, 4096, 8192, 16384, 65536):
    for U, rate, plen, exact in ((3, 1.77, 127, False), (3, 1.77, 127, True), (1, 2.5, 32, False), (3, 5.0/3.0, 127, True)):
        taps = synth.lowpass_taps(plen*U, 0.3/U, gain=float(U))
        for cplx in (True, False):
            w = 2 if cplx else 1
            m = (n // B) * B
            cap = int(m/rate) + 4*(m//B) + 4096
            y = api.DeviceArray(w*cap)
            r = api.Rs(taps, U, B, mode=lib.RS_RESAMPLE, data_complex=cplx)
            r.set_exact(exact)
            try:
                for _ in range(2): k = r.process_stream(x, m, y, cap, float(np.float32(rate)))
                t.start(); r.process_stream(x, m, y, cap, float(np.float32(rate))); t.stop()
                print(f"blksize {B:6d} U {U} rate {rate:.3f} plen {plen} exact {exact!s:5s} {'cplx' if cplx else 'real'}: {t.elapsed_ms():9.3f} ms", flush=True)
            except Exception as e:
                print(f"blksize {B:6d} U {U} rate {rate:.3f} plen {plen} exact {exact} {'cplx' if cplx else 'real'}: ERROR {e}", flush=True)
            r.close(); y.free()
