"""Transform-domain resample kernel against the direct (tiled, exact) kernel and the oracle,
plus a timing of both on the headline shape.  Run on the GPU box."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplefe_amd import api, lib, synth  # noqa: E402


def run(U, S, n_taps, n, nch=1, chunk=None, fft=True, exact=False, seed=0):
    rng = np.random.default_rng(seed)
    taps = (rng.standard_normal(n_taps) / np.sqrt(n_taps)).astype(np.float32)
    x = np.stack([synth.synth_f32(2 * n, ch=seed * 4 + c) for c in range(nch)])
    r = api.Rs(taps, U, 4096, mode=lib.RS_RESAMPLE, data_complex=True, n_channels=nch,
               algo=lib.RS_ALGO_FFT if fft else lib.RS_ALGO_DIRECT)
    r.set_exact(exact)
    y = r.resample_array(x, float(np.float32(S) / np.float32(U)), chunk=chunk)
    r.close()
    return y


CASES = [(3, 5, 381, 100000, 1, None), (3, 5, 381, 250001, 2, 65536), (2, 3, 200, 90000, 1, None),
         (2, 5, 301, 90000, 3, 40000), (3, 4, 255, 120000, 1, None), (1, 2, 128, 80000, 1, 30000),
         (1, 3, 200, 80000, 1, None), (1, 4, 256, 80000, 2, None), (1, 5, 333, 80000, 1, 20001),
         (3, 5, 30, 60000, 1, None), (3, 5, 1000, 200000, 1, 70000), (6, 10, 762, 100000, 1, None)]


def main():
    bad = 0
    cases = [] if "--time-only" in sys.argv else CASES
    for (U, S, nt, n, nch, chunk) in cases:
        a = run(U, S, nt, n, nch, chunk, fft=True)
        b = run(U, S, nt, n, nch, chunk, fft=False, exact=True)
        ok = a.shape == b.shape
        err = float(np.sqrt(np.sum((a.astype(np.float64) - b) ** 2) / np.sum(b.astype(np.float64) ** 2))) if ok else -1
        print(f"U={U} S={S} taps={nt} n={n} ch={nch} chunk={chunk}: shape {a.shape} vs {b.shape} rel-rms {err:.3e}", flush=True)
        if not ok or not err < 1e-5:
            bad += 1
            if ok:
                d = np.abs(a.astype(np.float64) - b).reshape(nch, -1, 2).sum(axis=2)
                w = np.argwhere(d > 1e-3)
                print("   first bad:", w[:5].tolist(), " last bad:", w[-3:].tolist(), " count", len(w))
    # timing, headline shape
    n = 1 << 28
    taps = synth.taps_cfg3()
    d_in = api.DeviceArray(2 * n)
    lib.check(lib.load().sfe_dsp_synth_fill(d_in.ptr, 2 * n, 1, 0, 0, None))
    cap = n * 3 // 5 + 16
    d_out = api.DeviceArray(2 * cap)
    for fft in (False, True, False, True):
        r = api.Rs(taps, 3, 4096, mode=lib.RS_RESAMPLE, data_complex=True, algo=lib.RS_ALGO_FFT if fft else lib.RS_ALGO_DIRECT)
        ts = []
        for it in range(60):
            r.reset()
            tm = api.Timer()
            tm.start()
            k = r.process_stream(d_in, n, d_out, cap, 5.0 / 3.0)
            tm.stop()
            ts.append(tm.elapsed_ms())
        print(f"fft={fft}: k={k} ms first {ts[0]:.3f} mean[10:] {sum(ts[10:]) / len(ts[10:]):.4f} min {min(ts):.3f}", flush=True)
        r.close()
    # real data: 2^29 float32 samples (the same bytes), two real segments per transform
    nr = 1 << 29
    capr = nr * 3 // 5 + 16
    for fft in (False, True, False, True):
        r = api.Rs(taps, 3, 4096, mode=lib.RS_RESAMPLE, data_complex=False, algo=lib.RS_ALGO_FFT if fft else lib.RS_ALGO_DIRECT)
        ts = []
        for it in range(60):
            r.reset()
            tm = api.Timer()
            tm.start()
            k = r.process_stream(d_in, nr, d_out, capr, 5.0 / 3.0)
            tm.stop()
            ts.append(tm.elapsed_ms())
        print(f"real data fft={fft}: k={k} ms first {ts[0]:.3f} mean[10:] {sum(ts[10:]) / len(ts[10:]):.4f} min {min(ts):.3f}", flush=True)
        r.close()
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
