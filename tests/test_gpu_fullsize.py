"""BASELINE.json sizes on the GPU, checked through size-independent properties and oracle
windows (the oracle cannot chew 2^28 samples in test time).  `-m gpu`.

  * windows: seeded spans of the device-generated stream are regenerated on the host
    (synth.py is the bit-exact twin of the device generator) and run through the oracle with
    the filter's history in front -- valid because the filters have finite memory.
  * linearity: F(a*x1 + x2) == a*F(x1) + F(x2) to rounding, on the whole stream.
  * decimate/8 of a stream == every 8th sample of the 64-tap FIR of that stream (same taps).
"""
import numpy as np
import pytest

from simplefe_amd import synth

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def api():
    from simplefe_amd import api as a
    return a


@pytest.fixture(scope="module")
def L():
    from simplefe_amd import lib
    return lib


def _window(d, start, count):
    return d.to_numpy(2 * count, offset=2 * start)


def test_fir_cfg2_full_size_windows_and_seams(api, L, orc):
    """configs[1]: 256 taps, 2^28 cf32 samples, one pass; 8 windows incl. start, end and
    transform seams (multiples of 3840)."""
    n = 1 << 28
    taps = synth.taps_cfg2()
    x = api.DeviceArray(2 * n)
    x.fill_synth(synth.SEED)
    y = api.DeviceArray(2 * n)
    f = api.Fir(taps, data_complex=True, algo=L.FIR_ALGO_FFT)
    f.process_stream(x, y, n)
    W, H = 1 << 13, 255
    for s0 in (0, 3840 - 64, 3840 * 1000 - 4096, n // 3, n // 2 + 1234, 3840 * 50000 + 1, n - 3 * W, n - W):
        lo = max(0, s0 - H)
        seg = synth.synth_cf32(s0 + W - lo, first_sample=lo)
        got = _window(y, s0, W)
        for part in (0, 1):
            ref = orc.Blkconv(taps, 4096).stream(np.ascontiguousarray(seg[part::2]))[s0 - lo:]
            assert synth.rel_rms(got[part::2], ref) <= TOL, (s0, part)
    x.free()
    y.free()


def test_fir_linearity_full_stream(api, L):
    """F(2 x1 + x2) == 2 F(x1) + F(x2) on 2^24 samples (power-of-two scale: exact inputs)."""
    n = 1 << 24
    taps = synth.taps_cfg2()
    x1 = synth.synth_cf32(n, ch=1)
    x2 = synth.synth_cf32(n, ch=2)
    f = api.Fir(taps, data_complex=True, algo=L.FIR_ALGO_FFT)
    y1 = f.filter(x1)[0]
    f.reset()
    y2 = f.filter(x2)[0]
    f.reset()
    y12 = f.filter(np.float32(2.0) * x1 + x2)[0]
    assert synth.rel_rms(y12, 2.0 * y1.astype(np.float64) + y2) <= 2e-6


def test_decimate_cfg4_full_size(api, L, orc):
    """configs[3]: decimate by 8, 64 taps, 2^30 cf32 in (8 GiB) -> 2^27 out; exact mode is
    bit-exact with the oracle on windows at the start, middle and end."""
    n = 1 << 30
    taps = synth.taps_cfg4()
    x = api.DeviceArray(2 * n)
    x.fill_synth(synth.SEED)
    r = api.Rs(taps, 1, 4096, mode=L.RS_DECIMATE, data_complex=True)
    r.set_exact(True)
    cap = n // 8 + 8
    y = api.DeviceArray(2 * cap)
    k = r.process_stream(x, n, y, cap, 8.0)
    assert k == n // 8
    W, H = 4096, 72
    for k0 in (0, 12345, k // 2, k - W):
        a0 = max(0, 8 * k0 - 8 * 16)          # multiple of 8 >= history before output k0
        seg = synth.synth_cf32(8 * (k0 + W) - a0, first_sample=a0)
        got = _window(y, k0, W)
        j0 = k0 - a0 // 8
        for part in (0, 1):
            ref, _ = orc.Decimate(taps, 1, 4096).stream(np.ascontiguousarray(seg[part::2]), 8.0)
            if a0 == 0:
                assert np.array_equal(got[part::2], ref[j0:j0 + W]), (k0, part)
            else:       # oracle started mid-stream: its first H/8 outputs lack history
                assert np.array_equal(got[part::2][16:], ref[j0 + 16:j0 + W]), (k0, part)
    x.free()
    y.free()


def test_resample_cfg3_full_size(api, L, orc):
    """configs[2]: 5/3 resample, 381-tap prototype, 2^28 cf32 in -> 161 061 273(+1 pending)."""
    n = 1 << 28
    taps = synth.taps_cfg3()
    x = api.DeviceArray(2 * n)
    x.fill_synth(synth.SEED)
    r = api.Rs(taps, 3, 4096, mode=L.RS_RESAMPLE, data_complex=True)
    r.set_exact(True)
    cap = int(n * 3 / 5) + 8
    y = api.DeviceArray(2 * cap)
    k = r.process_stream(x, n, y, cap, 5.0 / 3.0)
    assert k in (161061273, 161061274)     # ceil(3*2^28/5), the last one may be a pending leftover
    W = 3000
    for k0 in (0, 3 * 1000, 3 * (k // 6), k - W - 2):
        k0 -= k0 % 3                          # start of a 3-output period: input index 5*k0/3
        nin0 = 5 * k0 // 3
        a0 = max(0, nin0 - 5 * 30)            # multiple of 5, >= 150 samples (> 127 taps) earlier
        j0 = k0 - 3 * a0 // 5
        seg = synth.synth_cf32(5 * (k0 + W) // 3 + 4 - a0, first_sample=a0)
        got = _window(y, k0, W)
        for part in (0, 1):
            ref, _ = orc.Resample(taps, 3, 4096).stream(np.ascontiguousarray(seg[part::2]), 5.0 / 3.0)
            m = min(W, len(ref) - j0)
            assert np.array_equal(got[part::2][:m], ref[j0:j0 + m]), (k0, part)
    x.free()
    y.free()


def test_decimate_equals_subsampled_fir(api, L):
    """Cross-kernel identity at 2^22: decimate/8 == FIR (direct kernel) then keep every 8th."""
    n = 1 << 22
    taps = synth.taps_cfg4()
    x = synth.synth_cf32(n)
    full = api.Fir(taps, data_complex=True, algo=L.FIR_ALGO_DIRECT).filter(x)[0]
    r = api.Rs(taps, 1, 4096, mode=L.RS_DECIMATE, data_complex=True)
    dec = r.resample_array(x, 8.0)[0]
    ref = full.reshape(-1, 2)[::8].reshape(-1)
    assert synth.rel_rms(dec, ref[: len(dec)]) <= 1e-6


def test_multichannel_cfg5_shape(api, L, orc):
    """configs[4] shape on one GPU: 8 channels (one rank's share of 64 over 8 GPUs) x 2^22."""
    nch, n = 8, 1 << 22
    taps = synth.taps_cfg2()
    x = api.DeviceArray(2 * n * nch)
    for c in range(nch):
        x.fill_synth(synth.SEED, channel=40 + c, n_floats=2 * n, offset=2 * n * c)
    y = api.DeviceArray(2 * n * nch)
    f = api.Fir(taps, data_complex=True, n_channels=nch, algo=L.FIR_ALGO_FFT)
    f.process_stream(x, y, n)
    W, H = 4096, 255
    for c in (0, 3, 7):
        for s0 in (0, n - W):
            lo = max(0, s0 - H)
            seg = synth.synth_cf32(s0 + W - lo, ch=40 + c, first_sample=lo)
            got = y.to_numpy(2 * W, offset=2 * (n * c + s0))
            ref = orc.Blkconv(taps, 4096).stream(np.ascontiguousarray(seg[0::2]))[s0 - lo:]
            assert synth.rel_rms(got[0::2], ref) <= TOL, (c, s0)


def test_bulk_kernels_are_deterministic_run_to_run():
    """scripts/soak_determinism.py: every bulk kernel (FIR, transform-domain resample complex and
    real, decimate) writes the same bits on repeated runs over a 2^26-sample stream -- a missing
    barrier or an LDS hazard between passes would show as run-to-run differences."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "soak_determinism.py"), "12"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("12 runs") == 4, r.stdout
