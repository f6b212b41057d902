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


def test_multichannel_cfg5_as_written_one_gpu(api, L, orc):
    """configs[4] as SURVEY 8(d) fixes it for one GPU: 64 channels x 2^24 cf32 (8 GiB in, 8 GiB
    out), shared 256-tap filter, per-channel seed = channel.  EVERY channel, I and Q, first and
    last window against the oracle (blkconv.cxx:77-110 law), plus one window per channel at a
    transform seam that differs per channel."""
    nch, n = 64, 1 << 24
    taps = synth.taps_cfg2()
    x = api.DeviceArray(2 * n * nch)
    for c in range(nch):
        x.fill_synth(synth.SEED, channel=c, n_floats=2 * n, offset=2 * n * c)
    y = api.DeviceArray(2 * n * nch)
    f = api.Fir(taps, data_complex=True, n_channels=nch, algo=L.FIR_ALGO_FFT)
    f.process_stream(x, y, n)
    W, H = 4096, 255
    worst = 0.0
    for c in range(nch):
        for s0 in (0, 3840 * (100 + 61 * c) - 2000, n - W):
            lo = max(0, s0 - H)
            seg = synth.synth_cf32(s0 + W - lo, ch=c, first_sample=lo)
            got = y.to_numpy(2 * W, offset=2 * (n * c + s0))
            for part in (0, 1):
                ref = orc.Blkconv(taps, 4096).stream(np.ascontiguousarray(seg[part::2]))[s0 - lo:]
                e = synth.rel_rms(got[part::2], ref)
                worst = max(worst, e)
                assert e <= TOL, (c, s0, part, e)
    # second call on the same handle: every channel's carried history (last 256 samples of call 1)
    f.process_stream(x, y, n)
    for c in (0, 31, 63):
        seg = np.concatenate([synth.synth_cf32(H, ch=c, first_sample=n - H), synth.synth_cf32(W, ch=c)])
        got = y.to_numpy(2 * W, offset=2 * n * c)
        for part in (0, 1):
            ref = orc.Blkconv(taps, 4096).stream(np.ascontiguousarray(seg[part::2]))[H:]
            assert synth.rel_rms(got[part::2], ref) <= TOL, (c, part)
    x.free()
    y.free()


def _rs_windows_vs_oracle(orc, cls, y, taps, U, S, k0s, W, n_in, exact, hist_samples):
    """Outputs [k0, k0+W) of an integer-step resampler (output k sits at upsampled position
    k*S, resample.cxx:125-148 with mu == 0) against an oracle object started `a0` input samples
    into the stream, a0*U a multiple of S so the phase sequence lines up; the first outputs of the
    restarted oracle lack history and are skipped (finite memory)."""
    g = int(np.gcd(S, U))
    per = S // g                                  # input samples per phase period
    rate = float(np.float32(S) / np.float32(U))
    worst = 0.0
    for k0 in k0s:
        nin0 = (k0 * S) // U
        a0 = max(0, ((nin0 - hist_samples - per) // per) * per)
        j0 = k0 - a0 * U // S
        n_span = min(((k0 + W) * S) // U + 2 - a0, n_in - a0)
        seg = synth.synth_cf32(n_span, first_sample=a0)
        got = y.to_numpy(2 * W, offset=2 * k0)
        for part in (0, 1):
            ref, _ = getattr(orc, cls)(taps, U, 4096).stream(np.ascontiguousarray(seg[part::2]), rate)
            m = min(W, len(ref) - j0)
            assert m > W // 2, (k0, m)
            if exact:
                assert np.array_equal(got[part::2][:m], ref[j0:j0 + m]), (k0, part)
            else:
                e = synth.rel_rms(got[part::2][:m], ref[j0:j0 + m])
                worst = max(worst, e)
                assert e <= TOL, (k0, part, e)
    return worst


def test_resample_cfg3_full_size_default_kernel(api, L, orc, monkeypatch):
    """configs[2] through the kernel bench.py times: DEFAULT mode (no set_exact) at 2^28 cf32 must
    dispatch the transform-domain kernel poly_fft256<5,3,2> (asserted: its bits differ from the
    exact-mode direct kernel's on the same stream) and stay within 1e-5 rel-RMS of the oracle
    (resample.cxx:100-148) at the stream start, at pass seams (a pass = 2 segments x 231 low-rate
    points = 2310 inputs = 1386 outputs), in the middle, where a persistent workgroup takes
    its second pass, and at the end (input byte offsets just below 2^31)."""
    n = 1 << 28
    taps = synth.taps_cfg3()
    x = api.DeviceArray(2 * n)
    x.fill_synth(synth.SEED)
    cap = int(n * 3 / 5) + 8
    y = api.DeviceArray(2 * cap)
    r = api.Rs(taps, 3, 4096, mode=L.RS_RESAMPLE, data_complex=True)
    k = r.process_stream(x, n, y, cap, 5.0 / 3.0)
    assert k in (161061273, 161061274)
    W = 4200                                       # three passes and a bit
    P = 1386
    k0s = [0, P - 100, P * 2048 - 700, P * 4096 * 3 + 5, 3 * (k // 6), P * 100000 + 693, k - 2 * W, k - W - 2]
    k0s = [k0 - k0 % 3 for k0 in k0s]
    worst = _rs_windows_vs_oracle(orc, "Resample", y, taps, 3, 5, k0s, W, n, exact=False, hist_samples=150)
    # the exact-mode (direct) kernel on the same stream: bit-exact with the oracle, and NOT the
    # bits the default produced -> the default really was a different (the transform) kernel
    y2 = api.DeviceArray(2 * cap)
    r2 = api.Rs(taps, 3, 4096, mode=L.RS_RESAMPLE, data_complex=True)
    r2.set_exact(True)
    assert r2.process_stream(x, n, y2, cap, 5.0 / 3.0) == k
    a, b = y.to_numpy(2 * W, offset=2 * k0s[3]), y2.to_numpy(2 * W, offset=2 * k0s[3])
    assert not np.array_equal(a, b)
    assert synth.rel_rms(a, b) <= TOL
    _rs_windows_vs_oracle(orc, "Resample", y2, taps, 3, 5, k0s[:2] + k0s[-1:], W, n, exact=True, hist_samples=150)
    # second call on the same handle (carried history + time state) through the default kernel
    k2 = r.process_stream(x, n, y, cap, 5.0 / 3.0)
    assert k + k2 in (322122547, 322122548)         # ceil(3 * 2^29 / 5) or one pending
    print(f"cfg3 default kernel worst rel-RMS {worst:.3e}")
    for d in (x, y, y2):
        d.free()


def test_decimate_cfg4_full_size_default_kernel(api, L, orc):
    """configs[3] through the kernel bench.py times: DEFAULT mode (fused multiply-add
    poly_tiled<8,1>) at 2^30 cf32 in (8 GiB).  Windows at the start, at workgroup-tile seams
    (512 outputs), around input byte offsets 2^31, 2^32 and 2^33 - eps (64-bit indexing), middle,
    end; rel-RMS <= 1e-5 vs the oracle (decimate.cxx:96-140), and not the exact kernel's bits."""
    n = 1 << 30
    taps = synth.taps_cfg4()
    x = api.DeviceArray(2 * n)
    x.fill_synth(synth.SEED)
    cap = n // 8 + 8
    y = api.DeviceArray(2 * cap)
    r = api.Rs(taps, 1, 4096, mode=L.RS_DECIMATE, data_complex=True)
    k = r.process_stream(x, n, y, cap, 8.0)
    assert k == n // 8
    W = 4096
    k0s = [0, 512 - 40, 512 * 2048 * 7 - 300, (1 << 25) - 2000, (1 << 26) - 2000, k // 2 + 17, k - 2 * W, k - W]
    worst = _rs_windows_vs_oracle(orc, "Decimate", y, taps, 1, 8, k0s, W, n, exact=False, hist_samples=80)
    r2 = api.Rs(taps, 1, 4096, mode=L.RS_DECIMATE, data_complex=True)
    r2.set_exact(True)
    y2 = api.DeviceArray(2 * cap)
    assert r2.process_stream(x, n, y2, cap, 8.0) == k
    a, b = y.to_numpy(2 * W, offset=2 * k0s[4]), y2.to_numpy(2 * W, offset=2 * k0s[4])
    assert not np.array_equal(a, b) and synth.rel_rms(a, b) <= TOL
    print(f"cfg4 default kernel worst rel-RMS {worst:.3e}")
    for d in (x, y, y2):
        d.free()


def test_fir_cfg2_complex_taps_full_size(api, L, orc):
    """configs[1] read with complex taps (four real blkconv passes in the oracle,
    one complex pass on the GPU) at 2^28 cf32; windows at the start, a transform seam, the end."""
    n = 1 << 28
    tr, ti = synth.complex_taps(256, 0.2)
    x = api.DeviceArray(2 * n)
    x.fill_synth(synth.SEED)
    y = api.DeviceArray(2 * n)
    f = api.Fir(tr + 1j * ti, data_complex=True, algo=L.FIR_ALGO_FFT)
    f.process_stream(x, y, n)
    W, H = 1 << 13, 255
    blk = lambda t, v: orc.Blkconv(t, 4096).stream(np.ascontiguousarray(v))
    for s0 in (0, 3840 * 33333 - 4000, n - W):
        lo = max(0, s0 - H)
        seg = synth.synth_cf32(s0 + W - lo, first_sample=lo)
        xr, xi = seg[0::2], seg[1::2]
        ref_r = (blk(tr, xr) - blk(ti, xi))[s0 - lo:]
        ref_i = (blk(tr, xi) + blk(ti, xr))[s0 - lo:]
        got = _window(y, s0, W)
        assert synth.rel_rms(got[0::2], ref_r) <= TOL and synth.rel_rms(got[1::2], ref_i) <= TOL, s0
    x.free()
    y.free()


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


def test_fir_single_channel_beyond_4_gib(api, L, orc):
    """One channel of 2^29 + 12 345 cf32 samples: 4 GiB + of input and of output in ONE stream, so byte
    offsets inside a channel pass 2^32 (the 64-channel config only has channel BASES that far out).
    Windows at the start, either side of the 4 GiB byte boundary (sample 2^29), at a transform seam
    beyond it and at the ragged end."""
    n = (1 << 29) + 12345
    taps = synth.taps_cfg2()
    x = api.DeviceArray(2 * n)
    x.fill_synth(synth.SEED)
    y = api.DeviceArray(2 * n)
    f = api.Fir(taps, data_complex=True)          # the product's default kernel
    f.process_stream(x, y, n)
    W, H = 1 << 12, 255
    seam = 3840 * ((1 << 29) // 3840 + 2)
    for s0 in (0, (1 << 29) - W // 2, (1 << 29) + 5, seam - 100, n - W):
        lo = max(0, s0 - H)
        seg = synth.synth_cf32(s0 + W - lo, first_sample=lo)
        got = _window(y, s0, W)
        for part in (0, 1):
            ref = orc.Blkconv(taps, 4096).stream(np.ascontiguousarray(seg[part::2]))[s0 - lo:]
            assert synth.rel_rms(got[part::2], ref) <= TOL, (s0, part)
    x.free()
    y.free()


def test_resample_beyond_4_gib_input(api, L, orc, monkeypatch):
    """Resample 5/3 (381 taps, the default transform-domain kernel) over 2^29 + 2^22 + 777 cf32 samples:
    input byte offsets pass 2^32 with 32 MiB to go.  Windows of the output whose inputs sit either side of that boundary, at a
    pass seam beyond it and at the end, against the oracle."""
    n = (1 << 29) + (1 << 22) + 777
    taps = synth.taps_cfg3()
    x = api.DeviceArray(2 * n)
    x.fill_synth(synth.SEED)
    cap = n * 3 // 5 + 8
    y = api.DeviceArray(2 * cap)
    r = api.Rs(taps, 3, 4096, mode=L.RS_RESAMPLE, data_complex=True)
    k = r.process_stream(x, n, y, cap, 5.0 / 3.0)
    assert abs(k - (n * 3 + 4) // 5) <= 1
    W, P = 4200, 1386
    kb = ((1 << 29) * 3) // 5                  # the output whose input is sample 2^29 (byte offset 2^32)
    k0s = [0, kb - W // 2, kb + 7, P * (kb // P + 3) - 700, k - W - 2]
    k0s = [k0 - k0 % 3 for k0 in k0s]
    _rs_windows_vs_oracle(orc, "Resample", y, taps, 3, 5, k0s, W, n, exact=False, hist_samples=150)
    x.free()
    y.free()


def _seeded_starts(rng, total, W, fixed, count=32):
    """`count` window starts in [0, total - W]: the fixed ones (stream start, last window, seams) plus
    seeded random ones, sorted, each pulled down to a multiple of 6 (whole phase periods of the 5/3 law)."""
    starts = [min(max(0, int(s)), total - W) for s in fixed]
    while len(starts) < count:
        starts.append(int(rng.integers(0, total - W)))
    return sorted({s - s % 6 for s in starts})


def test_parity_at_scale_32_windows_of_2pow16(api, L, orc, monkeypatch):
    """SURVEY.md 8(d) "parity check at scale", as written: for the 2^28 / 2^30 configs compare 32 seeded
    windows of 2^16 outputs -- always the stream start, the last window and windows straddling the GPU's
    block seams -- against the CPU oracle fed the corresponding input span plus the filter's history.
    The kernels bench.py times (default mode): cfg2 FIR, cfg3 resample 5/3, cfg4 decimate by 8."""
    rng = np.random.default_rng(synth.SEED)
    W = 1 << 16
    # ---- cfg2: 256-tap FIR, 2^28 cf32 (transform seams every 3840 samples; runs of eight per work counter)
    n = 1 << 28
    taps = synth.taps_cfg2()
    x = api.DeviceArray(2 * n)
    x.fill_synth(synth.SEED)
    y = api.DeviceArray(2 * n)
    api.Fir(taps, data_complex=True).process_stream(x, y, n)
    worst, H = 0.0, 255
    for s0 in _seeded_starts(rng, n, W, [0, n - W, 3840 * 8 * 1000 - W // 2, 3840 * 34953 - 100, n // 2]):
        lo = max(0, s0 - H)
        seg = synth.synth_cf32(s0 + W - lo, first_sample=lo)
        got = _window(y, s0, W)
        for part in (0, 1):
            ref = orc.Blkconv(taps, 4096).stream(np.ascontiguousarray(seg[part::2]))[s0 - lo:]
            e = synth.rel_rms(got[part::2], ref)
            worst = max(worst, e)
            assert e <= TOL, ("fir", s0, part, e)
    print(f"cfg2: 32 x 2^16 windows, worst rel-RMS {worst:.2e}")
    y.free()
    # ---- cfg3: resample 5/3, 381 taps, the same 2^28 stream (a pass = 1386 outputs)
    taps3 = synth.taps_cfg3()
    cap = n * 3 // 5 + 8
    y3 = api.DeviceArray(2 * cap)
    k = api.Rs(taps3, 3, 4096, mode=L.RS_RESAMPLE, data_complex=True).process_stream(x, n, y3, cap, 5.0 / 3.0)
    k0s = _seeded_starts(rng, k - 2, W, [0, k - 2 - W, 1386 * 40000 - W // 2, 1386 * 8 * 5000 + 3])
    worst = _rs_windows_vs_oracle(orc, "Resample", y3, taps3, 3, 5, k0s, W, n, exact=False, hist_samples=150)
    print(f"cfg3: {len(k0s)} x 2^16 windows, worst rel-RMS {worst:.2e}")
    y3.free()
    x.free()
    # ---- cfg4: decimate by 8, 64 taps, 2^30 cf32 (workgroup tiles of 512 outputs)
    n4 = 1 << 30
    taps4 = synth.taps_cfg4()
    x4 = api.DeviceArray(2 * n4)
    x4.fill_synth(synth.SEED)
    cap4 = n4 // 8 + 8
    y4 = api.DeviceArray(2 * cap4)
    k4 = api.Rs(taps4, 1, 4096, mode=L.RS_DECIMATE, data_complex=True).process_stream(x4, n4, y4, cap4, 8.0)
    assert k4 == n4 // 8
    k0s = _seeded_starts(rng, k4, W, [0, k4 - W, 512 * 100000 - W // 2, (1 << 26) - W // 2])
    worst = _rs_windows_vs_oracle(orc, "Decimate", y4, taps4, 1, 8, k0s, W, n4, exact=False, hist_samples=80)
    print(f"cfg4: {len(k0s)} x 2^16 windows, worst rel-RMS {worst:.2e}")
    x4.free()
    y4.free()


def _general_rate_windows(api, L, orc, taps, U, rate, n, cuts, fmt, n_windows, seed, calls_per_window):
    """One stream of n samples through the DEFAULT dispatch at a general rate, in the process_stream calls `cuts` names
    (multiples of the 4096-sample reference call), then windows of whole reference calls against the oracle.  The float32 time
    recurrence has no closed form, so the oracle is walked to each window: the time law of the calls before it without their
    samples (orc.Resample.skip_calls -- pinned against the compiled reference in tests/test_oracle.py), ONE real call for the
    history and the pending-output value, then the window's calls for real.  fmt: "cf32" | "f32" | "u8" (complex bytes).
    Returns (outputs, worst rel-RMS).  The TOTAL number of outputs and every window's offset come from the oracle's own count."""
    B = 4096
    cplx = fmt != "f32"
    w = 2 if cplx else 1
    rate = float(np.float32(rate))
    assert n % B == 0 and all(c % B == 0 for c in cuts)
    n_calls = n // B
    if fmt == "u8":
        raw = np.random.default_rng(seed).integers(0, 256, size=2 * n, dtype=np.uint8)
        x = api.DeviceArray.from_bytes(raw)
        host = lambda a, m: orc.rx_u8_to_f32(raw[2 * a: 2 * (a + m)])
    else:
        x = api.DeviceArray(w * n)
        x.fill_synth(synth.SEED, channel=seed)
        host = (lambda a, m: synth.synth_cf32(m, ch=seed, first_sample=a)) if cplx else (lambda a, m: synth.synth_f32(m, synth.SEED, seed, first=a))
    cap = int(n / rate) + 4 * (n // B) + 4096          # (the library wants room for every reference call's worst case)
    y = api.DeviceArray(w * cap)
    r = api.Rs(taps, U, B, mode=L.RS_RESAMPLE, data_complex=cplx)
    if fmt == "u8":
        r.set_input_format(L.FMT_U8)
    bytes_per = {"cf32": 8, "f32": 4, "u8": 2}[fmt]
    k = 0
    for a0, a1 in zip([0] + list(cuts), list(cuts) + [n]):
        k += r.process_stream(x.ptr + bytes_per * a0, a1 - a0, y.ptr + 4 * w * k, cap - k, rate)
    total = orc.Resample(taps, U, B).skip_calls(n_calls, B, rate)
    assert k == total, (k, total)                     # the reference's own count over all the calls
    rng = np.random.default_rng(seed + 1)
    fixed = [0, n_calls - calls_per_window] + [c // B - calls_per_window // 2 for c in cuts] + [3968 * 7 // B, n_calls // 2]
    starts = [min(max(0, int(c)), n_calls - calls_per_window) for c in fixed]
    while len(starts) < n_windows:
        starts.append(int(rng.integers(0, n_calls - calls_per_window)))
    worst = 0.0
    walker, at, k_at = orc.Resample(taps, U, B), 0, 0      # the time law alone, walked once through the stream
    for c0 in sorted(set(starts))[:max(n_windows, len(fixed))]:
        pre = 1 if c0 > 0 else 0                          # one real call in front (history, m_last_remain)
        k_at += walker.skip_calls(c0 - pre - at, B, rate) if c0 - pre > at else 0
        at = max(at, c0 - pre)
        assert at == c0 - pre                             # (windows ascending; they may overlap, the walker never passes one)
        seg = host((c0 - pre) * B, (calls_per_window + pre) * B)
        ol = U * B + 8                                    # what one call can emit at most (step >= 1)
        refs, k0 = [], k_at
        for part in range(w):
            o = orc.Resample(taps, U, B)
            o.set_time(walker.get_time())
            xs = np.ascontiguousarray(seg[part::w])
            outs = [o.process(xs[i * B:(i + 1) * B], ol, rate) for i in range(calls_per_window + pre)]
            if pre:
                k0 = k_at + outs[0][0]
                outs = outs[1:]
            refs.append(np.concatenate([v[:m] for m, v in outs]))
        m = len(refs[0])
        assert m >= (1 << 16)
        got = y.to_numpy(w * m, offset=w * k0)
        for part in range(w):
            e = synth.rel_rms(got[part::w], refs[part])
            worst = max(worst, e)
            assert e <= TOL, (fmt, rate, c0, part, e)
    x.free()
    y.free()
    return k, worst


@pytest.mark.parametrize("rate,calls", [(1.77, 30), (0.77, 13)])
def test_general_rate_parity_at_scale_32_windows(api, L, orc, rate, calls):
    """VERDICT r4 missing 2: /root/reference/libdsp/resample.cxx:100-148 at a NON-integer step on 2^28 cf32 samples through the
    DEFAULT dispatch (poly_gen4096_kernel), BASELINE cfg3's 381 taps in 3 phases, rates 1.77 (the reference driver's,
    libdsp/test/test_decimate.py:24) and 0.77 (test_resample.py:24: more outputs than inputs): 32 seeded windows of >= 2^16
    outputs -- stream start, the last calls, the transform kernel's block seams (every 3968 samples: each window spans
    several), a process_stream call seam in the middle of a window -- against the oracle walked to each window; n_out equal to
    the reference's count over all 65 536 calls."""
    n = 1 << 28
    cuts = [(1 << 27) + 5 * 4096]
    k, worst = _general_rate_windows(api, L, orc, synth.taps_cfg3(), 3, rate, n, cuts, "cf32", 32, 0, calls)
    print(f"general rate {rate}: {k} outputs, 32 windows of {calls} calls, worst rel-RMS {worst:.2e}")


@pytest.mark.parametrize("fmt", ["f32", "u8"])
def test_general_rate_real_and_u8_streams_at_2pow26(api, L, orc, fmt):
    """The same law on a REAL float32 stream (libdsp's native type: two blocks per transform) and on the u8 receive wire
    format (converted on load), 2^26 samples, rate 1.77, default dispatch: 12 windows incl. start, end and the call seam."""
    n = 1 << 26
    k, worst = _general_rate_windows(api, L, orc, synth.taps_cfg3(), 3, 1.77, n, [(1 << 25) + 3 * 4096], fmt, 12, 5, 30)
    print(f"general rate 1.77, {fmt}: {k} outputs, worst rel-RMS {worst:.2e}")


def _windows_any_format(orc, y, taps, U, S, k0s, W, n_in, fmt, hist_samples):
    """_rs_windows_vs_oracle for the three input forms of ONE buffer of synthetic float32: "cf32" (complex samples), "f32" (real samples:
    float i of the stream) and "u8" (the stream's bytes read as (I, Q) byte pairs, converted by the oracle's own rx converter); the
    fused default kernels, so within TOL.  Returns the worst rel-RMS."""
    w = 1 if fmt == "f32" else 2
    g = int(np.gcd(S, U))
    per = S // g
    rate = float(np.float32(S) / np.float32(U))
    worst = 0.0
    for k0 in k0s:
        nin0 = (k0 * S) // U
        a0 = max(0, ((nin0 - hist_samples - per) // per) * per)
        if fmt == "u8":
            a0 -= a0 % (2 * per)                   # whole floats: two complex u8 samples each
        j0 = k0 - a0 * U // S
        n_span = min(((k0 + W) * S) // U + 2 - a0, n_in - a0)
        if fmt == "u8":
            n_span -= n_span % 2
            seg = orc.rx_u8_to_f32(synth.synth_f32(n_span // 2, first=a0 // 2).view(np.uint8))
        else:
            seg = synth.synth_f32(w * n_span, first=w * a0)
        got = y.to_numpy(w * W, offset=w * k0)
        for part in range(w):
            ref, _ = orc.Resample(taps, U, 4096).stream(np.ascontiguousarray(seg[part::w]), rate)
            m = min(W, len(ref) - j0)
            assert m > W // 2, (fmt, k0, m)
            e = synth.rel_rms(got[part::w][:m], ref[j0:j0 + m])
            worst = max(worst, e)
            assert e <= TOL, (fmt, U, S, k0, part, e)
    return worst


def test_round5_kernels_parity_at_scale(api, L, orc):
    """SURVEY.md 8(d) "parity check at scale" for the kernels of poly_rt_dma.hip, at the sizes bench.py times them: 2^31 bytes of synthetic
    stream read as 2^28 complex samples (decimate by 7, 7/4: the LDS-DMA kernel), as 2^29 real samples (decimate by 3, 2/3, interpolate x4:
    the register-window kernel) and as u8 (I, Q) byte pairs (decimate by 7 from the first 2^29 bytes: the raw tile by DMA, converted once) --
    16 windows of 2^15 outputs each: the stream's start and end, tile seams, seeded random places; the oracle fed the window's input span
    plus history (libdsp/decimate.cxx:132-140, libdsp/resample.cxx:100-114 at an integer step)."""
    rng = np.random.default_rng(synth.SEED + 5)
    W = 1 << 15
    nf = 1 << 29                                   # floats in the buffer
    x = api.DeviceArray(nf)
    x.fill_synth(synth.SEED)
    for fmt, U, S in (("cf32", 1, 7), ("cf32", 4, 7), ("f32", 1, 3), ("f32", 3, 2), ("f32", 4, 1), ("u8", 1, 7)):
        w = 1 if fmt == "f32" else 2
        n = nf if fmt == "f32" else (nf // 2 if fmt == "cf32" else nf // 2)      # samples: u8 reads the first 2^29 bytes = 2^28 complex samples
        if fmt == "u8":
            n = nf // 2
        taps = synth.lowpass_taps(32 * U - (1 if U > 1 else 0), 0.9 * min(1.0 / U, 1.0 / S), gain=float(U))
        rate = float(np.float32(S) / np.float32(U))
        cap = n * U // S + 64
        y = api.DeviceArray(w * cap)
        r = api.Rs(taps, U, 4096, mode=L.RS_RESAMPLE, data_complex=fmt != "f32")
        if fmt == "u8":
            r.set_input_format(L.FMT_U8)
        k = r.process_stream(x, n, y, cap, rate)
        assert abs(k - n * U // S) <= 1, (fmt, U, S, k)
        per_out = U // int(np.gcd(S, U))
        starts = [0, k - 2 - W, (k // 2), 4096 * per_out * 1000 - W // 2, 2048 * per_out * 33333 + 5]
        while len(starts) < 16:
            starts.append(int(rng.integers(0, k - 2 - W)))
        k0s = sorted({max(0, min(s0, k - 2 - W) - min(s0, k - 2 - W) % (12 * per_out)) for s0 in starts})       # inside the stream, on whole phase periods
        worst = _windows_any_format(orc, y, taps, U, S, k0s, W, n, fmt, hist_samples=40)
        print(f"{fmt} {S}/{U}: {len(k0s)} windows of 2^15 outputs at n = 2^{n.bit_length() - 1}, worst rel-RMS {worst:.2e}")
        r.close()
        y.free()
    x.free()
