"""The general (non-integer-step) rate in the transform domain (poly_gen.hip; VERDICT r3 missing 3): the
reference's law -- all U phases of every input, two of them blended per output instant of the float32
time recurrence, libdsp/resample.cxx:100-148 -- by one forward and U inverse 4096-point transforms per
block.  Checked against the exact-mode direct kernel (poly_seg_kernel, itself bit-identical to the
compiled reference, tests/test_gpu_fuzz.py): the SAME number of outputs per call -- the (pos, mu)
sequence is the reference's own -- and values within 1e-5 rel-RMS (observed ~3e-7); and against the
oracle on the shapes the reference's own driver uses."""
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


def _pair(api, L, taps, U, B, cplx=True, nch=1):
    exact = api.Rs(taps, U, B, mode=L.RS_RESAMPLE, data_complex=cplx, n_channels=nch)
    exact.set_exact(True)
    fast = api.Rs(taps, U, B, mode=L.RS_RESAMPLE, data_complex=cplx, n_channels=nch)
    fast.set_algo(L.RS_ALGO_FFT)                  # the transform-domain kernel whatever the call's size
    return exact, fast


@pytest.mark.parametrize("seed", list(range(24)) + list(range(100, 116)) + list(range(200, 224)))
def test_transform_domain_general_rate_random_shapes(api, L, seed):
    """seeds 100-115: rates BELOW 1 -- more outputs than inputs, down to the reference's own limit 1 / U
    (libdsp/resample.cxx:91): a block then owns fewer input samples than its transform yields, so that its outputs fit.
    seeds >= 200: REAL float32 streams (libdsp's native type): two consecutive blocks per transform; every third of them
    at a rate below 1."""
    rng = np.random.default_rng(7000 + seed)
    cplx = seed < 200
    below = 100 <= seed < 200 or (seed >= 200 and seed % 3 == 0)
    U = int(rng.choice([1, 2, 3, 4, 5, 8]) if not below else rng.choice([2, 3, 4, 5, 8]))
    plen = int(rng.choice([1, 8, 43, 127, 128, 255, 256, 257, 300]))
    n_taps = U * plen - int(rng.integers(0, U))
    B = int(rng.choice([4096, 4096, 5000, 16384, 3840]))          # calls at least one block advance long (shorter: the direct kernel)
    taps = (rng.standard_normal(n_taps) / np.sqrt(plen)).astype(np.float32)
    rate = float(np.float32(rng.uniform(1.0, 6.0)))
    if seed % 6 == 0:
        rate = float(np.float32(1.0 + rng.uniform(0.0, 0.01)))        # just above 1: close to one output per input sample
    if below:
        rate = float(np.float32(rng.uniform(1.0 / U, 1.0)))
        if seed % 4 == 0:
            rate = float(np.float32(1.0 / U) * np.float32(1.0 + rng.uniform(0.0, 0.003)))       # next to the limit: a step of ~1
        while float(np.float32(rate) * np.float32(U)) < 1.0 or rate < 1.0 / U:                  # the reference refuses rate < 1 / U
            rate = float(np.nextafter(np.float32(rate), np.float32(2.0)))
    nch = int(rng.choice([1, 1, 3]))
    n = int(rng.choice([700, 4096, 12345, 40000, 150001]))
    w = 2 if cplx else 1                                                            # floats per sample
    x = np.stack([(synth.synth_cf32 if cplx else synth.synth_f32)(n, ch=300 + seed * 4 + c) for c in range(nch)])
    exact, fast = _pair(api, L, taps, U, B, cplx=cplx, nch=nch)
    cuts = sorted(set([0, n] + [int(v) // B * B for v in rng.integers(1, n, size=2)]))     # whole reference calls per piece
    for a0, a1 in zip(cuts[:-1], cuts[1:]):
        if a1 == a0:
            continue
        m = a1 - a0
        seg = np.ascontiguousarray(x[:, w * a0: w * a1])
        d_in = api.DeviceArray.from_numpy(seg)
        cap = int(m / rate) + 16 + 2 * (m // B + 1)
        de, df = api.DeviceArray(w * cap * nch), api.DeviceArray(w * cap * nch)
        ke = exact.process_stream(d_in, m, de, cap, rate)
        kf = fast.process_stream(d_in, m, df, cap, rate)
        assert ke == kf, (seed, U, plen, rate, n, a0, ke, kf)
        if ke == 0:
            continue
        ye = de.to_numpy().reshape(nch, w * cap)[:, : w * ke]
        yf = df.to_numpy().reshape(nch, w * cap)[:, : w * kf]
        for c in range(nch):
            assert synth.rel_rms(yf[c], ye[c]) <= TOL, (seed, U, plen, rate, n, a0, c, synth.rel_rms(yf[c], ye[c]))
            assert np.abs(yf[c] - ye[c]).max() <= 2e-5 * max(1.0, float(np.abs(ye[c]).max())), (seed, U, plen, rate)


@pytest.mark.parametrize("rate", [1.77, 2.5])
def test_transform_domain_general_rate_against_the_oracle(api, L, orc, g4, rate):
    """The reference driver's own shape (libdsp/test/test_decimate.py:13-25: 31 taps, U = 4) on a longer stream,
    I and Q as two real passes of the oracle; per-part rel-RMS and identical length."""
    taps, U = g4["taps"], int(g4["U"])
    n = 50000
    x = synth.synth_cf32(n, ch=77)
    r = api.Rs(taps, U, 4096, mode=L.RS_RESAMPLE, data_complex=True)
    r.set_algo(L.RS_ALGO_FFT)
    y = r.resample_array(x, float(np.float32(rate)))[0]
    for part in (0, 1):
        ref, _ = orc.Resample(taps, U, 4096).stream(np.ascontiguousarray(x[part::2]), float(np.float32(rate)))
        got = y[part::2]
        assert len(ref) - len(got) in (0, 1)
        assert synth.rel_rms(got, ref[: len(got)]) <= TOL


def test_default_dispatch_takes_the_transform_kernel_for_bulk_calls_only(api, L):
    """AUTO: bulk complex calls in fused arithmetic take the transform-domain kernel (round 4's second pass: at any rate the
    reference takes, below 1 too), and so do bulk REAL streams (two blocks per transform; include/sfe_dsp.h, api_rs.hip); small
    calls, the exact mode and SFE_RS_ALGO_DIRECT keep the direct kernel -- all of them the same law."""
    taps, U, rate = synth.taps_cfg3(), 3, float(np.float32(1.77))
    n = 1 << 18
    x = synth.synth_cf32(n, ch=5)
    ref = api.Rs(taps, U, 4096, mode=L.RS_RESAMPLE, data_complex=True)
    ref.set_exact(True)
    want = ref.resample_array(x, rate)[0]
    for algo in (L.RS_ALGO_AUTO, L.RS_ALGO_DIRECT, L.RS_ALGO_FFT):
        r = api.Rs(taps, U, 4096, mode=L.RS_RESAMPLE, data_complex=True)
        r.set_algo(algo)
        got = r.resample_array(x, rate)[0]
        assert len(got) == len(want) and synth.rel_rms(got, want) <= TOL, algo
    below = float(np.float32(0.77))
    r = api.Rs(taps, U, 4096, mode=L.RS_RESAMPLE, data_complex=True)
    e = api.Rs(taps, U, 4096, mode=L.RS_RESAMPLE, data_complex=True)
    e.set_exact(True)
    a, b = r.resample_array(x[: 2 << 16], below)[0], e.resample_array(x[: 2 << 16], below)[0]
    assert len(a) == len(b) and synth.rel_rms(a, b) <= TOL


@pytest.mark.parametrize("cplx", [True, False])
@pytest.mark.parametrize("rate", [1.77, 0.77])
def test_transform_domain_general_rate_u8_input(api, L, orc, cplx, rate):
    """The transform kernel reading the receive wire format (u8 offset binary) converts on load: the same bits as the
    same kernel fed the converted float32 samples in the same calls, and within tolerance of converter -> oracle."""
    taps, U = synth.taps_cfg3(), 3
    rate = float(np.float32(rate))
    n, B, w = 70000, 4096, 2 if cplx else 1
    b = np.random.default_rng(31).integers(0, 256, size=w * n, dtype=np.uint8)
    xf = orc.rx_u8_to_f32(b)
    outs = {}
    for fmt in ("f32", "u8"):
        r = api.Rs(taps, U, B, mode=L.RS_RESAMPLE, data_complex=cplx)
        r.set_algo(L.RS_ALGO_FFT)
        if fmt == "u8":
            r.set_input_format(L.FMT_U8)
        got = []
        for a0, a1 in ((0, 8 * B), (8 * B, 13 * B), (13 * B, n)):
            m = a1 - a0
            d_in = api.DeviceArray.from_bytes(b[w * a0:w * a1]) if fmt == "u8" else api.DeviceArray.from_numpy(xf[w * a0:w * a1])
            cap = int(m / rate) + 64
            d_out = api.DeviceArray(w * cap)
            k = r.process_stream(d_in, m, d_out, cap, rate)
            got.append(d_out.to_numpy(w * k))
        outs[fmt] = np.concatenate(got)
    assert np.array_equal(outs["u8"], outs["f32"])
    for part in range(w):
        ref, _ = orc.Resample(taps, U, B).stream(np.ascontiguousarray(xf[part::w]), rate)
        got = outs["u8"][part::w]
        assert len(ref) - len(got) in (0, 1) and synth.rel_rms(got, ref[: len(got)]) <= TOL


@pytest.mark.parametrize("cplx", [True, False])
@pytest.mark.parametrize("k", [1, 2, 3, 8, 31, 32, 33, 64])
def test_stream_that_ends_on_a_block_seam(api, L, k, cplx):
    """input sample n_in - 1 belongs to block floor(n_in / adv) -- one more block than ceil(n_in / adv) when the
    call's length is a multiple of the block advance (a real stream: an even multiple, two blocks per transform).  The
    reference still emits an output whose first sample is the call's last one while its phase leaves a successor
    (libdsp/resample.cxx:137-146); the launch was one workgroup short and that output was never stored.  Shape: BASELINE cfg3's
    381 taps in 3 phases (overlap 128, advance 3968) at rate 1.77; k = 32 is the advisor's own hit (last output at sample
    4095, phase 1 of the 31st call).  Count AND values against the exact kernel, the output buffer poisoned first."""
    taps, U, B = synth.taps_cfg3(), 3, 4096
    rate = float(np.float32(1.77))
    adv = 4096 - 128
    n = k * adv * (1 if cplx else 2)
    w = 2 if cplx else 1
    x = (synth.synth_cf32 if cplx else synth.synth_f32)(n, ch=900 + k)
    exact, fast = _pair(api, L, taps, U, B, cplx=cplx)
    d_in = api.DeviceArray.from_numpy(x)
    cap = int(n / rate) + 16 + 2 * (n // B + 1)
    poison = np.full(w * cap, 7.0e7, dtype=np.float32)
    de, df = api.DeviceArray.from_numpy(poison), api.DeviceArray.from_numpy(poison)
    ke = exact.process_stream(d_in, n, de, cap, rate)
    kf = fast.process_stream(d_in, n, df, cap, rate)
    assert ke == kf and ke > 0
    ye, yf = de.to_numpy()[: w * ke], df.to_numpy()[: w * kf]
    assert np.abs(yf).max() < 1.0e6, "an output of the call was never stored"
    assert synth.rel_rms(yf, ye) <= TOL
    assert np.abs(yf[-w:] - ye[-w:]).max() <= 2e-5 * max(1.0, float(np.abs(ye).max()))
