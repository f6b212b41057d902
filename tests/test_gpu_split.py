"""One stream cut into spans, on one GPU with G handles -- the
per-GPU objects of a G-GPU job -- each fed its span plus the halo: `-m gpu`.
Reference state being cut: libdsp/blkconv.cxx:105-109 (m_overlap), resample.cxx:119-150 (time law)."""
import numpy as np
import pytest

from simplefe_amd import shard, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from simplefe_amd import api as a
    return a


@pytest.fixture(scope="module")
def L():
    from simplefe_amd import lib
    return lib


def _fir_spans(api, L, taps, x, spans, cplx, halo_len):
    w = 2 if cplx else 1
    outs = []
    for first, count in spans:
        if count == 0:
            continue
        f = api.Fir(taps, data_complex=cplx, algo=L.FIR_ALGO_FFT)       # one handle per span == per GPU
        lo = max(0, first - halo_len)
        if first > lo:
            d_h = api.DeviceArray.from_numpy(x[lo * w: first * w])
            f.load_history(d_h, first - lo)
        d_in = api.DeviceArray.from_numpy(x[first * w:(first + count) * w])
        d_out = api.DeviceArray(count * w)
        f.process_stream(d_in, d_out, count)
        outs.append(d_out.to_numpy())
    return np.concatenate(outs)


@pytest.mark.parametrize("G", [2, 3, 8])
def test_fir_stream_split_on_transform_boundaries_is_bit_identical(api, L, G):
    """Cuts on multiples of the transform advance (3840) and a halo as long as the transform overlap
    (n_taps-1 rounded up to 256 samples): every span's transforms then see exactly the uncut
    stream's inputs, so the concatenated output is the one-handle output bit for bit.  With the
    minimal halo of n_taps-1 = 255 samples the 256th overlap sample is a zero instead of a sample
    the filter never reaches: same result to rounding (checked too)."""
    taps = synth.taps_cfg2()
    n = 3840 * 37 + 1234
    x = synth.synth_cf32(n)
    whole = api.Fir(taps, data_complex=True, algo=L.FIR_ALGO_FFT).filter(x)[0]
    spans = [shard.span_block(n, G, r, quantum=3840) for r in range(G)]
    assert np.array_equal(_fir_spans(api, L, taps, x, spans, True, 256), whole)
    assert synth.rel_rms(_fir_spans(api, L, taps, x, spans, True, 255), whole) <= 1e-6


def test_fir_stream_split_anywhere_short_halo_and_real_data(api, L, orc):
    """Arbitrary cut points move the transform grid: equal to float32 rounding (<= 1e-6), for cf32
    and real streams; a halo longer than n_taps-1 changes nothing; against the oracle too."""
    taps = synth.lowpass_taps(111, 0.2)
    n = 100003
    for cplx in (True, False):
        x = synth.synth_cf32(n) if cplx else synth.synth_f32(n)
        whole = api.Fir(taps, data_complex=cplx, algo=L.FIR_ALGO_FFT).filter(x)[0]
        spans = [(0, 33333), (33333, 1), (33334, 40001), (73335, n - 73335)]
        for halo in (110, 5000):
            got = _fir_spans(api, L, taps, x, spans, cplx, halo)
            assert synth.rel_rms(got, whole) <= 1e-6
        if not cplx:
            assert synth.rel_rms(got, orc.Blkconv(taps, 4096).stream(x)) <= 1e-5


@pytest.mark.parametrize("which,U,S,n_taps", [("resample", 3, 5, 381), ("decimate", 1, 8, 64), ("resample", 2, 3, 100)])
def test_resampler_stream_split_with_seek(api, L, orc, which, U, S, n_taps):
    """Integer-valued steps: the time state at a cut is closed-form (sfe_dsp_rs_seek), the history is
    the phase_len samples before it.  Exact mode: spans concatenate to the uncut result -- and to the
    oracle's -- bit for bit, whatever the cut points (a cut that leaves a pending 'leftover' output,
    resample.cxx:141-145, included); default mode to rounding."""
    taps = synth.taps_cfg3() if n_taps == 381 else (synth.taps_cfg4() if n_taps == 64 else synth.lowpass_taps(n_taps, 0.3, gain=U))
    mode = L.RS_RESAMPLE if which == "resample" else L.RS_DECIMATE
    rate = float(np.float32(S) / np.float32(U))
    n = 90000
    x = synth.synth_cf32(n, ch=5)
    # cuts chosen so that some satisfy (cut*U - 1) % S == 0: a pending leftover at the cut
    cuts = [0, 20001, 20002, 20003, 20004, 20005, 55557, n]
    plen = (n_taps + U - 1) // U + 1
    for exact in (True, False):
        r0 = api.Rs(taps, U, 4096, mode=mode, data_complex=True)
        r0.set_exact(exact)
        whole = r0.resample_array(x[None, :], rate)[0]
        outs, pend = [], 0
        for a, b in zip(cuts[:-1], cuts[1:]):
            r = api.Rs(taps, U, 4096, mode=mode, data_complex=True)      # one handle per span == per GPU
            r.set_exact(exact)
            r.seek(a, rate)
            pend += r.get_state().leftover
            lo = max(0, a - plen)
            if a > lo:
                r.load_history(api.DeviceArray.from_numpy(x[2 * lo: 2 * a]), a - lo)
            outs.append(r.resample_array(x[None, 2 * a: 2 * b], rate)[0])
        got = np.concatenate(outs)
        assert got.shape == whole.shape
        if exact:
            assert np.array_equal(got, whole)
            for part in (0, 1):
                ref, _ = getattr(orc, "Resample" if which == "resample" else "Decimate")(taps, U, 4096).stream(
                    np.ascontiguousarray(x[part::2]), rate)
                assert np.array_equal(got[part::2], ref[: len(got) // 2])
        else:
            assert synth.rel_rms(got, whole) <= 1e-6
    assert pend >= 1 or S == 8        # at least one cut exercised the pending-leftover branch


def test_general_rate_split_carries_the_state(api, L, orc):
    """Non-integer step: no closed form (seek refuses); the state read from the span before is what
    the next span starts from -- bit-exact with the uncut stream."""
    taps = synth.lowpass_taps(31, 0.18, gain=4.0)
    U, rate, n = 4, 1.77, 40000
    x = synth.synth_f32(n, ch=9)
    r0 = api.Rs(taps, U, 128, mode=L.RS_RESAMPLE)
    r0.set_exact(True)
    whole = r0.resample_array(x[None, :], rate)[0]
    with pytest.raises(api.SfeError):
        api.Rs(taps, U, 128, mode=L.RS_RESAMPLE).seek(1000, rate)
    cut = 128 * 100                        # the reference's chunking (blksize) must line up for the replay
    ra = api.Rs(taps, U, 128, mode=L.RS_RESAMPLE)
    ra.set_exact(True)
    ya = ra.resample_array(x[None, :cut], rate)[0]
    rb = api.Rs(taps, U, 128, mode=L.RS_RESAMPLE)
    rb.set_exact(True)
    rb.set_state(ra.get_state())
    rb.load_history(api.DeviceArray.from_numpy(x[cut - 64: cut]), 64)
    yb = rb.resample_array(x[None, cut:], rate)[0]
    assert np.array_equal(np.concatenate([ya, yb]), whole)
