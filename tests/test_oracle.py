"""The CPU oracle (oracle/sfe_oracle.c) against the golden fixtures and, when it has been
built, against the compiled reference itself (oracle/_ref).  No GPU."""
import numpy as np
import pytest

from simplefe_amd import synth

RATES = ("1p77", "5o3", "8", "2p5")


def _stream(cls, taps, U, B, x, rate, out_len):
    o = cls(taps, U, B)
    return o.stream(x, rate, chunk=B, out_len=out_len)


# ---------------------------------------------------------------- resample / decimate
@pytest.mark.parametrize("tag", RATES + ("0p77",))
def test_resample_restatement_bit_exact_vs_reference_vector(orc, g4, tag):
    """libdsp/test/test_resample.py:22-25 driver shape; outputs from the compiled reference."""
    y, ns = _stream(orc.Resample, g4["taps"], int(g4["U"]), int(g4["B"]), g4["x"],
                    float(g4[f"rate_{tag}"]), 4 * int(g4["B"]))
    assert ns == g4[f"n_{tag}"].tolist()
    assert np.array_equal(y, g4[f"y_{tag}"])          # bit-exact


@pytest.mark.parametrize("tag", RATES)
def test_decimate_restatement_bit_exact_vs_reference_vector(orc, g4, tag):
    """libdsp/test/test_decimate.py:22-25; decimate == resample exactly (test_decimate.py:36)."""
    y, ns = _stream(orc.Decimate, g4["taps"], int(g4["U"]), int(g4["B"]), g4["x"],
                    float(g4[f"rate_{tag}"]), 4 * int(g4["B"]))
    assert ns == g4[f"n_{tag}"].tolist()
    assert np.array_equal(y, g4[f"y_{tag}"])


@pytest.mark.parametrize("name", ["cfg3", "cfg4", "gen", "gen2"])
@pytest.mark.parametrize("B", [4096, 1000, 1001])
@pytest.mark.parametrize("which", ["Resample", "Decimate"])
def test_baseline_shapes_bit_exact(orc, g5, name, B, which):
    """BASELINE cfg3/cfg4 shapes, chunked at 4096/1000/1001 (1001 exercises the leftover
    branch resample.cxx:141-145,119-123)."""
    x = synth.synth_f32(int(g5["n"]), seed=int(g5["seed"]))
    rate = float(g5[f"{name}_rate"])
    ol = int(np.ceil(B / rate)) + 2
    y, ns = _stream(getattr(orc, which), g5[f"{name}_taps"], int(g5[f"{name}_U"]), B, x, rate, ol)
    key = f"{name}_y" if f"{name}_y" in g5.files else f"{name}_y_B{B}"
    assert ns == g5[f"{name}_n_B{B}"].tolist()
    assert np.array_equal(y, g5[key])


def test_parameter_errors_return_zero(orc, g4, capfd):
    """resample.cxx:91-98, decimate.cxx:75-87: message on stdout and return 0."""
    taps = g4["taps"]
    r = orc.Resample(taps, 4, 128)
    n, _ = r.process(np.zeros(129, np.float32), 512, 1.5)     # n_in > blksize
    assert n == 0
    n, _ = r.process(np.zeros(128, np.float32), 512, 0.2)     # rate < 1/U
    assert n == 0
    n, _ = r.process(np.zeros(128, np.float32), 10, 1.5)      # out_len too small
    assert n == 0
    d = orc.Decimate(taps, 4, 128)
    n, _ = d.process(np.zeros(128, np.float32), 512, 0.9)     # rate < 1
    assert n == 0
    n, _ = d.process(np.zeros(129, np.float32), 512, 2.0)
    assert n == 0


def test_against_live_reference_random(orc):
    """When oracle/_ref is built: random taps/rates/chunks, restatement == reference."""
    if orc.ref_lib() is None:
        pytest.skip("oracle/_ref/libsferef.so not present")
    rng = np.random.default_rng(3)
    for trial in range(12):
        U = int(rng.integers(1, 6))
        nt = int(rng.integers(U, 60))
        B = int(rng.integers(max(nt // U + 2, 8), 300))
        taps = rng.standard_normal(nt).astype(np.float32)
        x = rng.standard_normal(5 * B + 17).astype(np.float32)
        rate = float(np.float32(rng.uniform(1.0, 6.0)))
        ol = int(np.ceil(B / rate)) + 2
        a, na = _stream(orc.RefResample, taps, U, B, x, rate, ol)
        b, nb = _stream(orc.Resample, taps, U, B, x, rate, ol)
        c, nc = _stream(orc.RefDecimate, taps, U, B, x, rate, ol)
        d, nd = _stream(orc.Decimate, taps, U, B, x, rate, ol)
        assert na == nb and nc == nd
        assert np.array_equal(a, b) and np.array_equal(c, d)
        assert np.array_equal(a, c)
    # rate < 1 (resample only, >= 1/U)
    taps = rng.standard_normal(31).astype(np.float32)
    x = rng.standard_normal(1024).astype(np.float32)
    a, na = _stream(orc.RefResample, taps, 4, 128, x, 0.41, 4 * 128)
    b, nb = _stream(orc.Resample, taps, 4, 128, x, 0.41, 4 * 128)
    assert na == nb and np.array_equal(a, b)


@pytest.mark.parametrize("rate", [1.77, 0.77, 2.5, 1.0009])
def test_time_law_skip_lands_where_the_reference_is(orc, g4, rate):
    """orc_resample_skip_calls (the helper the full-size general-rate windows restart the oracle with, tests/test_gpu_fullsize.py):
    after the time law of c calls WITHOUT samples and ONE real call, every further call's outputs -- values and count -- are
    bit for bit those of the reference's own class run over the whole stream (oracle/_ref when present, else the restatement
    run in full); and the outputs skipped are as many as the reference emitted.  Several cuts, the reference driver's own
    shape (libdsp/test/test_decimate.py:10-17: 31 taps, U = 4, blksize 128) and BASELINE cfg3's."""
    rate = float(np.float32(rate))
    full_cls = orc.RefResample if orc.ref_lib() is not None else orc.Resample
    for taps, U, B, calls in ((g4["taps"], int(g4["U"]), 128, 60), (synth.taps_cfg3(), 3, 512, 24)):
        x = synth.synth_f32(calls * B, ch=31)
        ol = int(B / rate) + 8
        full = full_cls(taps, U, B)
        per_call = [full.process(x[c * B:(c + 1) * B], ol, rate) for c in range(calls)]
        plen = (len(taps) + U - 1) // U
        for cut in (1, 7, calls // 2, calls - 3):
            if (cut + 1) * B < plen + 2:
                continue                                  # the one real call must refill the whole history
            o = orc.Resample(taps, U, B)
            skipped = o.skip_calls(cut, B, rate)
            assert skipped == sum(n for n, _ in per_call[:cut]), (rate, cut)
            n, _ = o.process(x[cut * B:(cut + 1) * B], ol, rate)             # history and m_last_remain: values not compared
            assert n == per_call[cut][0]
            for c in range(cut + 1, calls):
                n, y = o.process(x[c * B:(c + 1) * B], ol, rate)
                assert n == per_call[c][0] and np.array_equal(y[:n], per_call[c][1][:n]), (rate, cut, c)


# ------------------------------------------------------------------------- blkconv
def test_blkconv_known_answer(orc, g1):
    """libdsp/test/test_blkconv.cxx:5-33: boxcar(5), fft 32 -> blksize 28; ones then zeros."""
    c = orc.Blkconv(g1["taps"], int(g1["fft_len"]))
    assert c.blk == int(g1["blksize"])
    c.buf[: c.blk] = g1["in1"]
    c.process()
    assert np.allclose(c.buf[: c.blk], g1["out1"], atol=float(g1["print_tol"]))
    c.buf[: c.blk] = g1["in2"]
    c.process()
    assert np.allclose(c.buf[: c.blk], g1["out2"], atol=float(g1["print_tol"]))


@pytest.mark.parametrize("name", ["kat", "bpsk", "cfg1", "cfg2"])
def test_blkconv_restatement_vs_reference_class(orc, g6, name):
    """The restatement against outputs of the reference's own blkconv class (blkconv.cxx compiled
    unmodified on ROCm's libhipfftw, run on a GPU box: tests/golden/make_golden_blkconv.py), fed
    block by block as its callers do.  The two use different float32 FFTs: equal to rounding."""
    taps, fft_len, x, want = g6[f"{name}_taps"], int(g6[f"{name}_fft_len"]), g6[f"{name}_x"], g6[f"{name}_y"]
    c = orc.Blkconv(taps, fft_len)
    assert c.blk == fft_len + 1 - len(taps) and len(x) % c.blk == 0
    got = np.empty_like(x)
    for off in range(0, len(x), c.blk):
        c.buf[: c.blk] = x[off: off + c.blk]
        c.process()
        got[off: off + c.blk] = c.buf[: c.blk]
    assert synth.rel_rms(got, want) < 1e-6
    assert np.abs(got - want).max() < 2e-6 * max(1.0, float(np.abs(want).max()))


@pytest.mark.parametrize("name", ["kat", "bpsk", "cfg1", "cfg2", "rrc551"])
def test_blkconv_restatement_vs_reference_on_its_own_fftw(orc, g7, name):
    """THE PIN AT THE FFTW BOUNDARY.  g7 = outputs of the reference itself: its blkconv.cxx calling
    its own vendored FFTW 3.3.5 (libfftw3f-3.dll mapped in process, oracle/pe/), block by block.
    The restatement carries an own float32 FFT, so equality is to float32 transform rounding:
    observed 3.0e-8 (kat) and 2.6-3.0e-7 (the streams) against the 1e-5 bar."""
    taps, fft_len, x, want = g7[f"{name}_taps"], int(g7[f"{name}_fft_len"]), g7[f"{name}_x"], g7[f"{name}_y"]
    c = orc.Blkconv(taps, fft_len)
    assert c.blk == fft_len + 1 - len(taps) and len(x) % c.blk == 0
    got = np.empty_like(x)
    for off in range(0, len(x), c.blk):
        c.buf[: c.blk] = x[off: off + c.blk]
        c.process()
        got[off: off + c.blk] = c.buf[: c.blk]
    assert synth.rel_rms(got, want) <= 1e-5           # north_star's bar
    assert synth.rel_rms(got, want) < 5e-7            # what is observed, with margin
    assert np.abs(got - want).max() < 2e-6 * max(1.0, float(np.abs(want).max()))


def test_blkconv_pulse_shaping_vs_float64(orc, g1):
    """bpsk.cxx:122-164 shape (111 taps, fft 2048) against float64 direct convolution."""
    c = orc.Blkconv(g1["g2_taps"], int(g1["g2_fft_len"]))
    y = c.stream(g1["g2_x"])
    assert synth.rel_rms(y, g1["g2_y64"]) < 1e-6


def test_blkconv_cfg1_vs_float64(orc):
    """BASELINE cfg1: 63-tap real FIR, 2^20 float32 samples, fft 1024 (blk 962)."""
    from scipy.signal import fftconvolve
    taps = synth.taps_cfg1()
    x = synth.synth_f32(1 << 20)
    c = orc.Blkconv(taps, 1024)
    assert c.blk == 962
    y = c.stream(x)
    ref = fftconvolve(x.astype(np.float64), taps.astype(np.float64))[: len(x)]
    assert synth.rel_rms(y, ref) < 1e-6


def test_blkconv_non_pow2_and_block_independence(orc):
    """The block law must not show in the output: fft 96 (slow path) == fft 128 == direct."""
    rng = np.random.default_rng(5)
    taps = rng.standard_normal(17).astype(np.float32)
    x = rng.standard_normal(1000).astype(np.float32)
    ref = np.convolve(x.astype(np.float64), taps.astype(np.float64))[: len(x)]
    for fl in (96, 128, 64):
        y = orc.Blkconv(taps, fl).stream(x)
        assert synth.rel_rms(y, ref) < 2e-6, fl


# ---------------------------------------------------------------------- converters
def test_rx_converters(orc):
    """source_c_impl.cc:121-132 / source_f_impl.cc:120-129: (b-128)/127."""
    b = np.arange(256, dtype=np.uint8)
    f = orc.rx_u8_to_f32(b)
    assert np.array_equal(f, ((b.astype(np.int32) - 128).astype(np.float32) * np.float32(1.0 / 127.0)))
    c = orc.rx_u8_to_cf32(b)
    assert np.array_equal(c, f)            # same arithmetic, interleaved


def test_tx_converter_packing(orc):
    """sink_f_impl.cc:117-143: ((short)(x*511)+512)&0x3FF, 4 samples -> 5 bytes."""
    x = np.array([0.0, 1.0, -1.0, 0.5, -0.25, 0.999, -0.999, 0.001], dtype=np.float32)
    out = orc.tx_f32_to_10bit(x)
    assert len(out) == 10
    u = ((x * np.float32(511)).astype(np.int16).astype(np.int32) + 512) & 0x3FF
    for g in range(2):
        v = u[4 * g: 4 * g + 4]
        assert out[5 * g] == (v[0] >> 8) | ((v[1] >> 8) << 2) | ((v[2] >> 8) << 4) | ((v[3] >> 8) << 6)
        assert out[5 * g + 1: 5 * g + 5].tolist() == [int(t & 0xFF) for t in v]


def test_synth_is_exact_and_in_range():
    x = synth.synth_f32(1 << 12)
    assert x.dtype == np.float32 and x.min() >= -1.0 and x.max() < 1.0
    assert np.array_equal(x * np.float32(2 ** 23), np.round(x * np.float32(2 ** 23)))
    assert np.array_equal(synth.synth_f32(100, first=50), synth.synth_f32(150)[50:])
    assert abs(float(x.mean())) < 0.05


def test_all_cores_helpers_agree_with_the_single_object(orc):
    """bench.py's all-host-cores baseline cuts one stream into spans, one object per span with
    n_taps-1 samples of lead-in: the stitched blkconv output equals the single
    object's to float32 FFT rounding, and the resamplers produce the same output count."""
    from simplefe_amd import synth
    x = synth.synth_f32(200000)
    t = synth.taps_cfg2()
    y1 = orc.Blkconv(t, 4096).stream(x)
    for C in (1, 3, 8):
        assert synth.rel_rms(orc.blkconv_stream_mt(t, 4096, x, C), y1) <= 1e-6
    t4 = synth.taps_cfg4()
    ref, _ = orc.Decimate(t4, 1, 4096).stream(x, 8.0)
    assert orc.rs_stream_mt("decimate", t4, 1, 4096, 8.0, x, 8, 1) == len(ref)
    k4 = orc.rs_stream_mt("decimate", t4, 1, 4096, 8.0, x, 8, 4)
    assert len(ref) <= k4 <= len(ref) + 4 * 12           # + the lead-ins' outputs
    if orc.ref_lib() is not None and hasattr(orc.ref_lib(), "ref_rs_stream_mt"):
        assert orc.rs_stream_mt("decimate", t4, 1, 4096, 8.0, x, 8, 4, reference=True) == k4
