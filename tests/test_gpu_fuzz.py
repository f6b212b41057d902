"""Randomised parity sweep on the GPU (fixed seeds): many small random shapes through the C ABI
against the oracle -- ragged lengths, odd chunkings, channel strides, tap counts around the
kernel's internal boundaries (256/257 taps: one vs two overlap rows; 3840-sample transform
advance; 512-m polyphase tiles).  `-m gpu`."""
import numpy as np
import pytest

from simplefe_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from simplefe_amd import api as a
    return a


@pytest.fixture(scope="module")
def L():
    from simplefe_amd import lib
    return lib


def _cuts(rng, n, k):
    c = sorted(set([0, n] + [int(v) for v in rng.integers(1, max(n, 2), size=k)]))
    return [(a, b) for a, b in zip(c[:-1], c[1:]) if b > a]


@pytest.mark.parametrize("seed", range(12))
def test_fir_random_shapes(api, L, seed):
    rng = np.random.default_rng(1000 + seed)
    n_taps = int(rng.choice([1, 2, 31, 63, 111, 255, 256, 257, 258, 511, 513, 1000, 2049, 3841]))
    cplx = bool(rng.integers(0, 2))
    ctaps = cplx and bool(rng.integers(0, 2))
    nch = int(rng.choice([1, 1, 2, 3]))
    n = int(rng.choice([1, 7, 255, 3839, 3840, 3841, 7680, 7681, 12001, 20000, 50001]))
    stride = n + int(rng.integers(0, 50))
    taps = (rng.standard_normal(n_taps) / np.sqrt(n_taps)).astype(np.float32)
    if ctaps:
        taps = (taps + 1j * (rng.standard_normal(n_taps) / np.sqrt(n_taps))).astype(np.complex64)
    w = 2 if cplx else 1
    x = np.zeros((nch, w * stride), np.float32)
    for c in range(nch):
        x[c, : w * n] = synth.synth_f32(w * n, ch=seed * 8 + c)
    f = api.Fir(taps, data_complex=cplx, n_channels=nch)
    wo = 2 if (cplx or ctaps) else 1
    y = np.zeros((nch, wo * n), np.float32)
    for a, b in _cuts(rng, n, int(rng.integers(0, 4))):
        m = b - a
        d_in = api.DeviceArray.from_numpy(np.ascontiguousarray(x[:, w * a: w * a + w * stride - w * a][:, : w * (stride - a)]))
        d_out = api.DeviceArray(nch * wo * (m + 3))
        f.process_stream(d_in, d_out, m, in_stride=stride - a, out_stride=m + 3)
        y[:, wo * a: wo * b] = d_out.to_numpy().reshape(nch, wo * (m + 3))[:, : wo * m]
    from scipy.signal import fftconvolve
    for c in range(nch):
        xc = x[c, : w * n].astype(np.float64)
        xz = xc[0::2] + 1j * xc[1::2] if cplx else xc
        ref = fftconvolve(xz, taps.astype(np.complex128 if ctaps else np.float64))[:n]
        got = y[c].astype(np.float64)
        gz = got[0::2] + 1j * got[1::2] if wo == 2 else got
        err = np.sqrt(np.sum(np.abs(gz - ref) ** 2) / max(np.sum(np.abs(ref) ** 2), 1e-30))
        assert err <= 1e-5, (seed, n_taps, cplx, ctaps, nch, n, c, err)


@pytest.mark.parametrize("seed", range(12))
def test_rs_random_shapes_bit_exact(api, L, orc, seed):
    """Exact mode against the oracle (itself bit-exact with the compiled reference): random
    upsample / taps / rate (integer-valued and not) / chunkings, real and complex."""
    rng = np.random.default_rng(2000 + seed)
    U = int(rng.integers(1, 6))
    n_taps = int(rng.integers(U, 200))
    B = int(rng.choice([128, 1000, 1001, 4096]))
    if (n_taps + U - 1) // U > B:
        n_taps = U * (B // 2)
    taps = rng.standard_normal(n_taps).astype(np.float32)
    S = int(rng.integers(max(1, U), 4 * U + 2))
    rate = float(np.float32(S) / np.float32(U)) if seed % 2 == 0 else float(np.float32(rng.uniform(1.0, 5.0)))
    cplx = bool(rng.integers(0, 2))
    w = 2 if cplx else 1
    n = int(rng.choice([500, 4096, 9999, 30000]))
    x = synth.synth_f32(w * n, ch=100 + seed)
    r = api.Rs(taps, U, B, mode=L.RS_RESAMPLE, data_complex=cplx)
    r.set_exact(True)
    chunk = None if seed % 3 else int(rng.integers(1, 40)) * B       # bulk calls of whole blksize multiples
    y = r.resample_array(x, rate, chunk=chunk)[0]
    for part in range(w):
        ref, _ = orc.Resample(taps, U, B).stream(x[part::w], rate)
        got = y[part::w]
        assert len(ref) - len(got) in (0, 1), (seed, U, n_taps, B, rate, len(ref), len(got))
        assert np.array_equal(got, ref[: len(got)]), (seed, U, n_taps, B, rate, cplx, chunk)


@pytest.mark.parametrize("seed", range(40))
def test_general_rate_random_shapes_every_bit(api, L, orc, seed):
    """The general-rate kernel (poly_seg_kernel: an output's two sums run together over shared samples, taps four at a
    time from aligned rows, the last phase's second sum one sample on) against the oracle, exact mode, on rates below
    and above 1, up to eight phases, 1 to 40 taps per phase, streams that start in zero history -- compared as BIT
    PATTERNS (the sign of a zero included, which == would let through)."""
    rng = np.random.default_rng(5000 + seed)
    U = int(rng.integers(1, 9))
    plen = int(rng.integers(1, 41))
    n_taps = max(U, U * plen - int(rng.integers(0, U)))
    B = int(rng.choice([256, 1000, 1001, 4096]))
    taps = rng.standard_normal(n_taps).astype(np.float32)
    lo = 1.0 / U + 0.02
    rate = float(np.float32(rng.uniform(lo, 1.0) if seed % 4 == 0 and U > 1 else rng.uniform(1.0, 6.0)))
    cplx = bool(rng.integers(0, 2))
    w = 2 if cplx else 1
    n = int(rng.choice([700, 4096, 12345, 40000]))
    x = synth.synth_f32(w * n, ch=200 + seed)
    if seed % 5 == 0:
        x[: w * 300] = 0.0                    # a silent start: sums of signed zeros
    r = api.Rs(taps, U, B, mode=L.RS_RESAMPLE, data_complex=cplx)
    r.set_exact(True)
    chunk = int(rng.integers(1, 12)) * B       # bulk calls of whole blksize multiples: the run-length (poly_seg) path
    y = r.resample_array(x, rate, chunk=chunk)[0]
    for part in range(w):
        ref, _ = orc.Resample(taps, U, B).stream(x[part::w], rate)
        got = np.ascontiguousarray(y[part::w])
        assert len(ref) - len(got) in (0, 1), (seed, U, n_taps, B, rate, len(ref), len(got))
        assert np.array_equal(got.view(np.uint32), np.ascontiguousarray(ref[: len(got)]).view(np.uint32)), (seed, U, n_taps, B, rate, cplx, chunk)


@pytest.mark.parametrize("seed", range(48))
def test_integer_step_fused_kernels_random_shapes(api, L, orc, seed):
    """Round 5's kernels for integer-valued steps (poly_rt_dma.hip: the tile by LDS-DMA read in place; the register-window form for real
    interpolators) on RANDOM shapes: 1-8 phases, input steps 1-40, 1-70 taps per phase with ragged last rows, real and complex streams,
    1-3 channels, ragged cuts with carried state -- the same samples once from 16-byte-aligned channels (those kernels) and once shifted
    by one sample at an odd stride (the kernels they replace: poly_rt_kernel, poly_rt1_kernel, poly_tiled_kernel): the same number of
    outputs per call and the same BITS; and one channel against the oracle (libdsp/resample.cxx:100-114 at an integer step)."""
    rng = np.random.default_rng(7000 + seed)
    U = int(rng.integers(1, 9))
    step = 1 if seed % 3 == 0 else int(rng.integers(1, 41))
    plen = int(rng.integers(1, 71))
    n_taps = max(U, U * plen - int(rng.integers(0, U)))
    taps = (rng.standard_normal(n_taps) / np.sqrt(plen)).astype(np.float32)
    rate = float(np.float32(step) / np.float32(U))
    cplx = bool(rng.integers(0, 2))
    w = 2 if cplx else 1
    nch = int(rng.choice([1, 2, 3]))
    n = int(rng.choice([5000, 12345, 40000, 70001])) * max(1, step // 8)
    B = 4096
    x = np.stack([synth.synth_f32(w * n, ch=300 + seed * 4 + c) for c in range(nch)])
    cuts = sorted(set([0, n] + [int(v) // B * B for v in rng.integers(B, n, size=int(rng.integers(0, 3)))]))      # whole blocks: bulk calls
    cap = n * U // step + 64
    cap += (-cap) % 4

    def run(offset_samples, stride):
        buf = np.zeros(w * (offset_samples + stride * nch), np.float32)
        for c in range(nch):
            buf[w * (offset_samples + stride * c): w * (offset_samples + stride * c) + w * n] = x[c]
        d = api.DeviceArray.from_numpy(buf)
        d_out = api.DeviceArray(w * cap * nch)
        r = api.Rs(taps, U, B, mode=L.RS_RESAMPLE, data_complex=cplx, n_channels=nch)
        r.set_algo(L.RS_ALGO_DIRECT)
        ks, k = [], 0
        for a, b in zip(cuts[:-1], cuts[1:]):
            kk = r.process_stream(d.ptr + 4 * w * (offset_samples + a), b - a, d_out.ptr + 4 * w * k, cap - k, rate, in_stride=stride, out_stride=cap)
            ks.append(kk)
            k += kk
        return ks, d_out.to_numpy().reshape(nch, w * cap)[:, : w * k]

    al = (n + 3) // 4 * 4
    ks, y = run(0, al)
    ku, yu = run(1, al + 1)
    assert ks == ku, (seed, U, step, n_taps, cplx, nch, n, cuts)
    assert np.array_equal(y.view(np.uint32), yu.view(np.uint32)), (seed, U, step, n_taps, cplx, nch, n, cuts)
    c = nch - 1
    for part in range(w):
        ref, _ = orc.Resample(taps, U, B).stream(np.ascontiguousarray(x[c, part::w]), rate)
        got = y[c, part::w]
        assert len(ref) - len(got) in (0, 1), (seed, U, step, len(ref), len(got))
        assert synth.rel_rms(got, ref[: len(got)]) <= 1e-5, (seed, U, step, n_taps, cplx)


@pytest.mark.parametrize("seed", range(16))
def test_rs_fft_path_random_shapes(api, L, monkeypatch, seed):
    """Transform-domain kernel forced on, random instantiated (U, step) pairs, tap counts up to
    the longest overlap it accepts (beyond that the call silently takes the direct kernel), stream
    lengths around segment/pass boundaries, random chunkings and channel counts -- against the
    exact-mode kernels (bit-exact with the compiled reference, tested elsewhere)."""
    rng = np.random.default_rng(3000 + seed)
    U, S = [(3, 5), (2, 3), (2, 5), (3, 4), (1, 2), (1, 3), (1, 4), (1, 5), (1, 6), (1, 7), (1, 8), (2, 7), (4, 5),
            (6, 10), (2, 4), (3, 9)][seed]
    g = int(np.gcd(U, S))
    SP = S // g
    n_taps = int(rng.integers(U, 192 * SP * U + 40))
    taps = (rng.standard_normal(n_taps) / np.sqrt(n_taps)).astype(np.float32)
    nch = int(rng.integers(1, 4))
    n = int(rng.choice([7000 * SP, 231 * SP * 64 + 1, 100003, 180000]))
    x = np.stack([synth.synth_cf32(n, ch=50 + seed * 4 + c) for c in range(nch)])
    rate = float(np.float32(S) / np.float32(U))
    cuts = sorted(set([0, n] + [int(v) for v in rng.integers(1, n, size=int(rng.integers(0, 3)))]))
    outs = {}
    for exact in (False, True):
        r = api.Rs(taps, U, 4096, mode=L.RS_RESAMPLE, data_complex=True, n_channels=nch, algo=L.RS_ALGO_FFT)
        r.set_exact(exact)
        parts = []
        for a, b in zip(cuts[:-1], cuts[1:]):
            parts.append(r.resample_array(np.ascontiguousarray(x[:, 2 * a: 2 * b]), rate))
        outs[exact] = np.concatenate(parts, axis=1)
        r.close()
    assert outs[False].shape == outs[True].shape, (U, S, n_taps, n, cuts)
    for c in range(nch):
        assert synth.rel_rms(outs[False][c], outs[True][c]) <= 1e-5, (U, S, n_taps, n, nch, cuts, c)


@pytest.mark.parametrize("seed", range(10))
def test_blkconv_class_vs_compiled_reference_random(api, orc, seed):
    """The drop-in blkconv class beside the reference's own (blkconv.cxx on ROCm's libhipfftw),
    run live on random tap counts and fft lengths -- powers of two, even and odd lengths, taps from
    1 to most of the block -- through the same get_process_buf()/process() calls; and the CPU
    restatement beside both."""
    if orc.ref_blkconv_lib() is None:
        pytest.skip("oracle/_ref/libsferef_blkconv.so not prebuilt")
    rng = np.random.default_rng(4000 + seed)
    fft_len = int(rng.choice([32, 64, 100, 255, 1000, 1024, 2048, 4096, 5000, 16384]))
    n_taps = int(rng.integers(1, max(2, fft_len // 2)))
    taps = (rng.standard_normal(n_taps) / np.sqrt(n_taps)).astype(np.float32)
    ref = orc.RefBlkconv(taps, fft_len)
    dut = api.blkconv(taps, fft_len)
    cpu = orc.Blkconv(taps, fft_len)
    blk = ref.blk
    assert dut.get_blksize() == blk == cpu.blk == fft_len + 1 - n_taps
    nblk = int(rng.integers(2, 7))
    x = synth.synth_f32(nblk * blk, ch=200 + seed)
    buf = dut.get_process_buf()
    got, want, rest = np.empty_like(x), np.empty_like(x), np.empty_like(x)
    for off in range(0, len(x), blk):
        for obj, b, dst in ((ref, ref.buf, want), (dut, buf, got), (cpu, cpu.buf, rest)):
            b[:blk] = x[off: off + blk]
            obj.process()
            dst[off: off + blk] = b[:blk]
    assert synth.rel_rms(got, want) < 2e-6, (fft_len, n_taps)
    assert synth.rel_rms(rest, want) < 2e-6, (fft_len, n_taps)


def test_two_thousand_launches_on_one_handle_keep_the_work_counters_clean(api, L):
    """The FFT FIR kernel and the transform-domain resampler hand work to their persistent
    workgroups through device counters that every launch must leave at zero (the launch's last
    draw resets them).  2000 launches of random lengths -- one transform, fewer transforms than
    counter groups, thousands of transforms; 1-3 channels -- on ONE handle each: the concatenated
    outputs still equal one pass over the whole stream.  A counter left dirty would make the next
    launch skip transforms."""
    rng = np.random.default_rng(77)
    taps = synth.taps_cfg2()
    for nch in (1, 3):
        lens = [int(v) for v in rng.choice([1, 17, 3839, 3840, 3841, 9000, 30721, 100000], size=1000)]
        n = sum(lens)
        x = np.stack([synth.synth_cf32(n, ch=30 + c) for c in range(nch)])
        whole = api.Fir(taps, data_complex=True, n_channels=nch, algo=L.FIR_ALGO_FFT).filter(x)
        f = api.Fir(taps, data_complex=True, n_channels=nch, algo=L.FIR_ALGO_FFT)
        d_in = api.DeviceArray.from_numpy(x)
        d_out = api.DeviceArray(2 * n * nch)
        off = 0
        for m in lens:                                       # every launch: its own slice of the resident stream
            f.process_stream(d_in.ptr + 8 * off, d_out.ptr + 8 * off, m, in_stride=n, out_stride=n)
            off += m
        got = d_out.to_numpy().reshape(nch, 2 * n)
        assert synth.rel_rms(got, whole) <= 2e-6
        d_in.free()
        d_out.free()
    # the resampler's counters: 300 launches of the 5/3 shape, bulk default kernel
    taps3 = synth.taps_cfg3()
    lens = [int(v) for v in rng.choice([4096 * 5, 23105, 50000, 200000], size=300)]
    n = sum(lens)
    x = synth.synth_cf32(n, ch=40)
    r0 = api.Rs(taps3, 3, 4096, mode=L.RS_RESAMPLE, data_complex=True)
    whole = r0.resample_array(x[None, :], 5.0 / 3.0)[0]
    r = api.Rs(taps3, 3, 4096, mode=L.RS_RESAMPLE, data_complex=True)
    d_in = api.DeviceArray.from_numpy(x)
    d_out = api.DeviceArray(2 * (n * 3 // 5 + 16))
    off = k = 0
    for m in lens:
        k += r.process_stream(d_in.ptr + 8 * off, m, d_out.ptr + 8 * k, m * 3 // 5 + 8, 5.0 / 3.0)
        off += m
    assert abs(k - len(whole) // 2) <= 1
    got = d_out.to_numpy(2 * min(k, len(whole) // 2))
    assert synth.rel_rms(got, whole[: len(got)]) <= 2e-6
