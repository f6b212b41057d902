"""Capturable streams: ONE *_process_stream call captured into a hipGraph,
replayed over consecutive chunks of a stream, equals the eager stream bit for bit -- the carried
state (history; the resamplers' time state) lives on the device or does not move between replays
(api_fir.hip: stream_is_capturing).  The reference's carried state: libdsp/blkconv.h:56 (m_overlap),
libdsp/resample.h:49-59 (m_history, m_pos, m_mu, m_is_leftover)."""
import ctypes as C

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


class Hip:
    """The five HIP graph calls a capturing caller makes, through libamdhip64 (the product library is not involved)."""

    def __init__(self):
        self.h = C.CDLL("libamdhip64.so")
        for name, args in (("hipStreamCreate", [C.POINTER(C.c_void_p)]),
                           ("hipStreamBeginCapture", [C.c_void_p, C.c_int]),
                           ("hipStreamEndCapture", [C.c_void_p, C.POINTER(C.c_void_p)]),
                           ("hipGraphInstantiate", [C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
                           ("hipGraphLaunch", [C.c_void_p, C.c_void_p]),
                           ("hipStreamSynchronize", [C.c_void_p]),
                           ("hipGraphExecDestroy", [C.c_void_p]), ("hipGraphDestroy", [C.c_void_p]),
                           ("hipStreamDestroy", [C.c_void_p])):
            fn = getattr(self.h, name)
            fn.argtypes, fn.restype = args, C.c_int

    def ok(self, rc):
        assert rc == 0, f"HIP error {rc}"

    def capture(self, body):
        s = C.c_void_p()
        self.ok(self.h.hipStreamCreate(C.byref(s)))
        self.ok(self.h.hipStreamBeginCapture(s, 0))          # hipStreamCaptureModeGlobal
        try:
            body(s.value)
        finally:
            g = C.c_void_p()
            rc = self.h.hipStreamEndCapture(s, C.byref(g))
        self.ok(rc)
        ex = C.c_void_p()
        self.ok(self.h.hipGraphInstantiate(C.byref(ex), g, None, None, 0))
        return s, g, ex

    def launch(self, ex, s):
        self.ok(self.h.hipGraphLaunch(ex, s))

    def sync(self, s):
        self.ok(self.h.hipStreamSynchronize(s))

    def free(self, s, g, ex):
        self.h.hipGraphExecDestroy(ex)
        self.h.hipGraphDestroy(g)
        self.h.hipStreamDestroy(s)


def _h2d(api, L, dst, arr, stream):
    api.check(L.load().sfe_dsp_memcpy_h2d(dst.ptr, arr.ctypes.data, arr.nbytes, stream))


@pytest.mark.parametrize("n_taps,nch", [(256, 1), (256, 3), (3000, 1)])
def test_fir_call_captured_once_replays_the_next_chunks(api, L, n_taps, nch):
    hip = Hip()
    taps = synth.taps_cfg2() if n_taps == 256 else synth.lowpass_taps(n_taps, 0.1).astype(np.float32)
    n, reps = 3840 * 6 + 512, 8
    x = np.stack([synth.synth_cf32(n * (reps + 1), ch=40 + c) for c in range(nch)])       # (nch, 2 n (reps + 1))
    chunk = lambda i: np.ascontiguousarray(x[:, 2 * n * i: 2 * n * (i + 1)])
    # eager stream: one handle, reps + 1 consecutive calls
    fe = api.Fir(taps, data_complex=True, n_channels=nch)
    d_in, d_out = api.DeviceArray(2 * n * nch), api.DeviceArray(2 * n * nch)
    want = []
    for i in range(reps + 1):
        _h2d(api, L, d_in, chunk(i), None)
        fe.process_stream(d_in, d_out, n)
        want.append(d_out.to_numpy())
    # captured: chunk 0 eagerly (tables, first state), then ONE captured call replayed for chunks 1..reps
    fg = api.Fir(taps, data_complex=True, n_channels=nch)
    g_in, g_out = api.DeviceArray(2 * n * nch), api.DeviceArray(2 * n * nch)
    _h2d(api, L, g_in, chunk(0), None)
    fg.process_stream(g_in, g_out, n)
    api.sync()
    assert np.array_equal(g_out.to_numpy(), want[0])
    s, g, ex = hip.capture(lambda st: fg.process_stream(g_in, g_out, n, stream=st))
    for i in range(1, reps + 1):
        _h2d(api, L, g_in, chunk(i), s.value)
        hip.launch(ex, s)
        hip.sync(s)
        assert np.array_equal(g_out.to_numpy(stream=s.value), want[i]), i
    hip.free(s, g, ex)
    # and eager calls may continue the same stream behind the replays (the host-side bookkeeping never moved)
    fe2 = api.Fir(taps, data_complex=True, n_channels=nch)
    for i in range(reps + 1):
        _h2d(api, L, d_in, chunk(i), None)
        fe2.process_stream(d_in, d_out, n)
    extra = np.stack([synth.synth_cf32(n, ch=90 + c) for c in range(nch)])
    _h2d(api, L, d_in, extra, None)
    fe2.process_stream(d_in, d_out, n)
    _h2d(api, L, g_in, extra, None)
    fg.process_stream(g_in, g_out, n)
    api.sync()
    assert np.array_equal(g_out.to_numpy(), d_out.to_numpy())


def test_graph_replays_and_eager_calls_interleave_on_one_handle(api, L):
    """a captured call names the history buffer of the moment; an eager call used to move
    the handle to the other buffer, after which a replay read (and wrote) stale state.  Now a handle
    that has been captured keeps its state where the graph expects it: replays, eager calls, a call
    shorter than the history and load_history interleave and equal the all-eager stream bit for bit."""
    hip = Hip()
    taps = synth.taps_cfg2()
    n = 3840 * 5 + 100
    plan = ["e", "g", "g", "e", "g", "s", "g", "e", "e", "g"]           # eager / graph replay / short eager call (100 < hl)
    lens = [100 if k == "s" else n for k in plan]
    x = synth.synth_cf32(sum(lens), ch=61)
    fe = api.Fir(taps, data_complex=True)
    want, off = [], 0
    d_in, d_out = api.DeviceArray(2 * n), api.DeviceArray(2 * n)
    for m in lens:
        _h2d(api, L, d_in, np.ascontiguousarray(x[2 * off: 2 * (off + m)]), None)
        fe.process_stream(d_in, d_out, m)
        want.append(d_out.to_numpy(2 * m))
        off += m
    fg = api.Fir(taps, data_complex=True)
    g_in, g_out = api.DeviceArray(2 * n), api.DeviceArray(2 * n)
    s = g = ex = None
    off = 0
    for i, (kind, m) in enumerate(zip(plan, lens)):
        chunk = np.ascontiguousarray(x[2 * off: 2 * (off + m)])
        if kind == "g":
            if ex is None:
                s, g, ex = hip.capture(lambda st: fg.process_stream(g_in, g_out, n, stream=st))
            _h2d(api, L, g_in, chunk, s.value)
            hip.launch(ex, s)
            hip.sync(s)
        else:
            _h2d(api, L, g_in, chunk, None)
            fg.process_stream(g_in, g_out, m)
            api.sync()
        assert np.array_equal(g_out.to_numpy(2 * m), want[i]), (i, kind)
        off += m
    hip.free(s, g, ex)

    # the same for a resampler handle (integer-valued step)
    U, S = 3, 5
    rate = float(np.float32(S) / np.float32(U))
    n = S * 231 * 40
    cap = n * U // S + 8
    plan = ["e", "g", "e", "g", "g", "e", "g"]
    xr = synth.synth_cf32(n * len(plan), ch=62)
    re_ = api.Rs(synth.taps_cfg3(), U, 4096, mode=L.RS_RESAMPLE, data_complex=True)
    d_in, d_out = api.DeviceArray(2 * n), api.DeviceArray(2 * cap)
    want = []
    for i in range(len(plan)):
        _h2d(api, L, d_in, np.ascontiguousarray(xr[2 * n * i: 2 * n * (i + 1)]), None)
        k = re_.process_stream(d_in, n, d_out, cap, rate)
        want.append(d_out.to_numpy(2 * k))
    rg = api.Rs(synth.taps_cfg3(), U, 4096, mode=L.RS_RESAMPLE, data_complex=True)
    g_in, g_out = api.DeviceArray(2 * n), api.DeviceArray(2 * cap)
    s = g = ex = None
    ks = []
    for i, kind in enumerate(plan):
        chunk = np.ascontiguousarray(xr[2 * n * i: 2 * n * (i + 1)])
        if kind == "g":
            if ex is None:
                s, g, ex = hip.capture(lambda st: ks.append(rg.process_stream(g_in, n, g_out, cap, rate, stream=st)))
            _h2d(api, L, g_in, chunk, s.value)
            hip.launch(ex, s)
            hip.sync(s)
            k = ks[0]
        else:
            _h2d(api, L, g_in, chunk, None)
            k = rg.process_stream(g_in, n, g_out, cap, rate)
            api.sync()
        assert np.array_equal(g_out.to_numpy(2 * k), want[i]), (i, kind)
    hip.free(s, g, ex)


@pytest.mark.parametrize("which,U,S,n_taps,exact", [("resample", 3, 5, 381, False), ("resample", 3, 5, 381, True),
                                                    ("decimate", 1, 8, 64, False), ("decimate", 1, 8, 64, True)])
def test_rs_call_captured_once_replays_the_next_chunks(api, L, which, U, S, n_taps, exact):
    hip = Hip()
    taps = synth.taps_cfg3() if which == "resample" else synth.taps_cfg4()
    rate = float(np.float32(S) / np.float32(U))
    n, reps = S * 231 * 40, 8                              # n*U a multiple of the step: the time state does not move
    x = synth.synth_cf32(n * (reps + 1), ch=7)
    chunk = lambda i: np.ascontiguousarray(x[2 * n * i: 2 * n * (i + 1)])
    cap = n * U // S + 8
    mode = L.RS_RESAMPLE if which == "resample" else L.RS_DECIMATE
    re_ = api.Rs(taps, U, 4096, mode=mode, data_complex=True)
    re_.set_exact(exact)
    d_in, d_out = api.DeviceArray(2 * n), api.DeviceArray(2 * cap)
    want = []
    for i in range(reps + 1):
        _h2d(api, L, d_in, chunk(i), None)
        k = re_.process_stream(d_in, n, d_out, cap, rate)
        want.append(d_out.to_numpy(2 * k))
    rg = api.Rs(taps, U, 4096, mode=mode, data_complex=True)
    rg.set_exact(exact)
    g_in, g_out = api.DeviceArray(2 * n), api.DeviceArray(2 * cap)
    _h2d(api, L, g_in, chunk(0), None)
    k0 = rg.process_stream(g_in, n, g_out, cap, rate)
    api.sync()
    assert np.array_equal(g_out.to_numpy(2 * k0), want[0])
    ks = []
    s, g, ex = hip.capture(lambda st: ks.append(rg.process_stream(g_in, n, g_out, cap, rate, stream=st)))
    for i in range(1, reps + 1):
        _h2d(api, L, g_in, chunk(i), s.value)
        hip.launch(ex, s)
        hip.sync(s)
        assert ks[0] == len(want[i]) // 2
        assert np.array_equal(g_out.to_numpy(2 * ks[0], stream=s.value), want[i]), i
    hip.free(s, g, ex)


def test_calls_whose_state_would_move_refuse_to_be_captured(api, L):
    """A captured call whose replay would need host-side state is refused loudly, not replayed wrongly:
    a resampler call that leaves the time state elsewhere than it found it, a FIR call shorter than
    the history."""
    hip = Hip()
    r = api.Rs(synth.taps_cfg3(), 3, 4096, mode=L.RS_RESAMPLE, data_complex=True)
    f = api.Fir(synth.taps_cfg2(), data_complex=True)
    n = 5 * 231 * 40 + 1                                   # n*U is not a multiple of the step 5
    d_in, d_out = api.DeviceArray(2 * n), api.DeviceArray(2 * n)
    d_in.zero()
    r.process_stream(d_in, n - 1, d_out, n, 5.0 / 3.0)     # eager first: plans and tables exist
    api.sync()
    codes = []
    s = C.c_void_p()
    hip.ok(hip.h.hipStreamCreate(C.byref(s)))
    hip.ok(hip.h.hipStreamBeginCapture(s, 2))              # relaxed mode: the refused calls launch nothing
    try:
        for call in (lambda: r.process_stream(d_in, n, d_out, n, 5.0 / 3.0, stream=s.value),
                     lambda: f.process_stream(d_in, d_out, 100, stream=s.value)):
            try:
                call()
                codes.append(L.SFE_OK)
            except api.SfeError as e:
                codes.append(e.code)
    finally:
        g = C.c_void_p()
        hip.h.hipStreamEndCapture(s, C.byref(g))
    assert codes == [L.SFE_ESTATE, L.SFE_ESTATE], codes
    if g.value:
        hip.h.hipGraphDestroy(g)
    hip.h.hipStreamDestroy(s)
