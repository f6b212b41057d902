"""Python face of libsfe_dsp.so.

Two layers, both thin:
  * `blkconv`, `resample`, `decimate` -- the reference's SWIG projection
    (libdsp/test/pydsp.i:16-22): same class names, same constructor arguments, and
    `process(in_array, out_len, rate) -> (n_out, out_array)` for the resamplers, so the
    reference's driver scripts (libdsp/test/test_decimate.py:22-25) read the same.
  * `Fir`, `Rs`, `DeviceArray` -- the device-resident bulk path used by bench.py and the
    parity tests (sfe_dsp_*_process_stream).

Everything computes on the GPU through the C ABI; numpy is only the host container.
"""
import ctypes as C

import numpy as np

from . import lib as _l
from .lib import SfeError, check  # noqa: F401


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def device_count():
    n = C.c_int(0)
    check(_l.load().sfe_dsp_device_count(C.byref(n)))
    return n.value


def sync(stream=None):
    check(_l.load().sfe_dsp_sync(stream))


class DeviceArray:
    """A flat float32 device buffer (sfe_dsp_malloc / sfe_dsp_free)."""

    def __init__(self, n_floats):
        self._L = _l.load()
        self.n = int(n_floats)
        p = C.c_void_p()
        check(self._L.sfe_dsp_malloc(C.byref(p), self.n * 4))
        self.ptr = p.value

    @classmethod
    def from_numpy(cls, a, stream=None):
        a = _f32(a).ravel()
        d = cls(a.size)
        check(d._L.sfe_dsp_memcpy_h2d(d.ptr, a.ctypes.data, a.nbytes, stream))
        check(d._L.sfe_dsp_sync(stream))
        return d

    @classmethod
    def from_bytes(cls, b, stream=None):
        """Device copy of a uint8 array (the u8 wire format), padded to whole floats."""
        b = np.ascontiguousarray(b, dtype=np.uint8).ravel()
        d = cls((b.size + 3) // 4 + 4)
        check(d._L.sfe_dsp_memcpy_h2d(d.ptr, b.ctypes.data, b.nbytes, stream))
        check(d._L.sfe_dsp_sync(stream))
        return d

    def to_numpy(self, n_floats=None, offset=0, stream=None):
        n = self.n - offset if n_floats is None else int(n_floats)
        out = np.empty(n, dtype=np.float32)
        check(self._L.sfe_dsp_memcpy_d2h(out.ctypes.data, self.ptr + 4 * int(offset), out.nbytes, stream))
        check(self._L.sfe_dsp_sync(stream))
        return out

    def fill_synth(self, seed, channel=0, first=0, n_floats=None, offset=0, stream=None):
        n = self.n - offset if n_floats is None else int(n_floats)
        check(self._L.sfe_dsp_synth_fill(self.ptr + 4 * int(offset), n, seed, channel, first, stream))

    def zero(self, stream=None):
        check(self._L.sfe_dsp_memset(self.ptr, 0, self.n * 4, stream))

    def free(self):
        if getattr(self, "ptr", None):
            self._L.sfe_dsp_free(self.ptr)
            self.ptr = None

    __del__ = free


class Timer:
    """HIP events on the stream the kernels are launched on (bench.py roofline leg)."""

    def __init__(self):
        self._L = _l.load()
        p = C.c_void_p()
        check(self._L.sfe_dsp_timer_create(C.byref(p)))
        self._h = p.value

    def start(self, stream=None):
        check(self._L.sfe_dsp_timer_start(self._h, stream))

    def stop(self, stream=None):
        check(self._L.sfe_dsp_timer_stop(self._h, stream))

    def elapsed_ms(self):
        ms = C.c_float(0)
        check(self._L.sfe_dsp_timer_elapsed_ms(self._h, C.byref(ms)))
        return ms.value

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.sfe_dsp_timer_destroy(self._h)
            self._h = None


class Fir:
    """One FIR stream set (sfe_dsp_fir_*): taps real (n,) or complex (n,) complex64."""

    def __init__(self, taps, data_complex=True, n_channels=1, block_hint=0, device=0, algo=_l.FIR_ALGO_AUTO,
                 per_channel=False):
        self._L = _l.load()
        taps = np.asarray(taps)
        self.taps_complex = bool(np.iscomplexobj(taps))
        if per_channel:
            # taps: (n_channels, n_taps), one filter per channel (sfe_dsp_fir_create_per_channel); cf32 streams
            assert taps.ndim == 2
            n_channels, n_taps = taps.shape
            t = (np.ascontiguousarray(taps.astype(np.complex64)).view(np.float32) if self.taps_complex
                 else _f32(taps)).reshape(-1)
            self.n_taps, self.data_complex, self.out_complex, self.n_channels = n_taps, True, True, n_channels
            h = C.c_void_p()
            check(self._L.sfe_dsp_fir_create_per_channel(t.ctypes.data, n_taps, int(self.taps_complex), n_channels,
                                                         device, C.byref(h)))
            self._h = h.value
            return
        if self.taps_complex:
            t = np.ascontiguousarray(taps.astype(np.complex64)).view(np.float32)
            n_taps = t.size // 2
        else:
            t = _f32(taps)
            n_taps = t.size
        self.n_taps = n_taps
        self.data_complex = bool(data_complex)
        self.out_complex = self.data_complex or self.taps_complex
        self.n_channels = n_channels
        h = C.c_void_p()
        check(self._L.sfe_dsp_fir_create(t.ctypes.data, n_taps, int(self.taps_complex), int(self.data_complex),
                                         n_channels, block_hint, device, C.byref(h)))
        self._h = h.value
        if algo != _l.FIR_ALGO_AUTO:
            self.set_algo(algo)

    def set_algo(self, algo):
        check(self._L.sfe_dsp_fir_set_algo(self._h, algo))

    def set_variant(self, variant):
        """lib.FIR_VARIANT_*: fix the cf32 kernel's data-movement variant (AUTO: what calibrate() chose for the
        shape, register loads where nothing was calibrated)."""
        check(self._L.sfe_dsp_fir_set_variant(self._h, int(variant)))

    def get_variant(self):
        """(variant of the last bulk call, calibrations made by this handle, [median ms per variant])."""
        v, k = C.c_int(-1), C.c_int(0)
        ms = (C.c_float * 3)()
        check(self._L.sfe_dsp_fir_get_variant(self._h, C.byref(v), C.byref(k), ms))
        return v.value, k.value, [float(m) for m in ms]

    def calibrate(self, d_in, d_out, n, in_stride=None, out_stride=None, stream=None):
        """sfe_dsp_fir_calibrate: time the kernel's variants over these buffers (synchronous; the stream
        does not advance) and remember the choice for this shape; returns the chosen variant."""
        pi = d_in.ptr if isinstance(d_in, DeviceArray) else int(d_in)
        po = d_out.ptr if isinstance(d_out, DeviceArray) else int(d_out)
        v = C.c_int(0)
        check(self._L.sfe_dsp_fir_calibrate(self._h, pi, po, n, n if in_stride is None else in_stride,
                                            n if out_stride is None else out_stride, stream, C.byref(v)))
        return v.value

    def set_zero_copy_max(self, max_samples):
        check(self._L.sfe_dsp_fir_set_zero_copy_max(self._h, int(max_samples)))

    def set_input_format(self, fmt):
        """lib.FMT_U8: process_stream reads u8 offset-binary samples (fused RX converter)."""
        check(self._L.sfe_dsp_fir_set_input_format(self._h, fmt))

    def set_output_format(self, fmt):
        """lib.FMT_TX10: process_stream writes 10-bit packed bytes (fused TX converter, real streams)."""
        check(self._L.sfe_dsp_fir_set_output_format(self._h, fmt))

    def reset(self):
        check(self._L.sfe_dsp_fir_reset(self._h))

    def load_history(self, d_prev, n_prev, stride=None, stream=None):
        """Carried state from the n_prev samples that precede the next call's input (one stream cut
        into spans): d_prev a DeviceArray or raw device pointer, float32 elements."""
        p = d_prev.ptr if isinstance(d_prev, DeviceArray) else int(d_prev or 0)
        check(self._L.sfe_dsp_fir_load_history(self._h, p, n_prev, n_prev if stride is None else stride, stream))

    def process_stream(self, d_in, d_out, n, in_stride=None, out_stride=None, stream=None):
        """d_in/d_out: DeviceArray or raw device pointers; n samples per channel."""
        pi = d_in.ptr if isinstance(d_in, DeviceArray) else int(d_in)
        po = d_out.ptr if isinstance(d_out, DeviceArray) else int(d_out)
        check(self._L.sfe_dsp_fir_process_stream(self._h, pi, po, n, n if in_stride is None else in_stride,
                                                 n if out_stride is None else out_stride, stream))

    def filter(self, x):
        """Host convenience: x is (n_channels, n*) float32 with interleaved I/Q when complex."""
        x = _f32(x).reshape(self.n_channels, -1)
        n = x.shape[1] // (2 if self.data_complex else 1)
        d_in = DeviceArray.from_numpy(x)
        d_out = DeviceArray(self.n_channels * n * (2 if self.out_complex else 1))
        self.process_stream(d_in, d_out, n)
        y = d_out.to_numpy().reshape(self.n_channels, -1)
        d_in.free()
        d_out.free()
        return y

    def host_buffer(self):
        p = C.POINTER(C.c_float)()
        blk = C.c_int(0)
        check(self._L.sfe_dsp_fir_host_buffer(self._h, C.byref(p), C.byref(blk)))
        width = 2 if self.out_complex else 1
        return np.ctypeslib.as_array(p, shape=(blk.value * width,)), blk.value

    def process_block(self):
        check(self._L.sfe_dsp_fir_process_block(self._h))

    def close(self):
        """sfe_dsp_fir_destroy.  A handle a pipe still borrows is NOT destroyed (SFE_ESTATE): it is kept,
        with a warning, rather than dropped on the floor with its GPU buffers."""
        if getattr(self, "_h", None):
            rc = self._L.sfe_dsp_fir_destroy(self._h)
            if rc == _l.SFE_ESTATE:
                import warnings
                warnings.warn("Fir.close: a pipe still borrows this handle (sfe_dsp_pipe_destroy first); handle kept",
                              ResourceWarning, stacklevel=2)
                return
            self._h = None

    __del__ = close


class _Group:
    """Channel blocks of one multi-channel job on several devices of THIS process (sfe_dsp_*_group_*;
    the partition of shard.channel_block made inside the library)."""
    _prefix = None

    def _fn(self, name):
        return getattr(self._L, "sfe_dsp_%s_group_%s" % (self._prefix, name))

    def shards(self):
        """[(device, first_channel, n_channels, handle, stream), ...]"""
        n = C.c_int(0)
        check(self._fn("shards")(self._g, C.byref(n)))
        out = []
        for k in range(n.value):
            d, f, c, h, s = C.c_int(0), C.c_int(0), C.c_int(0), C.c_void_p(), C.c_void_p()
            check(self._fn("shard")(self._g, k, C.byref(d), C.byref(f), C.byref(c), C.byref(h), C.byref(s)))
            out.append((d.value, f.value, c.value, h.value, s.value))
        return out

    @staticmethod
    def _ptrs(arrs):
        return (C.c_void_p * len(arrs))(*[a.ptr if isinstance(a, DeviceArray) else int(a) for a in arrs])

    def sync(self):
        check(self._fn("sync")(self._g))

    def reset(self):
        check(self._fn("reset")(self._g))

    def close(self):
        if getattr(self, "_g", None):
            self._fn("destroy")(self._g)
            self._g = None

    __del__ = close


class FirGroup(_Group):
    _prefix = "fir"

    def __init__(self, taps, n_channels, devices, data_complex=True, per_channel=False):
        self._L = _l.load()
        taps = np.asarray(taps)
        ctaps = bool(np.iscomplexobj(taps))
        t = (np.ascontiguousarray(taps.astype(np.complex64)).view(np.float32) if ctaps else _f32(taps)).reshape(-1)
        n_taps = taps.shape[-1]
        dv = (C.c_int * len(devices))(*[int(d) for d in devices])
        g = C.c_void_p()
        check(self._L.sfe_dsp_fir_group_create(t.ctypes.data, n_taps, int(ctaps), int(data_complex), int(per_channel),
                                               n_channels, dv, len(devices), C.byref(g)))
        self._g = g.value

    def process_stream(self, d_in, d_out, n, in_stride=None, out_stride=None):
        """d_in / d_out: one DeviceArray (or raw device pointer) per shard; asynchronous on the shards' streams."""
        check(self._L.sfe_dsp_fir_group_process_stream(self._g, self._ptrs(d_in), self._ptrs(d_out), n,
                                                       n if in_stride is None else in_stride,
                                                       n if out_stride is None else out_stride))


class RsGroup(_Group):
    _prefix = "rs"

    def __init__(self, taps, upsample, blksize, n_channels, devices, mode=_l.RS_RESAMPLE, data_complex=True):
        self._L = _l.load()
        t = _f32(taps)
        dv = (C.c_int * len(devices))(*[int(d) for d in devices])
        g = C.c_void_p()
        check(self._L.sfe_dsp_rs_group_create(t.ctypes.data, t.size, upsample, blksize, int(data_complex), n_channels,
                                              dv, len(devices), mode, C.byref(g)))
        self._g = g.value

    def process_stream(self, d_in, n_in, d_out, out_cap, rate, in_stride=None, out_stride=None):
        k = C.c_size_t(0)
        check(self._L.sfe_dsp_rs_group_process_stream(self._g, self._ptrs(d_in), n_in, n_in if in_stride is None else in_stride,
                                                      self._ptrs(d_out), out_cap, out_cap if out_stride is None else out_stride,
                                                      rate, C.byref(k)))
        return k.value


class Rs:
    """One resample/decimate stream set (sfe_dsp_rs_*)."""

    def __init__(self, taps, upsample, blksize, mode=_l.RS_RESAMPLE, data_complex=False, n_channels=1, device=0,
                 algo=_l.RS_ALGO_AUTO):
        self._L = _l.load()
        t = _f32(taps)
        self.data_complex = bool(data_complex)
        self.n_channels = n_channels
        self.upsample = upsample
        self.blksize = blksize
        h = C.c_void_p()
        check(self._L.sfe_dsp_rs_create(t.ctypes.data, t.size, upsample, blksize, int(self.data_complex),
                                        n_channels, device, mode, C.byref(h)))
        self._h = h.value
        if algo != _l.RS_ALGO_AUTO:
            self.set_algo(algo)

    def set_exact(self, exact=True):
        check(self._L.sfe_dsp_rs_set_exact(self._h, int(bool(exact))))

    def set_algo(self, algo):
        """lib.RS_ALGO_AUTO / DIRECT / FFT / MFMA: the kernel of the integer-step bulk path."""
        check(self._L.sfe_dsp_rs_set_algo(self._h, algo))

    def set_input_format(self, fmt):
        check(self._L.sfe_dsp_rs_set_input_format(self._h, fmt))

    def reset(self):
        check(self._L.sfe_dsp_rs_reset(self._h))

    def load_history(self, d_prev, n_prev, stride=None, stream=None):
        p = d_prev.ptr if isinstance(d_prev, DeviceArray) else int(d_prev or 0)
        check(self._L.sfe_dsp_rs_load_history(self._h, p, n_prev, n_prev if stride is None else stride, stream))

    def seek(self, first_sample, rate):
        """Time state of a reference object that has consumed `first_sample` samples (integer-valued
        steps only; raises SfeError(SFE_ESTATE) otherwise)."""
        check(self._L.sfe_dsp_rs_seek(self._h, int(first_sample), float(rate)))

    def get_state(self):
        st = _l.TimeState()
        check(self._L.sfe_dsp_rs_get_state(self._h, C.byref(st)))
        return st

    def set_state(self, st):
        check(self._L.sfe_dsp_rs_set_state(self._h, C.byref(st)))

    def process(self, x, out_len, rate):
        """Host-pointer call == {resample,decimate}::process; returns (n_out, out_array)."""
        x = _f32(x)
        w = 2 if self.data_complex else 1
        n_in = x.size // w
        out = np.zeros((out_len + 1) * w, dtype=np.float32)
        n = C.c_int(0)
        check(self._L.sfe_dsp_rs_process(self._h, x.ctypes.data, n_in, out.ctypes.data, out_len, rate, C.byref(n)))
        return n.value, out[: out_len * w]

    def process_stream(self, d_in, n_in, d_out, out_cap, rate, in_stride=None, out_stride=None, stream=None):
        pi = d_in.ptr if isinstance(d_in, DeviceArray) else int(d_in)
        po = d_out.ptr if isinstance(d_out, DeviceArray) else int(d_out)
        n = C.c_size_t(0)
        check(self._L.sfe_dsp_rs_process_stream(self._h, pi, n_in, n_in if in_stride is None else in_stride, po,
                                                out_cap, out_cap if out_stride is None else out_stride, rate,
                                                C.byref(n), stream))
        return n.value

    def resample_array(self, x, rate, chunk=None):
        """Host convenience for tests: feed x (n_channels, n*) through process_stream in
        `chunk`-sample calls (default: all at once); returns (n_channels, n_out*)."""
        x = _f32(x).reshape(self.n_channels, -1)
        w = 2 if self.data_complex else 1
        n = x.shape[1] // w
        chunk = chunk or n
        d_in = DeviceArray(self.n_channels * chunk * w)
        cap = int(np.ceil(chunk / min(rate, 1e9))) + 4
        d_out = DeviceArray(self.n_channels * cap * w)
        outs = []
        for off in range(0, n, chunk):
            m = min(chunk, n - off)
            seg = np.ascontiguousarray(x[:, off * w:(off + m) * w])
            check(self._L.sfe_dsp_memcpy_h2d(d_in.ptr, seg.ctypes.data, seg.nbytes, None))
            k = self.process_stream(d_in, m, d_out, cap, rate, in_stride=m, out_stride=cap)
            y = d_out.to_numpy(self.n_channels * cap * w).reshape(self.n_channels, cap * w)[:, : k * w]
            outs.append(y.copy())
        d_in.free()
        d_out.free()
        return np.concatenate(outs, axis=1) if outs else np.zeros((self.n_channels, 0), np.float32)

    def close(self):
        """sfe_dsp_rs_destroy; a handle a pipe still borrows is kept, with a warning (see Fir.close)."""
        if getattr(self, "_h", None):
            rc = self._L.sfe_dsp_rs_destroy(self._h)
            if rc == _l.SFE_ESTATE:
                import warnings
                warnings.warn("Rs.close: a pipe still borrows this handle (sfe_dsp_pipe_destroy first); handle kept",
                              ResourceWarning, stacklevel=2)
                return
            self._h = None

    __del__ = close


def rs_plan(state, upsample, n_in, out_len, rate):
    """Host-only replay of one process() call's time law (sfe_dsp_rs_plan).
    state: lib.TimeState (updated in place).  Returns (rel_pos int32[], mu float32[])."""
    L = _l.load()
    cap = out_len + 1
    pos = np.zeros(cap, dtype=np.int32)
    mu = np.zeros(cap, dtype=np.float32)
    n = C.c_int(0)
    check(L.sfe_dsp_rs_plan(C.byref(state), upsample, n_in, out_len, rate, pos.ctypes.data, mu.ctypes.data, cap,
                            C.byref(n)))
    return pos[: n.value], mu[: n.value]


# ------------------------------------------------------------------ reference projection
class blkconv:
    """libdsp/blkconv.h:35-62 -- blkconv(taps, fft_len); get_blksize(); get_process_buf();
    process().  The buffer is pinned host memory owned by the object."""

    def __init__(self, taps, fft_len):
        self._fir = Fir(_f32(taps), data_complex=False, n_channels=1, block_hint=int(fft_len))
        self._buf, self._blk = self._fir.host_buffer()

    def get_blksize(self):
        return self._blk

    def get_process_buf(self):
        return self._buf

    def process(self):
        self._fir.process_block()


class _rs_class:
    _mode = None

    def __init__(self, taps, upsample, blksize):
        self._rs = Rs(taps, int(upsample), int(blksize), mode=self._mode)

    def process(self, x, out_len, rate):
        return self._rs.process(x, int(out_len), float(rate))


class resample(_rs_class):
    """libdsp/resample.h:33-61 via pydsp.i: process(in, out_len, rate) -> (n_out, out)."""
    _mode = _l.RS_RESAMPLE


class decimate(_rs_class):
    """libdsp/decimate.h:33-63 via pydsp.i."""
    _mode = _l.RS_DECIMATE
