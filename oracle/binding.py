"""ctypes bindings for the CPU oracle.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; nothing under simplefe_amd/ does (tests/test_layout.py checks that).

Two libraries:
  liboracle.so       -- oracle/sfe_oracle.c, the C restatement (classes Blkconv,
                        Resample, Decimate below)
  _ref/libsferef.so  -- the unmodified reference resample/decimate classes compiled from
                        /root/reference in the authoring container (RefResample,
                        RefDecimate); present on the GPU box only as the prebuilt .so
  _ref/libsferef_blkconv.so -- the unmodified reference blkconv class on ROCm's libhipfftw
                        (RefBlkconv); runs on a GPU box only
  _ref/libsferef_blkconv_fftw.so -- the unmodified reference blkconv class on the reference's
                        OWN FFTW 3.3.5 binary (contrib/fftw-3.3.5-dll64/libfftw3f-3.dll, mapped
                        in process by oracle/pe/; RefBlkconvFFTW); CPU, authoring container only
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


def build(ref=True):
    """(Re)build liboracle.so and, when /root/reference is present, _ref/libsferef.so."""
    subprocess.check_call(["make", "-s", "-C", HERE, "all" if ref else os.path.join(HERE, "liboracle.so")])


def _load(path):
    if not os.path.exists(path):
        return None
    return C.CDLL(path)


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        p = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(p):
            build(ref=os.path.isdir("/root/reference"))
        L = C.CDLL(p)
        L.orc_blkconv_create.restype = C.c_void_p
        L.orc_blkconv_create.argtypes = [_f32p, C.c_int, C.c_int]
        L.orc_blkconv_blksize.argtypes = [C.c_void_p]
        L.orc_blkconv_buf.restype = C.POINTER(C.c_float)
        L.orc_blkconv_buf.argtypes = [C.c_void_p]
        L.orc_blkconv_process.argtypes = [C.c_void_p]
        L.orc_blkconv_destroy.argtypes = [C.c_void_p]
        L.orc_blkconv_stream.argtypes = [C.c_void_p, _f32p, _f32p, C.c_long]
        for k in ("resample", "decimate"):
            getattr(L, f"orc_{k}_create").restype = C.c_void_p
            getattr(L, f"orc_{k}_create").argtypes = [_f32p, C.c_int, C.c_int, C.c_int]
            getattr(L, f"orc_{k}_process").argtypes = [C.c_void_p, _f32p, C.c_int, _f32p, C.c_int, C.c_float]
            getattr(L, f"orc_{k}_destroy").argtypes = [C.c_void_p]
        L.orc_resample_skip_calls.restype = C.c_long
        L.orc_resample_skip_calls.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_float]
        L.orc_resample_get_time.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_int)]
        L.orc_resample_set_time.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_int]
        L.orc_blkconv_stream_mt.restype = C.c_long
        L.orc_blkconv_stream_mt.argtypes = [_f32p, C.c_int, C.c_int, _f32p, C.c_void_p, C.c_long, C.c_int]
        L.orc_rs_stream_mt.restype = C.c_long
        L.orc_rs_stream_mt.argtypes = [C.c_int, _f32p, C.c_int, C.c_int, C.c_int, C.c_float, _f32p, C.c_long,
                                       C.c_long, C.c_int]
        L.orc_rx_u8_to_cf32.argtypes = [_f32p, _u8p, C.c_int]
        L.orc_rx_u8_to_f32.argtypes = [_f32p, _u8p, C.c_int]
        L.orc_tx_f32_to_10bit.argtypes = [_u8p, _f32p, C.c_int]
        _lib = L
    return _lib


def ref_lib():
    """The compiled reference, or None when oracle/_ref/libsferef.so is absent."""
    global _ref
    if _ref is None:
        p = os.path.join(HERE, "_ref", "libsferef.so")
        if not os.path.exists(p) and os.path.isdir("/root/reference"):
            build(ref=True)
        R = _load(p)
        if R is None:
            return None
        for k in ("resample", "decimate"):
            getattr(R, f"ref_{k}_create").restype = C.c_void_p
            getattr(R, f"ref_{k}_create").argtypes = [_f32p, C.c_int, C.c_int, C.c_int]
            getattr(R, f"ref_{k}_process").argtypes = [C.c_void_p, _f32p, C.c_int, _f32p, C.c_int, C.c_float]
            getattr(R, f"ref_{k}_destroy").argtypes = [C.c_void_p]
        if hasattr(R, "ref_rs_stream_mt"):
            R.ref_rs_stream_mt.restype = C.c_long
            R.ref_rs_stream_mt.argtypes = [C.c_int, _f32p, C.c_int, C.c_int, C.c_int, C.c_float, _f32p, C.c_long,
                                           C.c_long, C.c_int]
        _ref = R
    return _ref


_ref_blk = None


def ref_blkconv_lib():
    """The compiled reference blkconv class (oracle/_ref/libsferef_blkconv.so: blkconv.cxx on
    ROCm's libhipfftw), or None when it is absent.  Creating an object needs a GPU."""
    global _ref_blk
    if _ref_blk is None:
        p = os.path.join(HERE, "_ref", "libsferef_blkconv.so")
        if not os.path.exists(p) and os.path.isdir("/root/reference"):
            build(ref=True)
        try:
            R = _load(p)
        except OSError:
            R = None
        if R is None:
            return None
        R.ref_blkconv_create.restype = C.c_void_p
        R.ref_blkconv_create.argtypes = [_f32p, C.c_int, C.c_int]
        R.ref_blkconv_blksize.argtypes = [C.c_void_p]
        R.ref_blkconv_buf.restype = C.POINTER(C.c_float)
        R.ref_blkconv_buf.argtypes = [C.c_void_p]
        R.ref_blkconv_process.argtypes = [C.c_void_p]
        R.ref_blkconv_destroy.argtypes = [C.c_void_p]
        _ref_blk = R
    return _ref_blk


FFTW_DLL = os.environ.get("SFE_FFTW_DLL") or "/root/reference/contrib/fftw-3.3.5-dll64/libfftw3f-3.dll"
FFTW_DLL_SHA256 = "42ca18fff35dd12890e04478bc990005b3969cb744f6843976bd436ccd7f0a4c"      # pinned in oracle/pe/peload.c too
FFTW_OPT_IN = "SFE_ORACLE_RUN_FFTW_DLL"


def ref_blkconv_fftw_available():
    """The reference's blkconv class on the reference's own FFTW 3.3.5 binary can run here: the DLL exists, is byte for byte the
    pinned file, and the caller opted in (SFE_ORACLE_RUN_FFTW_DLL=1).  ADVICE r4: an opaque binary from the reference tree is
    never mapped into THIS process -- oracle/ref_fftw_child.py does that in a child of its own (resource limits, scratch
    directory, bare environment), and only on request."""
    if os.environ.get(FFTW_OPT_IN) != "1" or not os.path.exists(FFTW_DLL):
        return False
    import hashlib
    with open(FFTW_DLL, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest() == FFTW_DLL_SHA256


def _fftw_child(op, taps=None, fft_len=0, x=None):
    """One request to the child process: arrays travel as .npy files in a scratch directory."""
    import sys
    import tempfile
    if not ref_blkconv_fftw_available():
        raise RuntimeError("the FFTW-pinned reference is opt-in: set %s=1 (and /root/reference must hold the pinned DLL)" % FFTW_OPT_IN)
    so = os.path.join(HERE, "_ref", "libsferef_blkconv_fftw.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref_fftw"])
    with tempfile.TemporaryDirectory(prefix="sfe_fftw_") as d:
        if taps is not None:
            np.save(os.path.join(d, "taps.npy"), _f32(taps))
            np.save(os.path.join(d, "x.npy"), _f32(x))
        env = {"PATH": "/usr/bin:/bin", FFTW_OPT_IN: "1", "SFE_FFTW_DLL": FFTW_DLL}
        r = subprocess.run([sys.executable, os.path.join(HERE, "ref_fftw_child.py"), so, op, str(int(fft_len)), d],
                           cwd=d, env=env, capture_output=True, text=True, timeout=300)
        if r.returncode != 0:
            raise RuntimeError("oracle/ref_fftw_child.py failed: " + r.stderr[-2000:])
        if op == "version":
            return r.stdout.strip()
        return np.load(os.path.join(d, "y.npy"))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Blkconv:
    """orc_blkconv_* : libdsp/blkconv.h:35-62"""

    def __init__(self, taps, fft_len):
        taps = _f32(taps)
        self._L = lib()
        self._h = self._L.orc_blkconv_create(taps, len(taps), int(fft_len))
        self.blk = self._L.orc_blkconv_blksize(self._h)
        p = self._L.orc_blkconv_buf(self._h)
        self.buf = np.ctypeslib.as_array(p, shape=(fft_len + 2,))

    def process(self):
        self._L.orc_blkconv_process(self._h)

    def stream(self, x):
        x = _f32(x)
        y = np.empty_like(x)
        self._L.orc_blkconv_stream(self._h, x, y, len(x))
        return y

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.orc_blkconv_destroy(self._h)
            self._h = None


class RefBlkconv:
    """The reference's own blkconv class (libdsp/blkconv.cxx:34-122), compiled unmodified."""

    _getlib = staticmethod(ref_blkconv_lib)
    _libname = "libsferef_blkconv.so"

    def __init__(self, taps, fft_len):
        taps = _f32(taps)
        self._L = self._getlib()
        if self._L is None:
            raise RuntimeError("oracle/_ref/%s not built or not usable here" % self._libname)
        self._h = self._L.ref_blkconv_create(taps, len(taps), int(fft_len))
        self.blk = self._L.ref_blkconv_blksize(self._h)
        p = self._L.ref_blkconv_buf(self._h)
        self.buf = np.ctypeslib.as_array(p, shape=(fft_len + 2,))

    def process(self):
        self._L.ref_blkconv_process(self._h)

    def stream(self, x):
        """Feed x block by block the way the reference's callers do (write [0, blk), process(),
        read [0, blk)); a ragged tail is zero-padded and the padding's outputs dropped."""
        x = _f32(x)
        y = np.empty_like(x)
        for off in range(0, len(x), self.blk):
            m = min(self.blk, len(x) - off)
            self.buf[:m] = x[off:off + m]
            self.buf[m:self.blk] = 0.0
            self.process()
            y[off:off + m] = self.buf[:m]
        return y

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.ref_blkconv_destroy(self._h)
            self._h = None


class RefBlkconvFFTW:
    """The reference's own blkconv class on the reference's own FFTW 3.3.5 (the vendored Win64 DLL mapped by oracle/pe/),
    run in a CHILD process (oracle/ref_fftw_child.py); opt-in, CPU, exists only where /root/reference does."""

    def __init__(self, taps, fft_len):
        self.taps, self.fft_len = _f32(taps), int(fft_len)

    def stream(self, x):
        """ref_blkconv_stream of the wrapper: the class driven block by block inside the child's C++"""
        return _fftw_child("stream", self.taps, self.fft_len, x)

    def stream_blocks(self, x):
        """the same through get_process_buf() / process() from Python, block by block (RefBlkconv.stream's loop)"""
        return _fftw_child("blocks", self.taps, self.fft_len, x)

    @classmethod
    def fftw_version(cls):
        return _fftw_child("version") if ref_blkconv_fftw_available() else None


class _Rs:
    _prefix = None
    _getlib = staticmethod(lib)

    def __init__(self, taps, upsample, blksize):
        taps = _f32(taps)
        self._L = self._getlib()
        if self._L is None:
            raise RuntimeError("oracle/_ref/libsferef.so not built")
        self._p = self._prefix
        self._h = getattr(self._L, self._p + "_create")(taps, len(taps), int(upsample), int(blksize))
        self.blksize = blksize

    def process(self, x, out_len, rate):
        x = _f32(x)
        out = np.zeros(out_len, dtype=np.float32)
        n = getattr(self._L, self._p + "_process")(self._h, x, len(x), out, int(out_len), float(rate))
        return n, out

    def stream(self, x, rate, chunk=None, out_len=None):
        """Call process() chunk by chunk (as libdsp/test/test_decimate.py:22-25 does);
        returns (concatenated outputs, list of per-call n_out)."""
        x = _f32(x)
        chunk = chunk or self.blksize
        ys, ns = [], []
        for off in range(0, len(x), chunk):
            seg = x[off:off + chunk]
            ol = out_len if out_len is not None else int(np.ceil(len(seg) / min(rate, 1e9))) + 2
            n, o = self.process(seg, ol, rate)
            ys.append(o[:n])
            ns.append(n)
        return (np.concatenate(ys) if ys else np.zeros(0, np.float32)), ns

    def __del__(self):
        if getattr(self, "_h", None):
            getattr(self._L, self._p + "_destroy")(self._h)
            self._h = None


class Resample(_Rs):
    _prefix = "orc_resample"

    def skip_calls(self, n_calls, n_in, rate):
        """The time law of n_calls process() calls of n_in samples each, without their samples (sfe_oracle.c:
        orc_resample_skip_calls): the state (pos, mu, leftover) the reference's object would hold after them.  Returns how many
        outputs those calls emit.  Run one REAL call before checking values (history, m_last_remain)."""
        k = self._L.orc_resample_skip_calls(self._h, int(n_calls), int(n_in), float(rate))
        if k < 0:
            raise ValueError("skip_calls: n_in / rate outside what process() takes")
        return int(k)

    def get_time(self):
        """(m_pos, m_mu, m_is_leftover)"""
        p, m, l = C.c_int(), C.c_float(), C.c_int()
        self._L.orc_resample_get_time(self._h, C.byref(p), C.byref(m), C.byref(l))
        return p.value, m.value, l.value

    def set_time(self, state):
        self._L.orc_resample_set_time(self._h, int(state[0]), float(state[1]), int(state[2]))


class Decimate(_Rs):
    _prefix = "orc_decimate"


class RefResample(_Rs):
    _prefix = "ref_resample"
    _getlib = staticmethod(ref_lib)


class RefDecimate(_Rs):
    _prefix = "ref_decimate"
    _getlib = staticmethod(ref_lib)


def blkconv_stream_mt(taps, fft_len, x, n_threads, want_output=True):
    """One real stream through n_threads blkconv objects, one per span with n_taps-1 samples of
    lead-in (all-host-cores baseline, bench.py).  Returns y (or None) ."""
    taps, x = _f32(taps), _f32(x)
    y = np.empty_like(x) if want_output else None
    lib().orc_blkconv_stream_mt(taps, len(taps), int(fft_len), x, y.ctypes.data if want_output else None,
                                len(x), int(n_threads))
    return y


def rs_stream_mt(which, taps, upsample, blksize, rate, x, quantum, n_threads, reference=False):
    """One real stream through n_threads resample/decimate objects (spans cut on whole phase
    periods, phase_len+1 samples of lead-in); returns the number of outputs produced (lead-ins
    included).  reference=True runs the reference's own classes (oracle/_ref)."""
    taps, x = _f32(taps), _f32(x)
    fn = ref_lib().ref_rs_stream_mt if reference else lib().orc_rs_stream_mt
    return fn(1 if which == "decimate" else 0, taps, len(taps), int(upsample), int(blksize), float(rate), x, len(x),
              int(quantum), int(n_threads))


def rx_u8_to_cf32(b):
    b = np.ascontiguousarray(b, dtype=np.uint8)
    out = np.empty(len(b), dtype=np.float32)
    lib().orc_rx_u8_to_cf32(out, b, len(b))
    return out


def rx_u8_to_f32(b):
    b = np.ascontiguousarray(b, dtype=np.uint8)
    out = np.empty(len(b), dtype=np.float32)
    lib().orc_rx_u8_to_f32(out, b, len(b))
    return out


def tx_f32_to_10bit(x):
    x = _f32(x)
    out = np.zeros(len(x) // 4 * 5, dtype=np.uint8)
    n = lib().orc_tx_f32_to_10bit(out, x, len(x))
    return out[:n]
