#!/usr/bin/env python3
"""Child process of oracle/binding.py: RefBlkconvFFTW -- the ONLY process that maps the reference's vendored FFTW binary.

    python oracle/ref_fftw_child.py <libsferef_blkconv_fftw.so> <stream|blocks|version> <fft_len> <scratch dir>

Test infrastructure (the parity checker), never the product.  The library it loads maps libfftw3f-3.dll (oracle/pe/) only if
SFE_ORACLE_RUN_FFTW_DLL=1 and the file's SHA-256 is the pinned one.  Before anything of it runs this process gives up what it does
not need: CPU time, file size and descriptor limits, no new privileges; it works in the scratch directory it was started in and
talks to its parent through .npy files there.
"""
import ctypes as C
import os
import resource
import sys

import numpy as np


def main():
    so, op, fft_len, d = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    resource.setrlimit(resource.RLIMIT_CPU, (120, 120))
    resource.setrlimit(resource.RLIMIT_FSIZE, (1 << 28, 1 << 28))
    resource.setrlimit(resource.RLIMIT_NOFILE, (64, 64))
    resource.setrlimit(resource.RLIMIT_NPROC, (resource.getrlimit(resource.RLIMIT_NPROC)[0],) * 2)
    try:
        C.CDLL(None).prctl(38, 1, 0, 0, 0)                  # PR_SET_NO_NEW_PRIVS
    except Exception:                                       # noqa: BLE001
        pass
    f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
    R = C.CDLL(so)
    R.sfe_pe_fftwf_version.restype = C.c_char_p
    if op == "version":
        print(R.sfe_pe_fftwf_version().decode())
        return 0
    R.ref_blkconv_create.restype = C.c_void_p
    R.ref_blkconv_create.argtypes = [f32p, C.c_int, C.c_int]
    R.ref_blkconv_blksize.argtypes = [C.c_void_p]
    R.ref_blkconv_buf.restype = C.POINTER(C.c_float)
    R.ref_blkconv_buf.argtypes = [C.c_void_p]
    R.ref_blkconv_process.argtypes = [C.c_void_p]
    R.ref_blkconv_destroy.argtypes = [C.c_void_p]
    R.ref_blkconv_stream.argtypes = [C.c_void_p, f32p, f32p, C.c_long]
    taps = np.ascontiguousarray(np.load(os.path.join(d, "taps.npy")), dtype=np.float32)
    x = np.ascontiguousarray(np.load(os.path.join(d, "x.npy")), dtype=np.float32)
    h = R.ref_blkconv_create(taps, len(taps), fft_len)
    y = np.empty_like(x)
    if op == "stream":
        R.ref_blkconv_stream(h, x, y, len(x))
    else:
        blk = R.ref_blkconv_blksize(h)
        buf = np.ctypeslib.as_array(R.ref_blkconv_buf(h), shape=(fft_len + 2,))
        for off in range(0, len(x), blk):                   # the way the reference's callers drive the class
            m = min(blk, len(x) - off)
            buf[:m] = x[off:off + m]
            buf[m:blk] = 0.0
            R.ref_blkconv_process(h)
            y[off:off + m] = buf[:m]
    R.ref_blkconv_destroy(h)
    np.save(os.path.join(d, "y.npy"), y)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
