// ref_wrap_blkconv.cxx -- extern "C" handles around the UNMODIFIED reference blkconv class.
//
// TEST INFRASTRUCTURE ONLY.  oracle/Makefile compiles this together with
// /root/reference/libdsp/blkconv.cxx, from where it lies, into oracle/_ref/libsferef_blkconv.so.
// blkconv.cxx needs the FFTW3 single-precision API.  The image has no libfftw3f (the reference
// ships it only as Win64 DLLs), but it does have ROCm's libhipfftw.so -- hipFFT's implementation
// of that API (fftwf_plan_dft_r2c_1d, fftwf_execute, ... on host pointers) -- so the class is
// compiled against the reference's own vendored header (contrib/fftw-3.3.5-dll64/fftw3.h, used
// in place) and linked to libhipfftw.  Nothing is stubbed.  Because hipFFT executes on the GPU
// this library only RUNS on a GPU box: it pins the restatement in `-m gpu` tests and generated
// tests/golden/g6_blkconv_reference.npz (tests/golden/make_golden_blkconv.py, run there).
//
// Round 4: the same wrapper is ALSO linked against oracle/pe/ (an in-process mapper for the
// reference's own vendored FFTW 3.3.5 Win64 binary) into oracle/_ref/libsferef_blkconv_fftw.so,
// which runs on the CPU of the authoring container and is what PINS blkconv at the FFTW boundary
// (tests/golden/make_golden_fftw.py -> g7_blkconv_fftw.npz).
#include "blkconv.h"   // -I/root/reference/libdsp

#include <string.h>

extern "C" {

void *ref_blkconv_create(float *taps, int n_taps, int fft_len) { return new blkconv(taps, n_taps, fft_len); }
int ref_blkconv_blksize(void *h) { return static_cast<blkconv *>(h)->get_blksize(); }
float *ref_blkconv_buf(void *h) { return static_cast<blkconv *>(h)->get_process_buf(); }
void ref_blkconv_process(void *h) { static_cast<blkconv *>(h)->process(); }
void ref_blkconv_destroy(void *h) { delete static_cast<blkconv *>(h); }

// n samples through the object the way the reference's callers drive it (examples/bpsk/bpsk.cxx:
// 145-164: write [0, blk), process(), read [0, blk)); a ragged tail is zero-filled.  For timing
// the class without a Python loop around it.
void ref_blkconv_stream(void *h, const float *x, float *y, long n)
{
    blkconv *b = static_cast<blkconv *>(h);
    const long blk = b->get_blksize();
    float *buf = b->get_process_buf();
    for (long off = 0; off < n; off += blk) {
        const long m = n - off < blk ? n - off : blk;
        memcpy(buf, x + off, m * sizeof(float));
        if (m < blk) memset(buf + m, 0, (blk - m) * sizeof(float));
        b->process();
        memcpy(y + off, buf, m * sizeof(float));
    }
}

}  // extern "C"
