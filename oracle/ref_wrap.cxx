// ref_wrap.cxx -- extern "C" handles around the UNMODIFIED reference classes.
//
// TEST INFRASTRUCTURE ONLY.  Compiled by oracle/Makefile together with
// /root/reference/libdsp/resample.cxx and decimate.cxx, from where they lie, into
// oracle/_ref/libsferef.so (git-ignored; never copied into the repo).  Used to pin
// the C restatement (sfe_oracle.c), to generate tests/golden/*.npz, and as the
// "reference" CPU baseline in bench.py.
//
// blkconv.cxx is built separately (ref_wrap_blkconv.cxx): it needs an FFTW3-API library, and
// the one this image has (ROCm's libhipfftw) runs on a GPU box only.
#include "decimate.h"   // -I/root/reference/libdsp
#include "resample.h"

extern "C" {

void *ref_resample_create(float *taps, int n_taps, int upsample, int blksize)
{
    return new resample(taps, n_taps, upsample, blksize);
}
int ref_resample_process(void *h, float *in, int n_in, float *out, int out_len, float rate)
{
    return static_cast<resample *>(h)->process(in, n_in, out, out_len, rate);
}
void ref_resample_destroy(void *h) { delete static_cast<resample *>(h); }

void *ref_decimate_create(float *taps, int n_taps, int upsample, int blksize)
{
    return new decimate(taps, n_taps, upsample, blksize);
}
int ref_decimate_process(void *h, float *in, int n_in, float *out, int out_len, float rate)
{
    return static_cast<decimate *>(h)->process(in, n_in, out, out_len, rate);
}
void ref_decimate_destroy(void *h) { delete static_cast<decimate *>(h); }

}  // extern "C"
