// resample.h -- drop-in for libdsp's `resample` on MI355X (libdsp/resample.h:33-61).
//
// resample(taps, n_taps, upsample, blksize); process(in, n_in, out, out_len, rate) returns
// the number of outputs, with the reference's parameter checks, stdout messages and
// "return 0" behaviour (libdsp/resample.cxx:91-98) and its leftover / float32 time-recurrence
// state (resample.cxx:119-150).  Host pointers in and out; the polyphase arithmetic runs on
// the GPU in the reference's operation order (bit-exact with the CPU class).
#ifndef SFE_DROPIN_RESAMPLE_H_
#define SFE_DROPIN_RESAMPLE_H_

#include <stdio.h>
#include <stdlib.h>

#include "sfe_dsp.h"

class resample
{
public:
    resample(float *taps, int n_taps, int upsample, int blksize) : m_h(0)
    {
        int dev = 0;                        // the calling thread's current device (sfe_dsp_set_device); no environment is read
        (void)sfe_dsp_get_device(&dev);
        int rc = sfe_dsp_rs_create(taps, n_taps, upsample, blksize, /*data_complex*/ 0,
                                   /*n_channels*/ 1, dev, SFE_RS_RESAMPLE, &m_h);
        if (rc != SFE_OK) {
            fprintf(stderr, "resample::resample: %s (code %d)\n", sfe_dsp_last_error(), rc);
            abort();
        }
    }
    ~resample() { sfe_dsp_rs_destroy(m_h); }

    int process(float *in, int n_in, float *out, int out_len, float rate)
    {
        int n_out = 0;
        int rc = sfe_dsp_rs_process(m_h, in, n_in, out, out_len, rate, &n_out);
        if (rc != SFE_OK) {
            fprintf(stderr, "resample::process: %s (code %d)\n", sfe_dsp_last_error(), rc);
            abort();
        }
        return n_out;
    }

private:
    resample(const resample &);
    resample &operator=(const resample &);
    sfe_rs_t m_h;
};

#endif
