// decimate.h -- drop-in for libdsp's `decimate` on MI355X (libdsp/decimate.h:33-63).
//
// decimate(taps, n_taps, upsample, blksize); process(in, n_in, out, out_len, rate) with the
// reference's checks and messages (libdsp/decimate.cxx:75-87: rate >= 1, n_in <= blksize,
// out_len) and state (decimate.cxx:96-127).  Output is bit-identical to `resample` for the
// same arguments, as in the reference (libdsp/test/test_decimate.py:36).
#ifndef SFE_DROPIN_DECIMATE_H_
#define SFE_DROPIN_DECIMATE_H_

#include <stdio.h>
#include <stdlib.h>

#include "sfe_dsp.h"

class decimate
{
public:
    decimate(float *taps, int n_taps, int upsample, int blksize) : m_h(0)
    {
        int dev = 0;                        // the calling thread's current device (sfe_dsp_set_device); no environment is read
        (void)sfe_dsp_get_device(&dev);
        int rc = sfe_dsp_rs_create(taps, n_taps, upsample, blksize, /*data_complex*/ 0,
                                   /*n_channels*/ 1, dev, SFE_RS_DECIMATE, &m_h);
        if (rc != SFE_OK) {
            fprintf(stderr, "decimate::decimate: %s (code %d)\n", sfe_dsp_last_error(), rc);
            abort();
        }
    }
    ~decimate() { sfe_dsp_rs_destroy(m_h); }

    int process(float *in, int n_in, float *out, int out_len, float rate)
    {
        int n_out = 0;
        int rc = sfe_dsp_rs_process(m_h, in, n_in, out, out_len, rate, &n_out);
        if (rc != SFE_OK) {
            fprintf(stderr, "decimate::process: %s (code %d)\n", sfe_dsp_last_error(), rc);
            abort();
        }
        return n_out;
    }

private:
    decimate(const decimate &);
    decimate &operator=(const decimate &);
    sfe_rs_t m_h;
};

#endif
