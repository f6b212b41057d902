// blkconv.h -- drop-in for libdsp's `blkconv` on MI355X.
//
// Same class name, constructor and members as the reference (libdsp/blkconv.h:35-62), so a
// caller such as examples/bpsk/bpsk.cxx:125-164 or libdsp/test/test_blkconv.cxx compiles
// unchanged against this header and links libsfe_dsp.so instead of Libdsp + fftw3f.
// Everything forwards to the C ABI (sfe_dsp.h); the filtering runs on the GPU.
//
//   get_process_buf()  pinned host memory owned by the object, stable for its lifetime
//                      (bpsk.cxx:127 caches the pointer once); caller writes/reads [0, blksize)
//   process()          in place on that buffer: H2D, FIR kernel, D2H, synchronous; the
//                      overlap state is carried between calls (blkconv.cxx:105-109)
// Errors: the reference checks nothing (void, unchecked malloc).  Here a failed create or
// process prints the library's message to stderr and aborts -- there is no CPU fallback.
#ifndef SFE_DROPIN_BLKCONV_H_
#define SFE_DROPIN_BLKCONV_H_

#include <stdio.h>
#include <stdlib.h>

#include "sfe_dsp.h"

class blkconv
{
public:
    blkconv(float *taps, int n_taps, int fft_len) : m_h(0), m_buf(0), m_blk(0)
    {
        guard(sfe_dsp_fir_create(taps, n_taps, /*taps_complex*/ 0, /*data_complex*/ 0,
                                 /*n_channels*/ 1, /*block_hint*/ fft_len, device(), &m_h),
              "blkconv::blkconv");
        guard(sfe_dsp_fir_host_buffer(m_h, &m_buf, &m_blk), "blkconv::get_process_buf");
    }
    ~blkconv() { sfe_dsp_fir_destroy(m_h); }

    int get_blksize() { return m_blk; }
    float *get_process_buf() { return m_buf; }
    void process() { guard(sfe_dsp_fir_process_block(m_h), "blkconv::process"); }

private:
    blkconv(const blkconv &);             // the reference's implicit copy would double-free
    blkconv &operator=(const blkconv &);

    // the reference constructor has no device argument: the object lives on the calling thread's current
    // device (sfe_dsp_set_device / hipSetDevice; 0 unless chosen).  No environment is read.
    static int device()
    {
        int d = 0;
        (void)sfe_dsp_get_device(&d);       // no GPU: d stays 0 and the create call below reports SFE_ENODEV
        return d;
    }
    static void guard(int rc, const char *where)
    {
        if (rc != SFE_OK) {
            fprintf(stderr, "%s: %s (code %d)\n", where, sfe_dsp_last_error(), rc);
            abort();
        }
    }

    sfe_fir_t m_h;
    float    *m_buf;
    int       m_blk;
};

#endif
