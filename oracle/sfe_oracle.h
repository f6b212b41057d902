/*
 * sfe_oracle.h -- CPU oracle for the libdsp sample-stream hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the algorithms in
 * wnmusic/simpleFE's libdsp (blkconv / resample / decimate).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and only as
 * the checker / the reported CPU baseline -- never as the thing shipped or measured
 * as the product.  The product path (libsfe_dsp.so) does not link or call it.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - orc_resample_* / orc_decimate_* : pinned BIT-EXACT against the unmodified
 *     reference sources compiled in place into oracle/_ref/ (tests/golden fixtures).
 *   - orc_blkconv_* : pinned at the FFTW boundary against OUTPUTS OF THE REFERENCE ITSELF RUN HERE
 *     (round 4): the reference's FFT arithmetic lives in FFTW 3.3.5 (libfftw3f), vendored only as
 *     a Win64 binary (contrib/fftw-3.3.5-dll64/libfftw3f-3.dll).  oracle/pe/ maps that x86-64
 *     binary into the process (operating-system stubs only, no arithmetic) and the UNMODIFIED
 *     blkconv.cxx calls it; tests/golden/g7_blkconv_fftw.npz holds the results (five shapes,
 *     block by block).  The restatement carries an own float32 FFT, so it matches them to
 *     transform rounding: rel-RMS 3e-8 .. 3.0e-7 against the 1e-5 bar.  Cross-checks kept from
 *     earlier rounds: the same class on ROCm's libhipfftw (g6, GPU box only), the reference's
 *     known-answer scenario (libdsp/test/test_blkconv.cxx:5-33), float64 direct convolution.
 */
#ifndef SFE_ORACLE_H_
#define SFE_ORACLE_H_

#ifdef __cplusplus
extern "C" {
#endif

/* ---- blkconv : libdsp/blkconv.h:35-62, libdsp/blkconv.cxx:34-122 -------------- */
typedef struct orc_blkconv orc_blkconv;
orc_blkconv *orc_blkconv_create(const float *taps, int n_taps, int fft_len);
int          orc_blkconv_blksize(const orc_blkconv *c);      /* blkconv.h:40-43 */
float       *orc_blkconv_buf(orc_blkconv *c);                /* blkconv.h:44-47 */
void         orc_blkconv_process(orc_blkconv *c);            /* blkconv.cxx:77-110 */
void         orc_blkconv_destroy(orc_blkconv *c);
/* convenience: run a whole real stream through process() block by block; the last
 * partial block is zero-filled as a caller of the reference would.  y gets n floats. */
void         orc_blkconv_stream(orc_blkconv *c, const float *x, float *y, long n);

/* ---- resample : libdsp/resample.h:33-61, libdsp/resample.cxx:37-153 ----------- */
typedef struct orc_resample orc_resample;
orc_resample *orc_resample_create(const float *taps, int n_taps, int upsample, int blksize);
int           orc_resample_process(orc_resample *r, const float *in, int n_in,
                                   float *out, int out_len, float rate);
/* test helper: the time law of n_calls calls of n_in samples without their samples; returns the outputs they emit */
long          orc_resample_skip_calls(orc_resample *r, long n_calls, int n_in, float rate);
void          orc_resample_get_time(const orc_resample *r, int *pos, float *mu, int *leftover);
void          orc_resample_set_time(orc_resample *r, int pos, float mu, int leftover);
void          orc_resample_destroy(orc_resample *r);

/* ---- decimate : libdsp/decimate.h:33-63, libdsp/decimate.cxx:37-140 ----------- */
typedef struct orc_decimate orc_decimate;
orc_decimate *orc_decimate_create(const float *taps, int n_taps, int upsample, int blksize);
int           orc_decimate_process(orc_decimate *d, const float *in, int n_in,
                                   float *out, int out_len, float rate);
void          orc_decimate_destroy(orc_decimate *d);

/* ---- wire-format converters (gr-simplefe; "next" row N2) ---------------------- */
/* u8 offset-binary I/Q -> cf32, gr-simplefe/lib/source_c_impl.cc:121-132 */
int orc_rx_u8_to_cf32(float *dst, const unsigned char *src, int src_len);
/* u8 -> f32 single channel, gr-simplefe/lib/source_f_impl.cc:120-129 */
int orc_rx_u8_to_f32(float *dst, const unsigned char *src, int src_len);
/* f32 -> 10-bit offset binary, 4 samples in 5 bytes,
 * gr-simplefe/lib/sink_c_impl.cc:118-144 == examples/bpsk/bpsk.cxx:76-101 */
int orc_tx_f32_to_10bit(unsigned char *dst, const float *src, int src_len);

/* ---- all-host-cores baseline helpers (bench.py cpu_baseline.all_cores) ----------
 * One stream cut into n_threads spans, one object per span, each fed n_taps-1 (resamplers:
 * phase_len+1) samples of lead-in so its carried state is the stream's; the lead-in's outputs
 * are dropped.  blkconv: y (may be NULL) receives the n outputs.  Return: outputs produced. */
long orc_blkconv_stream_mt(const float *taps, int n_taps, int fft_len, const float *x, float *y,
                           long n, int n_threads);
/* quantum: cut points are multiples of it (input samples per phase period, step/gcd(step,U)) */
long orc_rs_stream_mt(int decimate_class, const float *taps, int n_taps, int upsample, int blksize,
                      float rate, const float *x, long n, long quantum, int n_threads);

#ifdef __cplusplus
}
#endif
#endif
