/*
 * sfe_dsp.h -- C ABI of libsfe_dsp.so, the MI355X (gfx950) implementation of simpleFE's
 * libdsp sample-stream hot path: blkconv (FIR block convolution), resample and decimate
 * (polyphase interpolating resamplers).
 *
 * This is the drop-in boundary.  The reference has no FFI/plugin registry; what it exposes
 * for this path is the C++ class surface of its static library `Libdsp`
 * (libdsp/CMakeLists.txt:16-21; classes at libdsp/blkconv.h:35-62, libdsp/resample.h:33-61,
 * libdsp/decimate.h:33-63) and its SWIG projection (libdsp/test/pydsp.i:16-22).  Every entry
 * point below names the reference member it stands behind; include/blkconv.h, resample.h,
 * decimate.h are header-only classes with the reference's names and signatures that forward
 * here, so existing callers (examples/bpsk/bpsk.cxx:125-164, libdsp/test/test_blkconv.cxx)
 * compile unchanged.  INTEGRATION.md shows the binding a maintainer would add.
 *
 * Conventions
 *   - plain C types only: pointers, sizes, ints.  No HIP, torch or C++ types.
 *   - every function returns an int status: 0 (SFE_OK) or a negative SFE_E* code;
 *     sfe_dsp_last_error() gives the message of the calling thread's last failure.
 *   - one handle == one stream of samples with its own filter state, exactly like one
 *     reference object.  Handles are not thread-safe; callers serialise per handle, the rule
 *     the reference has (examples/bpsk/bpsk.cxx:132-170).
 *   - sfe_stream_t is a hipStream_t passed as void* (NULL = the default stream).  The
 *     *_stream entry points are asynchronous on that stream; the host-pointer entry points
 *     (class-compatible) return when the result is in host memory.
 *   - complex samples are interleaved (re, im) float32 pairs -- gr_complex, as gr-simplefe
 *     uses (gr-simplefe/lib/source_c_impl.cc:46,123-128).  libdsp itself is real-only
 *     (libdsp/blkconv.h:38-48); complex data is the composition SURVEY.md 8(a) row A0 defines.
 *   - there is NO CPU fallback: without a usable GPU the create calls fail with SFE_ENODEV.
 */
#ifndef SFE_DSP_H_
#define SFE_DSP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SFE_OK       0
#define SFE_EINVAL  (-1)  /* bad argument */
#define SFE_ENOMEM  (-2)  /* host or device allocation failed */
#define SFE_EHIP    (-3)  /* a HIP runtime call failed (message has the HIP error string) */
#define SFE_ENODEV  (-4)  /* no usable gfx950 device */
#define SFE_ESTATE  (-5)  /* call not valid in the handle's current state */
#define SFE_ERANGE  (-6)  /* output buffer too small */

typedef void *sfe_stream_t;   /* hipStream_t */
typedef void *sfe_fir_t;      /* opaque: one FIR stream (blkconv) */
typedef void *sfe_pipe_t;     /* opaque: pipelined host streaming over one FIR handle */
typedef void *sfe_rs_t;       /* opaque: one resample/decimate stream */
typedef void *sfe_timer_t;    /* opaque: a pair of HIP events */

/* ------------------------------------------------------------------ runtime / plumbing */
const char *sfe_dsp_version(void);
const char *sfe_dsp_last_error(void);
int sfe_dsp_device_count(int *count);
/* The calling thread's current device (HIP's notion: hipSetDevice / hipGetDevice).  Handles name
 * their device at create and switch to it on every call, so these two matter only to what has no
 * device argument: sfe_dsp_malloc and friends, and the drop-in classes blkconv / resample /
 * decimate (the class headers under include/), whose reference constructors have no such argument -- an object lives on
 * the device that is current when it is constructed (0 unless the caller chose another).
 * get_device: SFE_ENODEV without a GPU. */
int sfe_dsp_set_device(int device);
int sfe_dsp_get_device(int *device);
int sfe_dsp_sync(sfe_stream_t stream);                       /* hipStreamSynchronize */
int sfe_dsp_malloc(void **dptr, size_t bytes);               /* device memory */
int sfe_dsp_free(void *dptr);
int sfe_dsp_host_alloc(void **hptr, size_t bytes);           /* pinned host memory */
int sfe_dsp_host_free(void *hptr);
int sfe_dsp_memcpy_h2d(void *dptr, const void *hptr, size_t bytes, sfe_stream_t stream);
int sfe_dsp_memcpy_d2h(void *hptr, const void *dptr, size_t bytes, sfe_stream_t stream);
int sfe_dsp_memset(void *dptr, int value, size_t bytes, sfe_stream_t stream);
/* HIP-event timer on `stream` (bench.py's roofline leg): start/stop record events there */
int sfe_dsp_timer_create(sfe_timer_t *t);
int sfe_dsp_timer_start(sfe_timer_t t, sfe_stream_t stream);
int sfe_dsp_timer_stop(sfe_timer_t t, sfe_stream_t stream);
int sfe_dsp_timer_elapsed_ms(sfe_timer_t t, float *ms);      /* synchronises on stop */
int sfe_dsp_timer_destroy(sfe_timer_t t);
/* Synthetic stream generator: float i of the run is
 * (int32(hash32(seed, channel, first + i)) >> 8) * 2^-23; simplefe_amd/synth.py is the
 * host twin. */
int sfe_dsp_synth_fill(void *dptr, uint64_t n_floats, uint32_t seed, uint32_t channel,
                       uint64_t first, sfe_stream_t stream);

/* ------------------------------------------------------------------------ FIR (blkconv)
 * y[n] = sum_{k < n_taps} h[k] x[n-k], zero initial state, state carried across calls:
 * the net effect of blkconv::process() over a stream (libdsp/blkconv.cxx:77-110), for
 * real or complex data and real or complex taps. */
#define SFE_FIR_ALGO_AUTO    0
#define SFE_FIR_ALGO_DIRECT  1   /* time-domain, LDS-staged (short filters; evidence kernel) */
#define SFE_FIR_ALGO_FFT     2   /* in-LDS 4096-point FFT overlap-save (the headline kernel) */

/* Replaces blkconv::blkconv(float *taps, int n_taps, int fft_len)   libdsp/blkconv.cxx:34-75.
 *   taps          n_taps floats, or n_taps (re,im) pairs when taps_complex; copied.
 *   data_complex  0: real float32 samples (libdsp's own case); 1: interleaved cf32.
 *   n_channels    independent streams sharing the taps, each with its own state (>= 1).
 *   block_hint    the caller's fft_len: only fixes the size of the class-compatible host
 *                 block, blk = block_hint + 1 - n_taps (blkconv.cxx:47); 0 = no host block.
 *   device        HIP device ordinal. */
int sfe_dsp_fir_create(const float *taps, int n_taps, int taps_complex, int data_complex,
                       int n_channels, int block_hint, int device, sfe_fir_t *out);

/* The same with a tap vector PER CHANNEL (what 64 reference objects with 64 different filters are):
 *   taps  [n_channels][n_taps] floats ([n_channels][n_taps] (re,im) pairs when taps_complex); copied.
 * Complex float32 streams, the FFT path (any n_taps it takes); one launch covers all channels: a
 * workgroup holds its channel's spectrum in registers as for a shared filter and reloads it when
 * the next transform it draws belongs to another channel (+3 % at 64 x 2^24).  No host block. */
int sfe_dsp_fir_create_per_channel(const float *taps, int n_taps, int taps_complex, int n_channels,
                                   int device, sfe_fir_t *out);
/* Host-only (no GPU): how a tap count is served by the 4096-point kernel -- the overlap of one
 * transform (a multiple of 256), the number of tap partitions (one launch each; 1 for any filter
 * a single transform overlaps economically, i.e. up to ~2800 taps) and the samples a transform
 * advances.  The reference's analogue is the caller's choice of fft_len (blkconv.cxx:47-48:
 * blk = fft_len + 1 - n_taps); here it is chosen by cost.  SFE_ERANGE beyond 1024 partitions. */
int sfe_dsp_fir_plan(int n_taps, int *overlap, int *partitions, int *advance);
/* Replaces get_process_buf()/get_blksize()  libdsp/blkconv.h:40-47: a pinned host buffer
 * owned by the handle, stable for its lifetime; the caller writes and reads [0, blk). */
int sfe_dsp_fir_host_buffer(sfe_fir_t h, float **buf, int *blk);
/* Replaces blkconv::process()  libdsp/blkconv.cxx:77-110: filters the blk samples in the
 * host buffer in place (H2D, kernel, D2H, synchronous); overlap state is carried. */
int sfe_dsp_fir_process_block(sfe_fir_t h);
/* Bulk device-resident form of the same law (the measured path): n samples per channel,
 * channel c at d_in + c*in_stride and d_out + c*out_stride (strides in samples; pass n for
 * packed; with n_channels > 1 both must be >= n).  The input and output byte ranges must not
 * overlap at all (SFE_EINVAL).  Asynchronous on `stream`.  Output is complex when data or taps
 * are complex, else real.  Buffers are aligned to their element: 8 bytes for complex float32,
 * 4 for real float32; with SFE_FMT_U8 input d_in needs 2-byte alignment for (I,Q) pairs and
 * none for real streams (16-byte aligned streams take the faster wide-lane request); with
 * SFE_FMT_TX10 output d_out needs none. */
int sfe_dsp_fir_process_stream(sfe_fir_t h, const void *d_in, void *d_out, size_t n,
                               size_t in_stride, size_t out_stride, sfe_stream_t stream);
/* Host-pointer form of the bulk law for ONE channel and any n: H2D, kernel, D2H through pinned
 * staging owned by the handle, synchronous, state carried.  This is what a GNU Radio
 * work(noutput_items, in, out) adapter calls (include/gr_sfe/): gr-simplefe's blocks hand
 * host buffers of scheduler-chosen length (gr-simplefe/lib/source_c_impl.cc:134-153). */
int sfe_dsp_fir_process_host(sfe_fir_t h, const void *in, void *out, size_t n);
/* One stream cut into spans: before a span's first call, hand the handle the
 * samples that precede the span -- d_prev[0 .. n_prev), float32 in the handle's element type, per
 * channel at `stride` samples; n_prev >= n_taps-1 reproduces the uncut stream exactly, fewer are
 * taken as preceded by zeros.  This is the reference's whole carried state: m_overlap is a
 * function of exactly those n_taps-1 inputs (libdsp/blkconv.cxx:105-109).  With cuts on multiples
 * of the transform advance (3840 for <= 257 taps) and n_prev >= the transform overlap (n_taps-1
 * rounded up to a multiple of 256) the spans' outputs are bit-identical to the one-handle
 * result; otherwise equal to float32 rounding.  Asynchronous on `stream`. */
int sfe_dsp_fir_load_history(sfe_fir_t h, const void *d_prev, size_t n_prev, size_t stride,
                             sfe_stream_t stream);
int sfe_dsp_fir_set_algo(sfe_fir_t h, int algo);
/* How the rows of an aligned complex float32 stream reach the FFT kernel's transform: guarded
 * register loads, LDS-DMA requested early, or LDS-DMA into a wave-private exchange layout
 * (DESIGN.md 4.1).  Same arithmetic, bit-identical output; which is fastest differs by a few
 * percent between devices of one pool and from run to run.  A stream call NEVER measures (round
 * 4; round 3's first large call blocked for ~100 ms to do so): it runs what set_variant fixed,
 * else what an earlier sfe_dsp_fir_calibrate chose for its (device, channels, size class,
 * overlap, per-channel taps), else register loads.
 * sfe_dsp_fir_calibrate is the measurement, made when the caller asks: it times every variant
 * over the caller's own buffers -- a call shaped like the stream calls to come: same n, strides
 * and channel count; d_out receives the filtered d_in each time -- in interleaved rounds for at
 * least 80 ms of launches, the last nine counted, SYNCHRONOUSLY on `stream`, and the process
 * remembers the choice for that shape: register loads unless another variant's median is more
 * than 1 % ahead.  It does not advance the stream: carried state and position are as before.
 * `chosen` (may be NULL) receives the variant.  Refused inside a hipGraph capture (SFE_ESTATE).
 * get_variant: what the handle's last bulk call ran, how many measurements this handle made, and
 * (ms_by_variant: 3 floats, may be NULL) the medians of its last measurement.
 * forget_calibrations drops the process-wide memory.  No reference counterpart: blkconv has one
 * code path (libdsp/blkconv.cxx:77-110). */
#define SFE_FIR_VARIANT_AUTO          (-1)
#define SFE_FIR_VARIANT_REGISTER_LOADS  0
#define SFE_FIR_VARIANT_LDS_DMA         1
#define SFE_FIR_VARIANT_WAVE_PRIVATE    2
int sfe_dsp_fir_set_variant(sfe_fir_t h, int variant);
int sfe_dsp_fir_get_variant(sfe_fir_t h, int *last_variant, int *calibrations, float *ms_by_variant);
int sfe_dsp_fir_forget_calibrations(void);
int sfe_dsp_fir_calibrate(sfe_fir_t h, const void *d_in, void *d_out, size_t n, size_t in_stride,
                          size_t out_stride, sfe_stream_t stream, int *chosen);
/* Host calls (blkconv::process() on the object's buffer, sfe_dsp_fir_process_host) of at most
 * max_samples samples let the kernel read and write pinned host memory itself -- one stream
 * operation instead of copy-in, launch, copy-out (default 2^20 samples; 0 = always the DMA copies).
 * Has no reference counterpart (the reference's buffer, libdsp/blkconv.h:44-47, is plain host
 * memory); a tuning knob of the compatibility path.  Set before the first process_host call. */
int sfe_dsp_fir_set_zero_copy_max(sfe_fir_t h, size_t max_samples);
/* Pipelined host streaming for scheduler-sized calls.  A GNU Radio scheduler
 * hands a block a few thousand items per work() call (gr-simplefe/lib/sink_c_impl.cc:157-174,
 * source_c_impl.cc:134-153); one synchronous round trip per call is launch/sync bound.  A pipe
 * over a single-channel FIR handle (float32 items, or u8 wire-format items in when the handle's
 * input format is SFE_FMT_U8; float32 items out, or -- output format SFE_FMT_TX10 -- the 10-bit
 * transmit wire format: an output ITEM is then one 5-byte group = 2 complex / 4 real samples, only
 * whole groups are ever sent on their way) collects pushed items in pinned batches of
 * `batch_items` (0 = 262144) and keeps up to four batches in flight on three streams (copy in,
 * filter, copy out); pull hands out finished items in order.  Item k out is the filter's output
 * for item k in: no delay is inserted, only latency.  While a pipe exists, drive its handle only
 * through the pipe: the pipe sized its batches from the handle's item formats, so the handle's
 * format setters that would change them, and its destroy call, return SFE_ESTATE until the pipe
 * is destroyed.
 *   push  copies up to n_items in; *n_taken < n_items means every batch is in flight: pull first.
 *   pull  copies up to max_items finished items out.  wait = 0: only what has already arrived;
 *         1: block for the oldest batch in flight; 2: also send a partly filled batch on its way
 *         and block for it (end of stream / drain).
 *   pending  items pushed and not yet pulled. */
int sfe_dsp_fir_pipe_create(sfe_fir_t fir, size_t batch_items, sfe_pipe_t *out);
/* The same pipe over a single-channel float32 resample / decimate handle at a fixed `rate`
 * ({resample,decimate}::process called chunk after chunk, resample.cxx:85-153): push takes input
 * items, pull hands out the outputs of finished batches -- as many as the reference object would
 * have produced for those inputs.  batch_items is rounded up to whole `blksize`-sample reference
 * calls, so the result is the bulk call's (sfe_dsp_rs_process_stream, incl. sfe_dsp_rs_set_exact)
 * for any rate.  A partly filled batch sent on its way early (pull with wait = 2) is cut on a whole
 * number of blksize-sample calls when the step rate*upsample is not integer-valued -- where a
 * reference caller's call would end -- and what is left (less than one call) stays for the next
 * batch unless it is all there is (then it goes out as the short last call a reference caller
 * makes at the end of a stream).  For an integer-valued step the items do not depend on the cuts.
 * Declared after sfe_rs_t below. */
int sfe_dsp_pipe_push(sfe_pipe_t p, const void *in, size_t n_items, size_t *n_taken);
int sfe_dsp_pipe_pull(sfe_pipe_t p, void *out, size_t max_items, int wait, size_t *n_got);
int sfe_dsp_pipe_pending(sfe_pipe_t p, size_t *items);
/* The same pipe without the two host copies -- what get_process_buf() is to blkconv (libdsp/blkconv.h:44-47:
 * the caller works in the object's own buffer): the producer writes its items straight into the pipe's pinned
 * input batch and the consumer reads finished items in place in the pinned output batch (a callback that
 * receives device samples, simpleFE.c:625-653, or a generator such as bpsk.cxx:145-159 has no buffer of its own
 * to copy from).  May be mixed freely with push / pull on the same pipe; same items, same order.
 *   acquire  *buf = where the next item goes, *room_items = how many fit before the batch is sent on its way
 *            (0, *buf NULL: every batch is in flight -- take finished items out first).  The pointer is valid until
 *            the next commit / push on this pipe, or a pull / peek with wait = 2 (which may send the open batch on
 *            its way and move what is left of it).
 *   commit   n_items (<= room) have been written at the acquired pointer; a full batch is sent on its way.
 *   peek     *out = the oldest finished items, *n_items of them contiguous (0: none ready); wait as for pull.
 *            The pointer is valid until they are released.
 *   release  n_items (<= what peek reported) have been consumed. */
int sfe_dsp_pipe_acquire(sfe_pipe_t p, void **buf, size_t *room_items);
int sfe_dsp_pipe_commit(sfe_pipe_t p, size_t n_items);
int sfe_dsp_pipe_peek(sfe_pipe_t p, const void **out, size_t *n_items, int wait);
int sfe_dsp_pipe_release(sfe_pipe_t p, size_t n_items);
int sfe_dsp_pipe_destroy(sfe_pipe_t p);
/* Fused receive converter: with SFE_FMT_U8 the bulk call reads the device
 * wire format directly -- u8 offset binary, one byte per real sample or an (I,Q) byte pair per
 * complex sample -- converting (b-128)*(1/127) on load exactly as fill_rx_buffer does
 * (gr-simplefe/lib/source_c_impl.cc:121-132, source_f_impl.cc:120-129): 2 bytes instead of 8
 * per complex sample from HBM and no separate conversion pass.  in_stride stays in samples.
 * Applies to *_process_stream only. */
#define SFE_FMT_F32 0
#define SFE_FMT_U8  1
int sfe_dsp_fir_set_input_format(sfe_fir_t h, int fmt);
/* Fused transmit converter: with SFE_FMT_TX10 the bulk call writes the device's transmit wire
 * format instead of floats -- ((short)(x*511)+512)&0x3FF, 4 floats packed in 5 bytes exactly as
 * fill_tx_buffer / convert_samples_to_bytes do on the host (gr-simplefe/lib/sink_f_impl.cc:117-143,
 * sink_c_impl.cc:118-144, examples/bpsk/bpsk.cxx:76-101).  Real stream (real taps): 4 samples per
 * group, d_out receives (n/4)*5 bytes per channel, channel c at byte offset c*(out_stride/4)*5.
 * Complex stream: a group is 2 samples (re, im, re, im), d_out receives (n/2)*5 bytes per
 * channel, channel c at byte offset c*(out_stride/2)*5.  Only whole groups are emitted, as the
 * reference does.  The stream leaves the GPU at 1.25 (real) / 2.5 (complex) instead of 4 / 8
 * bytes per sample. */
#define SFE_FMT_TX10 2
int sfe_dsp_fir_set_output_format(sfe_fir_t h, int fmt);
/* Zero the carried state (a fresh blkconv object: blkconv.cxx:52-55). */
int sfe_dsp_fir_reset(sfe_fir_t h);
/* Replaces blkconv::~blkconv()  libdsp/blkconv.cxx:113-122. */
int sfe_dsp_fir_destroy(sfe_fir_t h);

/* ------------------------------------------------- channel groups: one process, several GPUs
 * The reference's multi-channel form is one object per stream (libdsp/blkconv.h:35-62): channels
 * are independent, so they shard across devices with no exchange.  A group makes
 * that partition inside the library for a caller that owns several GPUs in ONE process:
 *   channels are cut into n_devices contiguous blocks -- block k holds channels
 *   [k*base + min(k, extra), ...) with base = n_channels / n_devices, extra = n_channels %
 *   n_devices, the first `extra` blocks one channel longer -- block k lives on devices[k] (a
 *   device may be named more than once) as an ordinary handle with a stream of its own there.
 * create: taps as sfe_dsp_fir_create's, or (per_channel != 0) [n_channels][n_taps] as
 *   sfe_dsp_fir_create_per_channel's (then the data are complex).  n_devices <= n_channels.
 * process_stream: d_in[k] / d_out[k] are block k's buffers ON devices[k], channel-major with the
 *   given strides, n samples per channel.  The launches go to every device before anything waits
 *   (asynchronous on the blocks' own streams); sync waits for all of them.  A call that fails after
 *   some blocks have taken their launch leaves the blocks out of step: further calls return
 *   SFE_ESTATE until _reset (a failure at the first block -- bad arguments -- has moved nothing).
 * shard: what block k is -- its device, first channel, channel count, the underlying handle (for
 *   sfe_dsp_fir_set_variant / load_history / calibrate ... on it) and its stream (to order the
 *   caller's copies with the block's launches).  Any out pointer may be NULL.
 * The result equals ONE handle over all n_channels on one device, bit for bit. */
typedef void *sfe_fir_group_t;
int sfe_dsp_fir_group_create(const float *taps, int n_taps, int taps_complex, int data_complex, int per_channel,
                             int n_channels, const int *devices, int n_devices, sfe_fir_group_t *out);
int sfe_dsp_fir_group_shards(sfe_fir_group_t g, int *n_shards);
int sfe_dsp_fir_group_shard(sfe_fir_group_t g, int shard, int *device, int *first_channel, int *n_channels,
                            sfe_fir_t *handle, sfe_stream_t *stream);
int sfe_dsp_fir_group_process_stream(sfe_fir_group_t g, const void *const *d_in, void *const *d_out, size_t n,
                                     size_t in_stride, size_t out_stride);
int sfe_dsp_fir_group_sync(sfe_fir_group_t g);
int sfe_dsp_fir_group_reset(sfe_fir_group_t g);
int sfe_dsp_fir_group_destroy(sfe_fir_group_t g);

/* -------------------------------------------------------------- resample / decimate
 * Polyphase interpolating resampler: outputs at upsampled-grid instants t (float32
 * recurrence t += rate*upsample), pos = floor(t), mu = t - pos,
 *   out = s(pos)*(1-mu) + mu*s(pos+1),  s(p) = sum_j taps[p%U + j*U] * x[p/U - j]
 * (libdsp/resample.cxx:85-153, libdsp/decimate.cxx:69-140; the two classes give identical
 * output, libdsp/test/test_decimate.py:36). */
#define SFE_RS_RESAMPLE 0   /* accepts rate >= 1/upsample  (resample.cxx:91) */
#define SFE_RS_DECIMATE 1   /* accepts rate >= 1           (decimate.cxx:75) */

/* Replaces resample::resample / decimate::decimate (float *taps, int n_taps, int upsample,
 * int blksize)   libdsp/resample.cxx:37-69, libdsp/decimate.cxx:37-59. */
int sfe_dsp_rs_create(const float *taps, int n_taps, int upsample, int blksize,
                      int data_complex, int n_channels, int device, int mode, sfe_rs_t *out);
/* Replaces {resample,decimate}::process(float *in, int n_in, float *out, int out_len,
 * float rate)   libdsp/resample.cxx:85-153, libdsp/decimate.cxx:69-129.  Host pointers,
 * one channel, synchronous.  Same parameter checks, same messages on stdout and *n_out = 0
 * (resample.cxx:91-98, decimate.cxx:75-87); same leftover / time-recurrence state. */
int sfe_dsp_rs_process(sfe_rs_t h, const float *in, int n_in, float *out, int out_len,
                       float rate, int *n_out);
/* Bulk device-resident form: consumes n_in samples per channel, writes *n_out samples per
 * channel (same count for every channel), equal to what the reference object produces when
 * fed the same stream in chunks of `blksize`.  When fl(rate*upsample) is integer-valued the
 * result does not depend on the chunking and one closed-form launch is used; otherwise the
 * float32 time recurrence is replayed on the host, chunk by chunk, and uploaded.
 * Fails with SFE_ERANGE if out_cap is too small (nothing is written).  Asynchronous.
 * Strides are in samples; with n_channels > 1, in_stride >= n_in and out_stride >= out_cap
 * (up to out_cap outputs per channel may be written).  Buffers are aligned to their element
 * (complex float32 8 bytes, real float32 4, SFE_FMT_U8 (I,Q) pairs 2, real u8 none) and the
 * input and output byte ranges -- (n_channels-1)*stride + n_in resp. out_cap elements -- must
 * not overlap; violations return SFE_EINVAL before anything is launched.  Any such alignment gives the same bits; channels whose
 * first sample sits on a 16-byte boundary (sfe_dsp_malloc memory, strides that are multiples of 16 bytes) take the kernels that fetch
 * their tiles by 16-byte DMA lanes -- 15-35 % faster at ratios without a compile-time kernel, up to 3x for u8 input.
 * SFE_FMT_U8 input is accepted at every rate the float32 path takes: where no kernel converts on load (steps of more than 64 samples, more
 * than 8 outputs per period, a general rate with a small blksize) the call converts its bytes once into a scratch buffer the handle owns and
 * grows (4 x the call's input bytes; not while the stream is being captured) and runs the float32 path -- the same bits. */
int sfe_dsp_rs_process_stream(sfe_rs_t h, const void *d_in, size_t n_in, size_t in_stride,
                              void *d_out, size_t out_cap, size_t out_stride, float rate,
                              size_t *n_out, sfe_stream_t stream);
/* Which kernel the bulk path (fused numerics) takes.  Integer-valued steps -- AUTO: the calibrated
 * rule of DESIGN.md 4.2c (transform-domain kernel for long filters on long calls, else a tiled direct
 * kernel: a compile-time (inputs, outputs)-per-period instantiation or, for any other ratio up to
 * 64 : 8, the runtime-shape one).  DIRECT: never a transform-domain kernel.  FFT: the
 * transform-domain kernel wherever the shape is instantiated.  MFMA: the f32 matrix-pipe form of the
 * direct kernel (measured slower, kept as evidence: DESIGN.md 4.2b).
 * Any other step (the general rate) -- AUTO: float32 calls (complex or real) of >= 65536 samples
 * with >= 12 taps per phase take the 4096-point transform kernel (all phases of every
 * input sample by one forward and `upsample` inverse transforms, blended by the reference's own
 * (pos, mu) sequence; a real stream's transforms carry two blocks each), at any rate the class
 * accepts -- below 1 too, down to 1 / upsample; FFT: at any size; DIRECT and the exact mode: the
 * direct kernel.  u8 streams (SFE_FMT_U8) at such a step ALWAYS take the transform kernel, which
 * converts on load (the direct kernels have no u8 form: the exact mode and DIRECT return
 * SFE_ESTATE for them).  The number of outputs per call is the reference's in every case.
 * The library reads no environment variable. */
#define SFE_RS_ALGO_AUTO    0
#define SFE_RS_ALGO_DIRECT  1
#define SFE_RS_ALGO_FFT     2
#define SFE_RS_ALGO_MFMA    3
int sfe_dsp_rs_set_algo(sfe_rs_t h, int algo);
/* exact = 1: separate multiply and add in the reference's order (bit-exact with the CPU
 * classes); exact = 0 (default for *_stream): fused multiply-add, same order. */
int sfe_dsp_rs_set_exact(sfe_rs_t h, int exact);
/* As sfe_dsp_fir_set_input_format, for the integer-step bulk path (fused numerics). */
int sfe_dsp_rs_set_input_format(sfe_rs_t h, int fmt);
/* One stream cut into spans.  load_history: as sfe_dsp_fir_load_history (the resamplers'
 * m_history, resample.h:49-59 / decimate.h:50-59; phase_len samples suffice).  seek: put the time
 * state {pos, mu, leftover} (resample.h:55-59) where a reference object stands after consuming
 * `first_sample` samples from a fresh start -- closed form, available only when fl(rate*upsample) is
 * integer-valued (both BASELINE resampler configs); otherwise SFE_ESTATE: the float32 recurrence
 * t += rate*U (resample.cxx:129-150) must be carried, which get_state / set_state do (a span's
 * final state is the next span's initial one). */
int sfe_dsp_rs_load_history(sfe_rs_t h, const void *d_prev, size_t n_prev, size_t stride,
                            sfe_stream_t stream);
int sfe_dsp_rs_seek(sfe_rs_t h, uint64_t first_sample, float rate);
/* the host pipe (sfe_dsp_pipe_*) over a single-channel resample / decimate handle at `rate`: batches of whole
 * blksize-sample reference calls; items in are float32 samples, or u8 wire-format items when the handle's input
 * format is SFE_FMT_U8 (integer-valued steps); items out float32 */
int sfe_dsp_rs_pipe_create(sfe_rs_t rs, size_t batch_items, float rate, sfe_pipe_t *out);
int sfe_dsp_rs_reset(sfe_rs_t h);
int sfe_dsp_rs_destroy(sfe_rs_t h);
/* Channel groups of resample / decimate objects (one reference object per stream,
 * libdsp/resample.h:33-61, libdsp/decimate.h:33-63): the same partition, calls and guarantees as
 * sfe_dsp_fir_group_* above.  All blocks are fed in lockstep, so every channel produces the same
 * *n_out; a block handle driven on its own makes the next group call fail with SFE_ESTATE. */
typedef void *sfe_rs_group_t;
int sfe_dsp_rs_group_create(const float *taps, int n_taps, int upsample, int blksize, int data_complex, int n_channels,
                            const int *devices, int n_devices, int mode, sfe_rs_group_t *out);
int sfe_dsp_rs_group_shards(sfe_rs_group_t g, int *n_shards);
int sfe_dsp_rs_group_shard(sfe_rs_group_t g, int shard, int *device, int *first_channel, int *n_channels,
                           sfe_rs_t *handle, sfe_stream_t *stream);
int sfe_dsp_rs_group_process_stream(sfe_rs_group_t g, const void *const *d_in, size_t n_in, size_t in_stride,
                                    void *const *d_out, size_t out_cap, size_t out_stride, float rate, size_t *n_out);
int sfe_dsp_rs_group_sync(sfe_rs_group_t g);
int sfe_dsp_rs_group_reset(sfe_rs_group_t g);
int sfe_dsp_rs_group_destroy(sfe_rs_group_t g);

/* Host-only: replay the time law of ONE process() call without touching the GPU
 * (resample.cxx:89,119-150).  state = {pos, mu, leftover} in/out.  Writes up to cap
 * entries: rel_pos[k] = upsampled position of output k relative to the chunk start (-1 for
 * a leftover output), mu[k] its interpolation weight.  Returns the output count in *n_out. */
typedef struct {
    int32_t pos;
    float   mu;
    int32_t leftover;
} sfe_rs_timestate;
int sfe_dsp_rs_plan(sfe_rs_timestate *state, int upsample, int n_in, int out_len, float rate,
                    int32_t *rel_pos, float *mu, int cap, int *n_out);
/* The handle's time state (m_pos, m_mu, m_is_leftover: libdsp/resample.h:55-59), read / written
 * between calls: lets a stream be cut at ANY rate by carrying one span's final state to the next. */
/* Host-only form of sfe_dsp_rs_seek (no handle, no GPU): the state after `first_sample` samples. */
int sfe_dsp_rs_plan_seek(sfe_rs_timestate *state, int upsample, uint64_t first_sample, float rate);
int sfe_dsp_rs_get_state(sfe_rs_t h, sfe_rs_timestate *state);
int sfe_dsp_rs_set_state(sfe_rs_t h, const sfe_rs_timestate *state);

/* ------------------------------------------- wire-format converters ("next" row N2)
 * RX: u8 offset-binary -> float32 (b-128)*(1/127)
 *     gr-simplefe/lib/source_c_impl.cc:121-132, source_f_impl.cc:120-129.
 * TX: float32 -> 10-bit offset binary ((short)(x*511)+512)&0x3FF, 4 samples in 5 bytes
 *     gr-simplefe/lib/sink_c_impl.cc:118-144, sink_f_impl.cc:117-143,
 *     examples/bpsk/bpsk.cxx:76-101.   Device pointers, asynchronous. */
int sfe_dsp_rx_u8_to_f32(const void *d_bytes, void *d_floats, size_t n_bytes, sfe_stream_t stream);
int sfe_dsp_tx_f32_to_10bit(const void *d_floats, void *d_bytes, size_t n_floats, sfe_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SFE_DSP_H_ */
