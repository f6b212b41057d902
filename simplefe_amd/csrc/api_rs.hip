// api_rs.hip -- the resample / decimate handle behind sfe_rs_t (libdsp/resample.cxx:37-153, libdsp/decimate.cxx:37-140):
// the time law and its run memo, the choice of kernel, carried state, and the sfe_dsp_rs_* entry points.  Host code only.
#include "host.h"
#include <stdlib.h>

namespace sfe {

// ------------------------------------------------------------------ resample / decimate

Rs *as_rs(void *h)
{
    Rs *r = static_cast<Rs *>(h);
    if (r && r->magic != 0x52533031u) {
        set_error("not a live resample/decimate handle");
        return nullptr;
    }
    return r;
}

void rs_free(Rs *r)
{
    if (!r) return;
    r->magic = 0;
    DeviceGuard g(r->device);
    if (r->d_taps) (void)hipFree(r->d_taps);
    r->plans.clear();
    r->mfma_plans.clear();
    r->fft_plans.clear();
    if (r->d_ticket) (void)hipFree(r->d_ticket);
    if (r->gen_tables) fir_free(r->gen_tables);
    for (int i = 0; i < 2; i++)
        if (r->d_hist[i]) (void)hipFree(r->d_hist[i]);
    if (r->d_in) (void)hipFree(r->d_in);
    if (r->d_out) (void)hipFree(r->d_out);
    if (r->d_pos) (void)hipFree(r->d_pos);
    if (r->d_mu) (void)hipFree(r->d_mu);
    if (r->h_stage) (void)hipHostFree(r->h_stage);
    if (r->d_u8f) (void)hipFree(r->d_u8f);
    if (r->h_pos) (void)hipHostFree(r->h_pos);
    if (r->h_mu) (void)hipHostFree(r->h_mu);
    if (r->d_segs) (void)hipFree(r->d_segs);
    if (r->d_chunks) (void)hipFree(r->d_chunks);
    if (r->h_segs) (void)hipHostFree(r->h_segs);
    if (r->h_chunks) (void)hipHostFree(r->h_chunks);
    if (r->ev_plan) (void)hipEventDestroy(r->ev_plan);
    if (r->stream) (void)hipStreamDestroy(r->stream);
    delete r;
}

static int rs_ensure_sched(Rs *r, size_t n)
{
    if (n <= r->sched_cap) return SFE_OK;
    size_t cap = r->sched_cap ? r->sched_cap : 1024;
    while (cap < n) cap *= 2;
    if (r->d_pos) (void)hipFree(r->d_pos);
    if (r->d_mu) (void)hipFree(r->d_mu);
    if (r->h_pos) (void)hipHostFree(r->h_pos);
    if (r->h_mu) (void)hipHostFree(r->h_mu);
    r->d_pos = nullptr; r->d_mu = nullptr; r->h_pos = nullptr; r->h_mu = nullptr;
    r->sched_cap = 0;
    SFE_HIP(hipMalloc(&r->d_pos, cap * sizeof(long long)));
    SFE_HIP(hipMalloc(&r->d_mu, cap * sizeof(float)));
    SFE_HIP(hipHostMalloc(&r->h_pos, cap * sizeof(long long)));
    SFE_HIP(hipHostMalloc(&r->h_mu, cap * sizeof(float)));
    r->sched_cap = cap;
    return SFE_OK;
}

static int rs_ensure_out(Rs *r, size_t n)
{
    if (n <= r->out_cap) return SFE_OK;
    size_t cap = r->out_cap ? r->out_cap : 1024;
    while (cap < n) cap *= 2;
    if (r->d_out) (void)hipFree(r->d_out);
    r->d_out = nullptr;
    r->out_cap = 0;
    SFE_HIP(hipMalloc(&r->d_out, cap * r->esz()));
    r->out_cap = cap;
    return SFE_OK;
}

static int rs_ensure_stage(Rs *r, size_t bytes)
{
    if (bytes <= r->h_stage_bytes) return SFE_OK;
    if (r->h_stage) (void)hipHostFree(r->h_stage);
    r->h_stage = nullptr;
    r->h_stage_bytes = 0;
    SFE_HIP(hipHostMalloc(&r->h_stage, bytes));
    r->h_stage_bytes = bytes;
    return SFE_OK;
}

}  // namespace sfe

using namespace sfe;

extern "C" {

// ------------------------------------------------------------------ resample / decimate
int sfe_dsp_rs_plan(sfe_rs_timestate *state, int upsample, int n_in, int out_len, float rate,
                    int32_t *rel_pos, float *mu, int cap, int *n_out)
{
    if (!state || upsample < 1 || n_in < 0 || !n_out) return SFE_EINVAL;
    int overflow = 0, k = 0;
    const int n = time_law(state, upsample, n_in, out_len, rate, [&](int p, float m) {
        if (k < cap) {
            if (rel_pos) rel_pos[k] = p;
            if (mu) mu[k] = m;
        } else overflow = 1;
        k++;
    });
    *n_out = n;
    return overflow ? SFE_ERANGE : SFE_OK;
}

int sfe_dsp_rs_create(const float *taps, int n_taps, int upsample, int blksize, int data_complex,
                      int n_channels, int device, int mode, sfe_rs_t *out)
{
    if (!out) return SFE_EINVAL;
    *out = nullptr;
    if (!taps || n_taps < 1 || upsample < 1 || blksize < 1 || n_channels < 1 ||
        (mode != SFE_RS_RESAMPLE && mode != SFE_RS_DECIMATE)) {
        set_error("rs_create: bad arguments");
        return SFE_EINVAL;
    }
    int prev_dev = -1;
    (void)hipGetDevice(&prev_dev);
    int rc = use_device(device);
    if (rc != SFE_OK) return rc;
    struct Restore { int d; ~Restore() { if (d >= 0) (void)hipSetDevice(d); } } restore__{prev_dev};
    Rs *r = new (std::nothrow) Rs;
    if (!r) return SFE_ENOMEM;
    r->U = upsample;
    r->n_taps = n_taps;
    r->blksize = blksize;
    r->data_complex = data_complex ? 1 : 0;
    r->n_channels = n_channels;
    r->device = device;
    r->mode = mode;
    // decimate appends a zero tap when n_taps is even (decimate.cxx:42-51); resample pads the
    // last phase with zeros (resample.cxx:43,55-64).  Both are "ceil to a whole phase row".
    const int eff = (mode == SFE_RS_DECIMATE && (n_taps % 2 == 0)) ? n_taps + 1 : n_taps;
    r->plen = (eff + upsample - 1) / upsample;
    r->hl = ((r->plen + 1 + 63) / 64) * 64;
    auto fail = [&](int code) { rs_free(r); return code; };
#define TRY(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return fail(hip_fail(e__, #call)); } while (0)
    TRY(hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking));
    std::vector<float> pm((size_t)upsample * r->plen, 0.0f);
    for (int j = 0; j < upsample; j++)
        for (int i = 0; i < r->plen; i++) {
            const int n = i * upsample + j;
            pm[(size_t)j * r->plen + i] = n < n_taps ? taps[n] : 0.0f;
        }
    r->h_taps_pm = pm;
    TRY(hipMalloc(&r->d_taps, pm.size() * sizeof(float)));
    TRY(hipMemcpy(r->d_taps, pm.data(), pm.size() * sizeof(float), hipMemcpyHostToDevice));
    const size_t hb = (size_t)n_channels * r->hl * r->esz();
    for (int i = 0; i < 2; i++) {
        TRY(hipMalloc(&r->d_hist[i], hb));
        TRY(hipMemset(r->d_hist[i], 0, hb));
    }
    TRY(hipMalloc(&r->d_in, (size_t)blksize * r->esz()));
    TRY(hipMalloc(&r->d_ticket, POLY_TICKET_GROUPS * 128));
    TRY(hipMemset(r->d_ticket, 0, POLY_TICKET_GROUPS * 128));
    TRY(hipDeviceSynchronize());
#undef TRY
    *out = r;
    return SFE_OK;
}

int sfe_dsp_rs_process(sfe_rs_t h, const float *in, int n_in, float *out, int out_len, float rate,
                       int *n_out)
{
    Rs *r = as_rs(h);
    if (!r || !n_out) return SFE_EINVAL;
    *n_out = 0;
    if (r->n_channels != 1) {
        set_error("rs_process: the host-pointer call is single-channel");
        return SFE_EINVAL;
    }
    // parameter checks, messages and "return 0 outputs" as the reference
    if (r->mode == SFE_RS_RESAMPLE) {
        if (n_in > r->blksize || rate < 1.0 / r->U) {                      // resample.cxx:91-94
            printf("input parameter is wrong, rate <= 1/upsample, n_in <= blksize\n");
            return SFE_OK;
        }
    } else {
        if (rate < 1.0) {                                                  // decimate.cxx:75-78
            printf("rate should be larger than 1.0\n");
            return SFE_OK;
        }
        if (n_in > r->blksize) {                                           // decimate.cxx:79-82
            printf("number of samples should be less than blksize\n");
            return SFE_OK;
        }
    }
    if (out_len < floorf(n_in * 1.0f / rate)) {                            // resample.cxx:95-98
        printf("output buffer is not large enough");
        return SFE_OK;
    }
    if (n_in < 0 || out_len < 0 || (n_in && !in) || (out_len && !out)) return SFE_EINVAL;
    SFE_ON_DEVICE(r->device);

    // The common call -- out_len roomy enough that the law, not the buffer, ends the outputs -- is the
    // bulk path on one block with the reference's arithmetic order (exact): the tiled / run-length kernels
    // stage the block in LDS instead of two global dot products per output, the outputs are written
    // straight into pinned host memory, and the stream sees copy-in, kernel, history instead of seven
    // operations.  Same bits (tests/test_gpu_parity.py: class calls against the compiled reference).
    if (n_in > 0 && !r->in_u8) {
        const float stepf = rate * (float)r->U;
        const bool int_step = stepf >= 1.0f && stepf == floorf(stepf) && stepf < 1.0e6f && r->ts.mu == 0.0f &&
                              ((double)r->blksize * r->U + stepf) < 16777216.0;
        bool roomy;
        if (int_step) {
            const long long S = (long long)stepf, pos0 = r->ts.leftover ? -1 : (long long)r->ts.pos;
            const long long lim = (long long)n_in * r->U - 2;
            roomy = (pos0 <= lim ? (lim - pos0) / S + 1 : 0) <= (long long)out_len;
        } else
            roomy = (long long)out_len >= (long long)ceilf((float)n_in / rate) + 2;
        if (roomy) {
            const size_t in_b = (size_t)n_in * r->esz(), out_off = (in_b + 255) & ~(size_t)255;
            int rc = rs_ensure_stage(r, out_off + ((size_t)out_len + 1) * r->esz());
            if (rc != SFE_OK) return rc;
            char *h_out = static_cast<char *>(r->h_stage) + out_off;
            memcpy(r->h_stage, in, in_b);
            SFE_HIP(hipMemcpyAsync(r->d_in, r->h_stage, in_b, hipMemcpyHostToDevice, r->stream));
            const int keep = r->exact_stream;
            r->exact_stream = 1;
            size_t n = 0;
            rc = sfe_dsp_rs_process_stream(h, r->d_in, (size_t)n_in, (size_t)n_in, h_out, (size_t)out_len, (size_t)out_len,
                                           rate, &n, r->stream);
            r->exact_stream = keep;
            if (rc != SFE_OK) return rc;
            SFE_HIP(hipStreamSynchronize(r->stream));
            if (n) memcpy(out, h_out, n * r->esz());
            *n_out = (int)n;
            return SFE_OK;
        }
    }

    int rc = rs_ensure_sched(r, (size_t)out_len + 1);
    if (rc != SFE_OK) return rc;
    rc = rs_ensure_out(r, (size_t)out_len + 1);
    if (rc != SFE_OK) return rc;
    const size_t in_b = (size_t)n_in * r->esz();
    const size_t out_b = ((size_t)out_len + 1) * r->esz();
    rc = rs_ensure_stage(r, in_b > out_b ? in_b : out_b);
    if (rc != SFE_OK) return rc;

    int k = 0;
    const int n = time_law(&r->ts, r->U, n_in, out_len, rate, [&](int p, float m) {
        r->h_pos[k] = p;
        r->h_mu[k] = m;
        k++;
    });
    if (n_in) {
        memcpy(r->h_stage, in, in_b);
        SFE_HIP(hipMemcpyAsync(r->d_in, r->h_stage, in_b, hipMemcpyHostToDevice, r->stream));
    }
    if (n > 0) {
        SFE_HIP(hipMemcpyAsync(r->d_pos, r->h_pos, (size_t)n * sizeof(long long), hipMemcpyHostToDevice, r->stream));
        SFE_HIP(hipMemcpyAsync(r->d_mu, r->h_mu, (size_t)n * sizeof(float), hipMemcpyHostToDevice, r->stream));
        PolyArgs a;
        memset(&a, 0, sizeof(a));
        a.in = r->d_in;
        a.out = r->d_out;
        a.hist = r->d_hist[r->cur];
        a.taps = r->d_taps;
        a.n_in = n_in;
        a.in_stride = n_in;
        a.out_stride = n;
        a.hl = r->hl;
        a.U = r->U;
        a.plen = r->plen;
        a.n_out = n;
        a.sched_pos = r->d_pos;
        a.sched_mu = r->d_mu;
        rc = launch_poly_sched(a, r->data_complex, 1, 1, r->stream);
        if (rc != SFE_OK) return rc;
    }
    rc = launch_history_update(r->d_in, n_in, n_in, r->d_hist[r->cur], r->d_hist[r->cur ^ 1], r->hl,
                               r->data_complex ? 2 : 1, 1, r->stream);
    if (rc != SFE_OK) return rc;
    if (r->captured)
        SFE_HIP(hipMemcpyAsync(r->d_hist[r->cur], r->d_hist[r->cur ^ 1], (size_t)r->hl * r->esz(), hipMemcpyDeviceToDevice, r->stream));
    else
        r->cur ^= 1;
    if (n > 0) SFE_HIP(hipMemcpyAsync(r->h_stage, r->d_out, (size_t)n * r->esz(), hipMemcpyDeviceToHost, r->stream));
    SFE_HIP(hipStreamSynchronize(r->stream));
    if (n > 0) memcpy(out, r->h_stage, (size_t)n * r->esz());
    *n_out = n;
    return SFE_OK;
}

static int rs_process_stream_impl(sfe_rs_t h, const void *d_in, size_t n_in, size_t in_stride,
                                  void *d_out, size_t out_cap, size_t out_stride, float rate,
                                  size_t *n_out, sfe_stream_t stream);

int sfe_dsp_rs_process_stream(sfe_rs_t h, const void *d_in, size_t n_in, size_t in_stride,
                              void *d_out, size_t out_cap, size_t out_stride, float rate,
                              size_t *n_out, sfe_stream_t stream)
{
    Rs *r = as_rs(h);
    if (r) r->u8_refused = false;
    int rc = rs_process_stream_impl(h, d_in, n_in, in_stride, d_out, out_cap, out_stride, rate, n_out, stream);
    if (rc != SFE_ESTATE || !r || !r->in_u8 || !r->u8_refused) return rc;
    // Wire-format input on a shape that has no fused u8 kernel (integer steps beyond the tiled kernels' -- more than 64 samples or 8 outputs per
    // period --, a general rate the transform-domain kernel does not take): until round 5 such a call was REFUSED.  Nothing has been launched and
    // no state touched: convert the call's bytes once (the same (b - 128) * (1 / 127) as every fused path: the same bits) and run the float path.
    SFE_ON_DEVICE(r->device);
    hipStream_t s = (hipStream_t)stream;
    const size_t w = r->data_complex ? 2 : 1, floats = (size_t)r->n_channels * n_in * w;
    if (floats > r->d_u8f_floats) {
        if (stream_is_capturing(s)) {
            set_error("rs_process_stream: this u8 call needs %zu bytes of conversion scratch, which cannot be allocated while the stream is being captured",
                      floats * 4);
            return SFE_ESTATE;
        }
        if (r->d_u8f) (void)hipFree(r->d_u8f);
        r->d_u8f = nullptr;
        r->d_u8f_floats = 0;
        SFE_HIP(hipMalloc(&r->d_u8f, floats * sizeof(float)));
        r->d_u8f_floats = floats;
    } else if (r->u8f_stream != s && r->u8f_stream) {
        SFE_HIP(hipStreamSynchronize(r->u8f_stream));         // the previous call on another stream may still read the scratch
    }
    r->u8f_stream = s;
    for (int c = 0; c < r->n_channels; c++) {
        rc = launch_rx_u8_to_f32(static_cast<const uint8_t *>(d_in) + (size_t)c * in_stride * w, r->d_u8f + (size_t)c * n_in * w, n_in * w, s);
        if (rc != SFE_OK) return rc;
    }
    r->in_u8 = 0;
    rc = rs_process_stream_impl(h, r->d_u8f, n_in, n_in, d_out, out_cap, out_stride, rate, n_out, stream);
    r->in_u8 = 1;
    return rc;
}

static int rs_process_stream_impl(sfe_rs_t h, const void *d_in, size_t n_in, size_t in_stride,
                                  void *d_out, size_t out_cap, size_t out_stride, float rate,
                                  size_t *n_out, sfe_stream_t stream)
{
    Rs *r = as_rs(h);
    if (!r || !n_out) return SFE_EINVAL;
    *n_out = 0;
    if (r->mode == SFE_RS_RESAMPLE ? (rate < 1.0 / r->U) : (rate < 1.0)) {
        set_error("rs_process_stream: rate %g not accepted by this mode", (double)rate);
        return SFE_EINVAL;
    }
    if (n_in == 0) return SFE_OK;
    if (!d_in || !d_out) {
        set_error("rs_process_stream: null buffer");
        return SFE_EINVAL;
    }
    if (r->n_channels > 1 && (in_stride < n_in || out_stride < out_cap)) {
        // out_cap outputs per channel may be written: a smaller stride would let channels overwrite each other
        set_error("rs_process_stream: channel stride smaller than the channel (in_stride >= n_in, out_stride >= out_cap)");
        return SFE_EINVAL;
    }
    {
        const size_t isz = r->in_u8 ? (r->data_complex ? 2 : 1) : (size_t)r->esz();
        const size_t osz = (size_t)r->esz();
        if ((reinterpret_cast<uintptr_t>(d_in) & (isz - 1)) || (reinterpret_cast<uintptr_t>(d_out) & (osz - 1))) {
            set_error("rs_process_stream: buffers must be aligned to their element (cf32 8 B, f32 4 B, u8 (I,Q) pairs 2 B)");
            return SFE_EINVAL;
        }
        const size_t in_b = ((size_t)(r->n_channels - 1) * in_stride + n_in) * isz;
        const size_t out_b = ((size_t)(r->n_channels - 1) * out_stride + out_cap) * osz;
        if (ranges_overlap(d_in, in_b, d_out, out_b)) {
            set_error("rs_process_stream: input and output ranges overlap");
            return SFE_EINVAL;
        }
    }
    SFE_ON_DEVICE(r->device);
    hipStream_t s = (hipStream_t)stream;
    const float stepf = rate * (float)r->U;
    const bool int_step = stepf >= 1.0f && stepf == floorf(stepf) && stepf < 1.0e6f && r->ts.mu == 0.0f &&
                          ((double)r->blksize * r->U + stepf) < 16777216.0;
    const bool capturing = stream_is_capturing(s);
    if (capturing && (!int_step || n_in < (size_t)r->hl || ((unsigned long long)n_in * (unsigned long long)r->U) % (unsigned long long)stepf != 0)) {
        set_error("rs_process_stream: a call captured into a hipGraph must leave the time state where it found it "
                  "(integer-valued step, n_in*upsample a multiple of it) and bring at least %d samples", r->hl);
        return SFE_ESTATE;
    }
    PolyArgs a;
    memset(&a, 0, sizeof(a));
    a.in = d_in;
    a.out = d_out;
    a.hist = r->d_hist[r->cur];
    a.taps = r->d_taps;
    a.n_in = (long long)n_in;
    a.in_stride = (long long)in_stride;
    a.out_stride = (long long)out_stride;
    a.hl = r->hl;
    a.U = r->U;
    a.plen = r->plen;
    int rc;
    bool hist_fused = false;
    if (int_step) {
        // closed form of the law: output k at pos0 + k*S, emitted while pos <= n_in*U - 2
        // (pos == n_in*U - 1 is the reference's "leftover": it comes out first next call).
        const long long S = (long long)stepf;
        const long long pos0 = r->ts.leftover ? -1 : (long long)r->ts.pos;
        const long long lim = (long long)n_in * r->U - 2;
        const long long K = pos0 <= lim ? (lim - pos0) / S + 1 : 0;
        if ((size_t)K > out_cap) {
            set_error("rs_process_stream: need room for %lld outputs, got %zu", K, out_cap);
            return SFE_ERANGE;
        }
        a.pos0 = pos0;
        a.step = (int)S;
        a.n_out = K;
        // matrix-pipe form (fused numerics, cf32): opt-in with sfe_dsp_rs_set_algo(SFE_RS_ALGO_MFMA).  Measured slower
        // than the VALU kernel on the one shape where its tap matrix is dense (polyphase.hip).
        const PolyMfmaPlan *mp = nullptr;
        if (r->use_mfma && !r->exact_stream && r->data_complex && !r->in_u8) {
            mp = get_mfma_plan(r->mfma_plans, r->h_taps_pm, r->U, r->plen, (int)S, pos0, &rc);
            if (rc != SFE_OK) return rc;
        }
        // transform-domain form (fused numerics, cf32): long filters on streams long enough to fill the chip
        const PolyFftPlan *fp = nullptr;
        if (!mp && !r->exact_stream && K >= 4096) {
            fp = get_fft_plan(r->fft_plans, r->h_taps_pm, r->U, r->plen, (int)S, pos0, r->fft_mode, &rc);
            if (rc != SFE_OK) return rc;
        }
        const PolyTiledPlan *pl = (mp || fp) ? nullptr : get_tiled_plan(r->plans, r->h_taps_pm, r->U, r->plen, (int)S, pos0, &rc);
        if (rc != SFE_OK) return rc;
        // the transform-domain and the tiled kernels write the next call's history themselves (one launch per call)
        const bool can_fuse = n_in >= (size_t)r->hl && K > 0 && !capturing;
        if (fp) {
            PolyFftArgs fa;
            memset(&fa, 0, sizeof(fa));
            fa.in = d_in;
            fa.out = d_out;
            fa.hist = r->d_hist[r->cur];
            fa.hist_out = can_fuse ? r->d_hist[r->cur ^ 1] : nullptr;
            hist_fused = can_fuse;
            fa.H = fp->d_H;
            fa.tw = fp->d_tw;
            fa.n_in = (long long)n_in;
            fa.in_stride = (long long)in_stride;
            fa.out_stride = (long long)out_stride;
            fa.n_out = K;
            fa.hl = r->hl;
            fa.e_max = fp->e_max;
            fa.ovl = fp->Li - 1;
            fa.V = 256 - fa.ovl;
            fa.ticket = r->d_ticket;
            rc = launch_poly_fft(*fp, fa, r->data_complex, r->in_u8, r->n_channels, s);
        } else if (mp) {
            PolyMfmaArgs ma;
            memset(&ma, 0, sizeof(ma));
            ma.in = d_in;
            ma.out = d_out;
            ma.hist = r->d_hist[r->cur];
            ma.A = mp->d_A;
            ma.n_in = (long long)n_in;
            ma.in_stride = (long long)in_stride;
            ma.out_stride = (long long)out_stride;
            ma.n_out = K;
            ma.hl = r->hl;
            ma.GS = mp->GS;
            ma.RG = mp->RG;
            ma.Kp = mp->Kp;
            ma.u_lo = mp->u_lo;
            rc = launch_poly_mfma(ma, r->n_channels, s);
        } else if (pl) {
            PolyTiledArgs ta;
            ta.in = d_in;
            ta.out = d_out;
            ta.hist = r->d_hist[r->cur];
            ta.hist_out = can_fuse && !r->exact_stream ? r->d_hist[r->cur ^ 1] : nullptr;      // exact kernels: separate carry-over launch
            hist_fused = ta.hist_out != nullptr;
            ta.G = pl->d_G;
            ta.Gt = pl->d_Gt;
            ta.n_in = (long long)n_in;
            ta.in_stride = (long long)in_stride;
            ta.out_stride = (long long)out_stride;
            ta.n_out = K;
            ta.hl = r->hl;
            ta.Lp = pl->Lp;
            ta.e_max = pl->e_max;
            // shapes outside the compiled tables, float32 streams (complex or real), fused arithmetic: the tile by LDS-DMA, read in place (poly_rt_dma.hip)
            rc = SFE_ESTATE;
            bool window_shape = !r->data_complex && poly_rt_dma_window_shape(pl->SP, pl->UP);      // real streams at small steps: the register-window kernel
#ifdef SFE_DIAG
            if (const char *e = getenv("SFE_RT_DMA_WINDOW")) window_shape = window_shape && pl->SP <= atoi(e);
#endif
            // wire-format input: four shapes have compile-time u8 kernels (/2, /4, /8, 5/3) and keep them; every other one -- the compile-time
            // float shapes included -- ran poly_rt_kernel's sample-by-sample byte loads (/7 0.71 ms where the float32 stream takes 0.45) and now
            // has its raw tile fetched by DMA and converted once (0.38 ms): profiles/r05/shapes_u8.txt
            const bool compiled = r->in_u8 ? poly_tiled_u8_is_compiled(pl->SP, pl->UP, pl->Lp) : poly_tiled_is_compiled(pl->SP, pl->UP, pl->Lp);
            bool try_dma = !r->exact_stream && (!compiled || window_shape);       // (real u8 streams at the window form's shapes: /2 0.72 -> 0.6 ms, profiles/r05/shapes_u8.txt)
#ifdef SFE_DIAG
            if (const char *e = getenv("SFE_RT_DMA_FORCE"))        // scripts/ab_dec8_dma.py: the LDS-DMA form also where a compile-time kernel exists
                if (atoi(e) && !r->exact_stream) try_dma = true;
            if (const char *e = getenv("SFE_RT_DMA_U8"))           // =0: wire-format input keeps poly_rt_kernel / the compile-time kernels (scripts/time_u8_shapes.py)
                if (!atoi(e) && r->in_u8) try_dma = false;
#endif
            if (try_dma) rc = launch_poly_rt_dma(*pl, ta, r->data_complex, r->in_u8, r->n_channels, s);
            if (rc == SFE_ESTATE && pl->UP <= 8) rc = launch_poly_tiled(*pl, ta, r->data_complex, r->exact_stream, r->in_u8, r->n_channels, s);
            if (rc == SFE_ESTATE && pl->UP > 8) {            // 9 ... 256 outputs per period and the tiled form declined: the generic kernel
                if (r->in_u8) {
                    r->u8_refused = true;
                    set_error("rs_process_stream: u8 input needs a tiled kernel for this rate/tap shape");
                    return SFE_ESTATE;
                }
                hist_fused = false;
                rc = launch_poly_int(a, r->data_complex, 0, r->exact_stream, r->n_channels, s);
            }
        } else {
            if (r->in_u8) {
                r->u8_refused = true;          // (the public entry converts the bytes and comes back with float32)
                set_error("rs_process_stream: u8 input needs a tiled kernel for this rate/tap shape");
                return SFE_ESTATE;
            }
            rc = launch_poly_int(a, r->data_complex, 0, r->exact_stream, r->n_channels, s);
        }
        if (rc != SFE_OK) return rc;
        const long long next = pos0 + K * S - (long long)n_in * r->U;
        r->ts.leftover = next == -1 ? 1 : 0;
        r->ts.pos = (int32_t)next;
        r->ts.mu = 0.0f;
        *n_out = (size_t)K;
    } else {
        // u8 input at a general rate: the transform-domain kernel converts on load (poly_gen.hip: IN_U8); the direct kernels
        // (exact mode, SFE_RS_ALGO_DIRECT) have no u8 form
        if (r->in_u8 && (r->exact_stream || r->fft_mode < 0)) {
            set_error("rs_process_stream: u8 input at a non-integer step runs the transform-domain kernel only (not the exact mode, not SFE_RS_ALGO_DIRECT)");
            return SFE_ESTATE;
        }
        // Replay the float32 recurrence call by call (blksize samples each), as the reference
        // object would see the stream -- in closed form: each call becomes a few constant-increment
        // runs (timelaw.h) that one workgroup expands on the GPU.  A call's runs are a function of
        // the state it starts in; full-size calls are memoised per start state (Rs::seg_memo).
        sfe_rs_timestate st = r->ts;
        if (r->memo_rate != rate || r->memo_m != r->blksize) {
            r->seg_memo.clear();
            r->seg_refs.clear();
            r->seg_table.clear();
            r->seg_uploaded = 0;
            r->memo_rate = rate;
            r->memo_m = r->blksize;
        }
        constexpr size_t MEMO_MAX_SEGS = (size_t)2 << 20;       // 64 MiB of runs: beyond, calls are planned without the memo
        std::vector<TlSeg> extra;                               // runs of calls that are not memoised (the ragged last one)
        std::vector<size_t> extra_chunks;
        std::vector<SegChunk> chunks;
        chunks.reserve(n_in / (size_t)r->blksize + 1);
        size_t K = 0;
        int max_m = 0;
        int prev_ref = -1;          // plan of the previous (memoised) call of this launch: its `next` link is followed / filled in
        // A reference call that runs out of out_len mid-block leaves its time state BEFORE the next call's first sample
        // (pos < -1; SURVEY.md section 5 -- it happens when the float32 recurrence drifts by more outputs than the
        // ceil(n_in / rate) + 2 a call is given, e.g. blksize 16384 x 3 phases at a step of 3.005).  The direct kernel
        // reproduces that state output for output; the transform-domain kernel assigns outputs to blocks by position and
        // does not take such calls.
        bool exhausted = false;
        for (size_t off = 0; off < n_in; off += (size_t)r->blksize) {
            const int m = (int)((n_in - off) < (size_t)r->blksize ? (n_in - off) : (size_t)r->blksize);
            const int cap = (int)ceilf((float)m / rate) + 2;
            exhausted = exhausted || st.pos < -1;
            SegChunk c;
            c.in_off = (long long)off;
            c.k_first = (long long)K;
            c.m = m;
            if (m == r->blksize && r->seg_table.size() < MEMO_MAX_SEGS) {
                int idx = prev_ref >= 0 ? r->seg_refs[(size_t)prev_ref].next : -1;
                if (idx < 0) {
                    uint32_t mu_bits;
                    memcpy(&mu_bits, &st.mu, 4);
                    const uint64_t key = ((uint64_t)(uint32_t)(st.pos + 1) << 33) | ((uint64_t)mu_bits << 1) | (uint64_t)(st.leftover ? 1 : 0);   // pos >= -1
                    auto it = r->seg_memo.find(key);
                    if (it == r->seg_memo.end()) {
                        Rs::SegPlanRef ref;
                        ref.seg_first = (int)r->seg_table.size();
                        sfe_rs_timestate st2 = st;
                        ref.n_out = time_law_segments(&st2, r->U, m, cap, rate, r->seg_table);
                        ref.n_seg = (int)r->seg_table.size() - ref.seg_first;
                        ref.after = st2;
                        idx = (int)r->seg_refs.size();
                        r->seg_refs.push_back(ref);
                        r->seg_memo.emplace(key, idx);
                    } else {
                        idx = it->second;
                    }
                    if (prev_ref >= 0) r->seg_refs[(size_t)prev_ref].next = idx;
                }
                const Rs::SegPlanRef &ref = r->seg_refs[(size_t)idx];
                st = ref.after;
                c.seg_first = ref.seg_first;
                c.n_seg = ref.n_seg;
                c.n_out = ref.n_out;
                prev_ref = idx;
            } else {
                prev_ref = -1;
                c.seg_first = (int)extra.size();                 // + the table's final size, below
                c.n_out = time_law_segments(&st, r->U, m, cap, rate, extra);
                c.n_seg = (int)extra.size() - c.seg_first;
                extra_chunks.push_back(chunks.size());
            }
            chunks.push_back(c);
            K += (size_t)c.n_out;
            max_m = m > max_m ? m : max_m;
        }
        const size_t n_table = r->seg_table.size(), n_segs = n_table + extra.size();
        for (size_t ci : extra_chunks) chunks[ci].seg_first += (int)n_table;
        auto seg_at = [&](size_t i) -> const TlSeg & { return i < n_table ? r->seg_table[i] : extra[i - n_table]; };
        if (K > out_cap) {
            set_error("rs_process_stream: need room for %zu outputs, got %zu", K, out_cap);
            return SFE_ERANGE;
        }
        PolySegArgs sa;
        memset(&sa, 0, sizeof(sa));
        sa.in = d_in;
        sa.out = d_out;
        sa.hist = r->d_hist[r->cur];
        sa.taps = r->d_taps;
        sa.n_in = (long long)n_in;
        sa.in_stride = (long long)in_stride;
        sa.out_stride = (long long)out_stride;
        sa.hl = r->hl;
        sa.U = r->U;
        sa.plen = r->plen;
        sa.n_chunks = (int)chunks.size();
        sa.max_m = max_m;
        sa.taps_global = 0;
        sa.split = 1;
        sa.tile_cap = 0;
        sa.span_slack = (int)ceilf(rate) + 64;          // one output's step in samples + room for the float32 recurrence's wobble
        // plan tables: grow-only device arrays + pinned staging.  The previous call's UPLOADS may still be reading the
        // staging: wait for them -- the event behind them -- not for the stream: that call's kernel runs on while this
        // call is planned and queued (waiting for the stream here made every call a full host/device round trip).
        // The device arrays themselves are ordered by the stream; a call on ANOTHER stream than the last waits for that one.
        if (r->ev_plan) {
            if (r->plan_stream != s) SFE_HIP(hipStreamSynchronize(r->plan_stream));
            else SFE_HIP(hipEventSynchronize(r->ev_plan));
        } else {
            SFE_HIP(hipEventCreateWithFlags(&r->ev_plan, hipEventDisableTiming));
        }
        if (n_segs > r->segs_cap) {
            if (r->d_segs) (void)hipFree(r->d_segs);
            if (r->h_segs) (void)hipHostFree(r->h_segs);
            r->d_segs = r->h_segs = nullptr;
            r->segs_cap = 0;
            r->seg_uploaded = 0;
            const size_t cap2 = n_segs * 2 + 1024;
            SFE_HIP(hipMalloc(&r->d_segs, cap2 * sizeof(TlSeg)));
            SFE_HIP(hipHostMalloc(&r->h_segs, cap2 * sizeof(TlSeg)));
            r->segs_cap = cap2;
        }
        if (chunks.size() > r->chunks_cap) {
            if (r->d_chunks) (void)hipFree(r->d_chunks);
            if (r->h_chunks) (void)hipHostFree(r->h_chunks);
            r->d_chunks = r->h_chunks = nullptr;
            r->chunks_cap = 0;
            const size_t cap2 = chunks.size() * 2 + 64;
            SFE_HIP(hipMalloc(&r->d_chunks, cap2 * sizeof(SegChunk)));
            SFE_HIP(hipHostMalloc(&r->h_chunks, cap2 * sizeof(SegChunk)));
            r->chunks_cap = cap2;
        }
        // upload what the device does not hold yet: the table's new tail, then this call's own runs behind it
        {
            TlSeg *hs = static_cast<TlSeg *>(r->h_segs);
            const size_t from = r->seg_uploaded < n_table ? r->seg_uploaded : n_table;
            if (n_table > from) memcpy(hs + from, r->seg_table.data() + from, (n_table - from) * sizeof(TlSeg));
            if (!extra.empty()) memcpy(hs + n_table, extra.data(), extra.size() * sizeof(TlSeg));
            if (n_segs > from)
                SFE_HIP(hipMemcpyAsync(static_cast<TlSeg *>(r->d_segs) + from, hs + from, (n_segs - from) * sizeof(TlSeg),
                                       hipMemcpyHostToDevice, s));
            r->seg_uploaded = n_table;
        }
        memcpy(r->h_chunks, chunks.data(), chunks.size() * sizeof(SegChunk));
        SFE_HIP(hipMemcpyAsync(r->d_chunks, r->h_chunks, chunks.size() * sizeof(SegChunk), hipMemcpyHostToDevice, s));
        SFE_HIP(hipEventRecord(r->ev_plan, s));
        r->plan_stream = s;
        sa.segs = r->d_segs;
        sa.chunks = static_cast<const SegChunk *>(r->d_chunks);
        // Bulk calls (complex or real, float32 or u8) in fused arithmetic, at any rate: the transform-domain kernel (poly_gen.hip) --
        // all U phases of every input by one forward and U inverse 4096-point transforms, outputs picked and blended from
        // LDS by the same runs.  sfe_dsp_rs_set_algo(SFE_RS_ALGO_DIRECT) and the exact mode keep poly_seg_kernel.
        rc = SFE_ESTATE;
        // (1 + U) transforms per block against two dot products of plen taps per output: the transform kernel while U <= 2 + plen / 16 (measured at
        // 2^26 samples and rate 1.77, profiles/r05/speed_sweep.txt: it takes 0.43 + 0.11 U ms, the direct kernel 0.69 + 0.0072 plen; until round 5
        // the rule was "12 taps per phase or more" and 16 phases of 127 taps ran 2.2 ms where the direct kernel takes 1.6)
        const bool gen_pays = r->plen >= 12 && r->U <= 2 + r->plen / 16;
        if (!exhausted && !r->exact_stream && r->fft_mode >= 0 &&
            (r->fft_mode > 0 || r->in_u8 || (n_in >= ((size_t)1 << 16) && gen_pays))) {
            if (!r->gen_tried) {
                // spectra of the U phase filters (taps[i U + j], i < plen; one zero behind so that the overlap the FIR
                // planner picks covers plen samples, not plen - 1) through fir_build_tables: one "channel" per phase
                r->gen_tried = true;
                std::vector<float> rows((size_t)r->U * (r->plen + 1), 0.0f);
                for (int j = 0; j < r->U; j++)
                    for (int i = 0; i < r->plen; i++) rows[(size_t)j * (r->plen + 1) + i] = r->h_taps_pm[(size_t)j * r->plen + i];
                sfe_fir_t gh = nullptr;
                if (fir_create_impl(rows.data(), r->plen + 1, 0, 1, r->U, 0, r->device, 1, &gh) == SFE_OK) {
                    r->gen_tables = static_cast<Fir *>(gh);
                    if (r->gen_tables->parts != 1) {
                        fir_free(r->gen_tables);
                        r->gen_tables = nullptr;
                    }
                }
            }
            if (r->gen_tables) {
                PolyGenArgs ga;
                memset(&ga, 0, sizeof(ga));
                ga.in = d_in;
                ga.out = d_out;
                ga.hist = r->d_hist[r->cur];
                ga.hs = r->gen_tables->d_hs;
                ga.tw1 = r->gen_tables->d_tw1;
                ga.tw2 = r->gen_tables->d_tw2;
                ga.segs = r->d_segs;
                ga.chunks = static_cast<const SegChunk *>(r->d_chunks);
                ga.n_in = (long long)n_in;
                ga.in_stride = (long long)in_stride;
                ga.out_stride = (long long)out_stride;
                ga.hl = r->hl;
                ga.U = r->U;
                ga.plen = r->plen;
                ga.ovl = (r->plen + 15) & ~15;              // >= plen; the spectra do not depend on it
                ga.blksize = r->blksize;
                ga.n_chunks = (int)chunks.size();
                ga.real = r->data_complex ? 0 : 1;
                ga.in_u8 = r->in_u8 ? 1 : 0;
                int max_runs = 0;              // over any two (a real stream's pairs of blocks: three) consecutive calls
                for (size_t i = 0; i < chunks.size(); i++) {
                    int sum = chunks[i].n_seg + (i + 1 < chunks.size() ? chunks[i + 1].n_seg : 0);
                    if (ga.real && i + 2 < chunks.size()) sum += chunks[i + 2].n_seg;
                    max_runs = sum > max_runs ? sum : max_runs;
                }
                rc = launch_poly_gen(ga, max_runs, stepf, r->n_channels, s);
            }
        }
        if (rc == SFE_ESTATE && r->in_u8) {
            r->u8_refused = true;              // (the public entry converts the bytes and comes back with float32)
            set_error("rs_process_stream: u8 input at a non-integer step: this call is outside what the transform-domain kernel takes "
                      "(more than 2032 taps per phase, more than 32 phases, blksize below a block's advance, or a call in the "
                      "reference's out_len-exhausted state)");
            return SFE_ESTATE;
        }
        if (rc == SFE_ESTATE) rc = launch_poly_seg(sa, r->data_complex, r->exact_stream, r->n_channels, s);
        if (rc == SFE_ESTATE) {
            // a call's input does not fit an LDS tile (huge blksize): expand on the host and use
            // the per-output schedule kernel
            std::vector<long long> pos(K);
            std::vector<float> mu(K);
            for (const SegChunk &c : chunks)
                for (int i = 0; i < c.n_seg; i++) {
                    const TlSeg &g = seg_at((size_t)c.seg_first + i);
                    for (int q = 0; q < g.count; q++) {
                        const double t = g.t0 + (double)q * (double)g.d, fl = floor(t);
                        pos[(size_t)c.k_first + g.k0 + q] = c.in_off * r->U + (long long)fl;
                        mu[(size_t)c.k_first + g.k0 + q] = (float)(t - fl);
                    }
                }
            rc = rs_ensure_sched(r, K + 1);
            if (rc != SFE_OK) return rc;
            SFE_HIP(hipStreamSynchronize(s));        // the schedule staging may still be read by the previous call's uploads
            memcpy(r->h_pos, pos.data(), K * sizeof(long long));
            memcpy(r->h_mu, mu.data(), K * sizeof(float));
            SFE_HIP(hipMemcpyAsync(r->d_pos, r->h_pos, K * sizeof(long long), hipMemcpyHostToDevice, s));
            SFE_HIP(hipMemcpyAsync(r->d_mu, r->h_mu, K * sizeof(float), hipMemcpyHostToDevice, s));
            a.n_out = (long long)K;
            a.sched_pos = r->d_pos;
            a.sched_mu = r->d_mu;
            rc = launch_poly_sched(a, r->data_complex, r->exact_stream, r->n_channels, s);
        }
        if (rc != SFE_OK) return rc;
        r->ts = st;
        *n_out = K;
    }
    if (capturing) {        // in place behind the main launch: with n_in >= hl the kernel reads `in` only; the time state did not move
        r->captured = true;
        return launch_history_update(d_in, (long long)n_in, (long long)in_stride, r->d_hist[r->cur],
                                     r->d_hist[r->cur], r->hl, r->data_complex ? 2 : 1, r->n_channels, s, r->in_u8);
    }
    if (!hist_fused) {
        rc = launch_history_update(d_in, (long long)n_in, (long long)in_stride, r->d_hist[r->cur],
                                   r->d_hist[r->cur ^ 1], r->hl, r->data_complex ? 2 : 1, r->n_channels, s, r->in_u8);
        if (rc != SFE_OK) return rc;
    }
    // a handle one of whose calls sits in a hipGraph keeps its history in d_hist[cur], the buffer the graph
    // names: eager calls copy the new history back instead of flipping (fir_carry_state has the reasoning)
    if (r->captured)
        SFE_HIP(hipMemcpyAsync(r->d_hist[r->cur], r->d_hist[r->cur ^ 1], (size_t)r->n_channels * r->hl * r->esz(),
                               hipMemcpyDeviceToDevice, s));
    else
        r->cur ^= 1;
    return SFE_OK;
}


// ---- cutting one stream into spans -------------------------
int sfe_dsp_rs_load_history(sfe_rs_t h, const void *d_prev, size_t n_prev, size_t stride, sfe_stream_t stream)
{
    Rs *r = as_rs(h);
    if (!r || (n_prev && !d_prev)) {
        set_error("rs_load_history: null handle or buffer");
        return SFE_EINVAL;
    }
    if (r->n_channels > 1 && stride < n_prev) {
        set_error("rs_load_history: channel stride smaller than n_prev");
        return SFE_EINVAL;
    }
    if (reinterpret_cast<uintptr_t>(d_prev) & (size_t)(r->esz() - 1)) {
        set_error("rs_load_history: buffer must be aligned to its element");
        return SFE_EINVAL;
    }
    SFE_ON_DEVICE(r->device);
    hipStream_t s = (hipStream_t)stream;
    const size_t hb = (size_t)r->n_channels * r->hl * r->esz();
    SFE_HIP(hipMemsetAsync(r->d_hist[r->cur], 0, hb, s));
    if (n_prev) {
        int rc = launch_history_update(d_prev, (long long)n_prev, (long long)stride, r->d_hist[r->cur], r->d_hist[r->cur ^ 1],
                                       r->hl, r->data_complex ? 2 : 1, r->n_channels, s, 0);
        if (rc != SFE_OK) return rc;
        if (r->captured) SFE_HIP(hipMemcpyAsync(r->d_hist[r->cur], r->d_hist[r->cur ^ 1], hb, hipMemcpyDeviceToDevice, s));
        else r->cur ^= 1;
    }
    return SFE_OK;
}

int sfe_dsp_rs_get_state(sfe_rs_t h, sfe_rs_timestate *state)
{
    Rs *r = as_rs(h);
    if (!r || !state) return SFE_EINVAL;
    *state = r->ts;
    return SFE_OK;
}

int sfe_dsp_rs_set_state(sfe_rs_t h, const sfe_rs_timestate *state)
{
    Rs *r = as_rs(h);
    if (!r || !state || state->pos < -1 || !(state->mu >= 0.0f && state->mu < 1.0f)) {
        set_error("rs_set_state: need pos >= -1 and 0 <= mu < 1");
        return SFE_EINVAL;
    }
    r->ts = *state;
    r->ts.leftover = state->leftover ? 1 : 0;
    return SFE_OK;
}

// The time state a reference object has after consuming `first_sample` samples of a stream from a
// fresh start, in closed form -- only when fl(rate*upsample) is integer-valued (then mu == 0 and the
// float32 recurrence resample.cxx:129-150 is exact): output k sits at upsampled position k*S, the
// object's m_pos is the first such position at or after first_sample*U - 1, relative to it, and a
// position of exactly first_sample*U - 1 is the pending "leftover" output (resample.cxx:141-145).
int sfe_dsp_rs_plan_seek(sfe_rs_timestate *state, int upsample, uint64_t first_sample, float rate)
{
    if (!state || upsample < 1) return SFE_EINVAL;
    const float stepf = rate * (float)upsample;
    if (!(stepf >= 1.0f && stepf == floorf(stepf) && stepf < 1.0e6f)) {
        set_error("rs_seek: fl(rate*upsample) = %g is not integer-valued: the float32 time recurrence has no closed form "
                  "(carry the state with sfe_dsp_rs_get_state / set_state instead)", (double)stepf);
        return SFE_ESTATE;
    }
    const unsigned long long S = (unsigned long long)stepf, U = (unsigned long long)upsample;
    if (first_sample == 0) {
        *state = {0, 0.0f, 0};
        return SFE_OK;
    }
    const unsigned long long edge = first_sample * U - 1;        // last upsampled position of the part before the cut
    const unsigned long long k = (edge + S - 1) / S;             // first output at or after it
    const long long rel = (long long)(k * S) - (long long)(first_sample * U);
    state->leftover = rel == -1 ? 1 : 0;
    state->pos = (int32_t)rel;
    state->mu = 0.0f;
    return SFE_OK;
}

int sfe_dsp_rs_seek(sfe_rs_t h, uint64_t first_sample, float rate)
{
    Rs *r = as_rs(h);
    if (!r) return SFE_EINVAL;
    return sfe_dsp_rs_plan_seek(&r->ts, r->U, first_sample, rate);
}

int sfe_dsp_rs_set_input_format(sfe_rs_t h, int fmt)
{
    Rs *r = as_rs(h);
    if (!r || (fmt != SFE_FMT_F32 && fmt != SFE_FMT_U8)) return SFE_EINVAL;
    if (r->piped && (fmt == SFE_FMT_U8) != (r->in_u8 != 0)) {
        set_error("rs_set_input_format: a pipe over this handle has frozen its item format (destroy the pipe first)");
        return SFE_ESTATE;
    }
    r->in_u8 = fmt == SFE_FMT_U8;
    return SFE_OK;
}

int sfe_dsp_rs_set_algo(sfe_rs_t h, int algo)
{
    Rs *r = as_rs(h);
    if (!r || algo < SFE_RS_ALGO_AUTO || algo > SFE_RS_ALGO_MFMA) return SFE_EINVAL;
    r->fft_mode = algo == SFE_RS_ALGO_FFT ? 1 : (algo == SFE_RS_ALGO_AUTO ? 0 : -1);
    r->use_mfma = algo == SFE_RS_ALGO_MFMA;
    // plans are cached per (step, pos0) together with the choice that made them; clearing frees
    // device tables a launch in flight may still read, so wait for the handle's device first, in
    // ITS context (the current device of a multi-GPU caller may be another one)
    SFE_ON_DEVICE(r->device);
    SFE_HIP(hipDeviceSynchronize());
    r->fft_plans.clear();
    return SFE_OK;
}

int sfe_dsp_rs_set_exact(sfe_rs_t h, int exact)
{
    Rs *r = as_rs(h);
    if (!r) return SFE_EINVAL;
    r->exact_stream = exact ? 1 : 0;
    return SFE_OK;
}

int sfe_dsp_rs_reset(sfe_rs_t h)
{
    Rs *r = as_rs(h);
    if (!r) return SFE_EINVAL;
    SFE_ON_DEVICE(r->device);
    SFE_HIP(hipDeviceSynchronize());
    const size_t hb = (size_t)r->n_channels * r->hl * r->esz();
    for (int i = 0; i < 2; i++) SFE_HIP(hipMemset(r->d_hist[i], 0, hb));
    if (r->d_ticket) SFE_HIP(hipMemset(r->d_ticket, 0, POLY_TICKET_GROUPS * 128));
    r->ts.pos = 0;
    r->ts.mu = 0.0f;
    r->ts.leftover = 0;
    return SFE_OK;
}

int sfe_dsp_rs_destroy(sfe_rs_t h)
{
    Rs *r = as_rs(h);
    if (!r) return SFE_OK;
    if (r->piped) {
        set_error("rs_destroy: a pipe still borrows this handle (sfe_dsp_pipe_destroy first)");
        return SFE_ESTATE;
    }
    DeviceGuard g(r->device);
    (void)hipDeviceSynchronize();
    rs_free(r);
    return SFE_OK;
}

}  // extern "C"
