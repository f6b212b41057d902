// api_fir.hip -- the FIR handle behind sfe_fir_t (blkconv, libdsp/blkconv.cxx:34-122): spectrum tables and tap partitions,
// the data-movement variants, carried state, and the sfe_dsp_fir_* entry points.  Host code only.
#include "host.h"

namespace sfe {

// ------------------------------------------------------------------------------ FIR

Fir *as_fir(void *h)
{
    Fir *f = static_cast<Fir *>(h);
    if (f && f->magic != 0x46495231u) {
        set_error("not a live FIR handle");
        return nullptr;
    }
    return f;
}

void fir_free(Fir *f)
{
    if (!f) return;
    f->magic = 0;
    DeviceGuard g(f->device);
    if (f->d_hs) (void)hipFree(f->d_hs);
    if (f->d_tw1) (void)hipFree(f->d_tw1);
    if (f->d_tw2) (void)hipFree(f->d_tw2);
    if (f->d_ticket) (void)hipFree(f->d_ticket);
    if (f->d_taps) (void)hipFree(f->d_taps);
    f->plans.clear();
    for (int i = 0; i < 2; i++)
        if (f->d_hist[i]) (void)hipFree(f->d_hist[i]);
    if (f->h_buf) (void)hipHostFree(f->h_buf);
    if (f->d_blk_in) (void)hipFree(f->d_blk_in);
    if (f->d_blk_out) (void)hipFree(f->d_blk_out);
    if (f->h_blk_out) (void)hipHostFree(f->h_blk_out);
    if (f->h_stage_out) (void)hipHostFree(f->h_stage_out);
    if (f->h_stage) (void)hipHostFree(f->h_stage);
    if (f->d_st_in) (void)hipFree(f->d_st_in);
    if (f->d_st_out) (void)hipFree(f->d_st_out);
    if (f->stream) (void)hipStreamDestroy(f->stream);
    delete f;
}

static int fir_build_tables(Fir *f, const float *taps)
{
    const int N = FFT_N;
    // spectrum of each zero-padded tap partition in double precision, scaled by 1/N (blkconv.cxx:50
    // folds the same 1/fft_len into its multiply), permuted to the kernel's F3 thread order:
    // thread t (k1 = t&15, k2 = t>>4), register k0 -> bin k2 + 16 k1 + 256 k0.
    // Partition p holds taps [p*ovl, (p+1)*ovl) (a single partition: all n_taps <= ovl+1 of them).
    std::vector<double> c(N / 2), sn(N / 2);
    for (int m = 0; m < N / 2; m++) {
        c[m] = cos(-2.0 * M_PI * m / N);
        sn[m] = sin(-2.0 * M_PI * m / N);
    }
    // in-place radix-2 decimation-in-time FFT in double precision (forward sign): table construction
    // only, so that a filter of many partitions does not cost N * n_taps trigonometric multiplies
    auto fft = [&](std::vector<double> &re, std::vector<double> &im) {
        for (int i = 1, j = 0; i < N; i++) {
            int bit = N >> 1;
            for (; j & bit; bit >>= 1) j ^= bit;
            j ^= bit;
            if (i < j) {
                std::swap(re[i], re[j]);
                std::swap(im[i], im[j]);
            }
        }
        for (int len = 2; len <= N; len <<= 1) {
            const int half = len >> 1, step = N / len;
            for (int base = 0; base < N; base += len)
                for (int k = 0; k < half; k++) {
                    const double wr = c[k * step], wi = sn[k * step];
                    const int a = base + k, b = a + half;
                    const double xr = re[b] * wr - im[b] * wi, xi = re[b] * wi + im[b] * wr;
                    re[b] = re[a] - xr;
                    im[b] = im[a] - xi;
                    re[a] += xr;
                    im[a] += xi;
                }
        }
    };
    // one set of spectra per tap vector: one for all channels, or (per_channel) channel by channel, [set][partition][k0][t]
    const int n_sets = f->per_channel ? f->n_channels : 1;
    std::vector<v2f> hs((size_t)n_sets * f->parts * 16 * 256), tw1(7 * 256), tw2(7 * 16);
    std::vector<double> hr(N), hi(N);
    for (int set = 0; set < n_sets; set++) {
    const float *tp = taps + (size_t)set * f->n_taps * (f->taps_complex ? 2 : 1);
    for (int p = 0; p < f->parts; p++) {
        const int first = f->parts == 1 ? 0 : p * f->ovl;
        const int count = f->parts == 1 ? f->n_taps : (f->n_taps - first < f->ovl ? f->n_taps - first : f->ovl);
        for (int n = 0; n < N; n++) {
            hr[n] = n < count ? (f->taps_complex ? tp[2 * (first + n)] : tp[first + n]) / (double)N : 0.0;
            hi[n] = n < count && f->taps_complex ? tp[2 * (first + n) + 1] / (double)N : 0.0;
        }
        fft(hr, hi);
        for (int t = 0; t < 256; t++)
            for (int k0 = 0; k0 < 16; k0++) {
                const int bin = (t >> 4) + 16 * (t & 15) + 256 * k0;
                hs[(((size_t)set * f->parts + p) * 16 + k0) * 256 + t] = (v2f){(float)hr[bin], (float)hi[bin]};
            }
    }
    }
    // twiddle bases: row k (1..3) = W^(e k), row k+3 = W^(4 e k); the kernel forms
    // W^(e (4a+b)) as row[a+3] * row[b]
    for (int k = 1; k < 4; k++)
        for (int t = 0; t < 256; t++) {
            const double a = -2.0 * M_PI * (double)(t * k) / 4096.0;
            tw1[k * 256 + t] = (v2f){(float)cos(a), (float)sin(a)};
            tw1[(k + 3) * 256 + t] = (v2f){(float)cos(4.0 * a), (float)sin(4.0 * a)};
        }
    for (int k = 1; k < 4; k++)
        for (int n0 = 0; n0 < 16; n0++) {
            const double a = -2.0 * M_PI * (double)(n0 * k) / 256.0;
            tw2[k * 16 + n0] = (v2f){(float)cos(a), (float)sin(a)};
            tw2[(k + 3) * 16 + n0] = (v2f){(float)cos(4.0 * a), (float)sin(4.0 * a)};
        }
    SFE_HIP(hipMalloc(&f->d_hs, hs.size() * sizeof(v2f)));
    SFE_HIP(hipMalloc(&f->d_tw1, tw1.size() * sizeof(v2f)));
    SFE_HIP(hipMalloc(&f->d_tw2, tw2.size() * sizeof(v2f)));
    SFE_HIP(hipMemcpy(f->d_hs, hs.data(), hs.size() * sizeof(v2f), hipMemcpyHostToDevice));
    SFE_HIP(hipMemcpy(f->d_tw1, tw1.data(), tw1.size() * sizeof(v2f), hipMemcpyHostToDevice));
    SFE_HIP(hipMemcpy(f->d_tw2, tw2.data(), tw2.size() * sizeof(v2f), hipMemcpyHostToDevice));
    SFE_HIP(hipMalloc(&f->d_ticket, FIR_TICKET_GROUPS_MAX * 128));
    SFE_HIP(hipMemset(f->d_ticket, 0, FIR_TICKET_GROUPS_MAX * 128));
    return SFE_OK;
}

// How a tap count is cut for the 4096-point kernel.  One launch with overlap hl costs ~1/(4096-hl) per
// output sample; P launches over partitions of `ovl` taps cost P/(4096-ovl) plus the read-modify-
// write of the output for every launch after the first (8 more bytes per sample: ~1/4 of a launch's
// traffic).  E.g. 3841 taps: one launch advances 256 samples per transform (16x the 256-tap work);
// two partitions of 2048 advance 2048 (2.25x).  Returns false beyond FIR_MAX_PARTS partitions.
constexpr int FIR_MAX_PARTS = 1024;       // ~3.9 million taps; 32 KiB of spectrum per partition
static bool fir_choose_partition(int n_taps, int *ovl, int *parts)
{
    const int need = n_taps > 1 ? n_taps - 1 : 1;
    double best = 1e300;
    *parts = 0;
    const int hl1 = ((need + 255) / 256) * 256;
    if (hl1 < FFT_N) {
        best = 1.0 / (FFT_N - hl1);
        *ovl = hl1;
        *parts = 1;
    }
    for (int o = 256; o < FFT_N; o += 256) {
        const int P = (n_taps + o - 1) / o;
        if (P < 2 || P > FIR_MAX_PARTS) continue;
        const double cost = (P + 0.25 * (P - 1)) / (FFT_N - o);
        if (cost < best) {
            best = cost;
            *ovl = o;
            *parts = P;
        }
    }
    return *parts > 0;
}

// A stream that is being captured into a hipGraph: the launches recorded now will be REPLAYED with the same
// arguments, so nothing of the stream's carried state may live on the host between a captured call and its
// replays.  A captured bulk call therefore (a) updates the history IN PLACE with the separate carry-over
// kernel behind the main launch (no double-buffer parity to flip on the host) and (b) is accepted only when
// its arguments do not depend on where in the stream it sits: n >= the history length, and for the
// resamplers an integer-valued step with n*U a multiple of it, so that every call starts in the time state
// the captured one started in.  Replaying the graph then processes the NEXT n samples found in d_in, exactly
// as the next eager call would (tests/test_gpu_graph.py).  VERDICT r2 item 8.
bool stream_is_capturing(hipStream_t s)
{
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return cap != hipStreamCaptureStatusNone;
}

// ---- which data-movement variant of the cf32 kernel (common.h: FIR_VAR_*) ------------------------
// The three variants compute the same bits and differ by a few percent in time, with a sign that
// depends on the box (profiles/r02/fir_walk_vs_tickets.txt against DESIGN.md 4.1's earlier tables:
// LDS-DMA from -5.7 % to +3.7 % against register loads).  Round 4: NOTHING IS
// MEASURED ON THE CALL PATH.  A stream call runs what sfe_dsp_fir_set_variant fixed, else what an
// earlier sfe_dsp_fir_calibrate call chose for this (device, channels, size class, overlap,
// per-channel taps), else register loads.  sfe_dsp_fir_calibrate is the measurement, made when the
// caller asks for it, synchronously and outside the stream: every variant over the caller's buffers
// (same output each time; the carried state and the stream position are not touched),
// FIR_CAL_ROUNDS interleaved rounds behind FIR_CAL_WARM_MS of launches, HIP events on the caller's
// stream; register loads unless another variant's median is more than 1 % ahead.
constexpr int FIR_CAL_ROUNDS = 9;                         // rounds that count: the LAST nine
constexpr float FIR_CAL_MARGIN = 0.99f;                   // another variant displaces register loads only by more than 1 %
constexpr int FIR_CAL_MAX_ROUNDS = 24;                    // ... of at most this many, and of at least FIR_CAL_WARM_MS of launches:
constexpr float FIR_CAL_WARM_MS = 80.0f;                  // the chip's first ~100 ms of work after idling run 5-6 % slow
struct FirVarKey {
    int device, n_channels, size_class, ovl, per_channel;
    bool operator<(const FirVarKey &o) const
    {
        return std::tie(device, n_channels, size_class, ovl, per_channel) < std::tie(o.device, o.n_channels, o.size_class, o.ovl, o.per_channel);
    }
};
static std::mutex g_fir_var_mutex;
static std::map<FirVarKey, int> g_fir_var_cache;

static FirVarKey fir_var_key(const Fir *f, const FirFftArgs &a)
{
    int sc = 0;
    for (unsigned long long v = (unsigned long long)a.nblk * f->n_channels; v > 1; v >>= 1) sc++;
    return FirVarKey{f->device, f->n_channels, sc, f->ovl, f->per_channel};
}

static bool fir_has_variants(const Fir *f, const FirFftArgs &a)
{
    return f->parts == 1 && fir_fft_has_variants(a, f->data_complex, f->out_complex, f->in_u8, f->out_tx10, f->n_channels, 0);
}

// what a stream call runs: a map look-up, no device work
static int fir_pick_variant(Fir *f, const FirFftArgs &a)
{
    if (f->variant != FIR_VAR_AUTO) return f->variant;     // sfe_dsp_fir_set_variant
    if (!fir_has_variants(f, a)) return FIR_VAR_REG;
    std::lock_guard<std::mutex> lk(g_fir_var_mutex);
    auto it = g_fir_var_cache.find(fir_var_key(f, a));
    return it != g_fir_var_cache.end() ? it->second : FIR_VAR_REG;
}

// the measurement (sfe_dsp_fir_calibrate): `a` describes the call, a.hist_out == nullptr
static int fir_calibrate(Fir *f, FirFftArgs &a, hipStream_t s, int *chosen)
{
    *chosen = FIR_VAR_REG;
    if (!fir_has_variants(f, a)) return SFE_OK;             // one variant: nothing to choose
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        if (e0) (void)hipEventDestroy(e0);
        return hip_fail(hipGetLastError(), "fir_calibrate: hipEventCreate");
    }
    const int nvar = f->per_channel ? 2 : FIR_VAR_COUNT;      // per-channel taps: no wave-private instantiation
    float t[FIR_VAR_COUNT][FIR_CAL_ROUNDS];
    int rc = SFE_OK;
    // interleaved rounds; a measurement made on a chip that has just come out of idle ranks the variants by
    // how they run at a clock the stream will never see again, so rounds go on until FIR_CAL_WARM_MS of
    // launches have run (and at least FIR_CAL_ROUNDS rounds) and only the last FIR_CAL_ROUNDS count
    float spent = 0.0f;
    for (int r = 0; r < FIR_CAL_MAX_ROUNDS && rc == SFE_OK && (r < FIR_CAL_ROUNDS || spent < FIR_CAL_WARM_MS); r++)
        for (int v = 0; v < nvar && rc == SFE_OK; v++) {
            a.variant = v;
            hipError_t e = hipEventRecord(e0, s);
            rc = launch_fir_fft(a, f->data_complex, f->out_complex, f->in_u8, f->out_tx10, f->n_channels, s, 0);
            if (rc != SFE_OK) break;
            if (e == hipSuccess) e = hipEventRecord(e1, s);
            if (e == hipSuccess) e = hipEventSynchronize(e1);
            float ms = 0.0f;
            if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
            if (e != hipSuccess) rc = hip_fail(e, "fir variant calibration");
            else {
                t[v][r % FIR_CAL_ROUNDS] = ms;         // a ring: the last FIR_CAL_ROUNDS rounds survive
                spent += ms;
            }
        }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc != SFE_OK) return rc;
    // Register loads are the reference point: with the straight-line store block they are the fastest or within
    // 0.5 % of it on every box of profiles/r03/fir_variants_box*.txt, where the three medians of one measurement lie
    // within 0.7 % of each other -- inside the noise of nine rounds.  Another variant is taken only when it is ahead
    // by more than 1 % (boxes on which LDS-DMA led by 4-6 % exist: DESIGN.md 4.1), the better of the two if both are.
    int best = FIR_VAR_REG;
    float best_ms = 0.0f;
    for (int v = 0; v < nvar; v++) {
        std::sort(t[v], t[v] + FIR_CAL_ROUNDS);
        const float med = t[v][FIR_CAL_ROUNDS / 2];
        f->cal_ms[v] = med;
        if (v == 0) best_ms = med * FIR_CAL_MARGIN;
        else if (med < best_ms) {
            best_ms = med;
            best = v;
        }
    }
    f->cal_runs++;
    {
        std::lock_guard<std::mutex> lk(g_fir_var_mutex);
        g_fir_var_cache[fir_var_key(f, a)] = best;
    }
    *chosen = best;
    return SFE_OK;
}

// the launch description of one bulk call over the transform kernel (partition 0)
static void fir_fill_args(const Fir *f, FirFftArgs &a, const void *d_in, void *d_out, size_t n, size_t in_stride,
                          size_t out_stride)
{
    a.in = d_in;
    a.out = d_out;
    a.hist = f->d_hist[f->cur];
    a.tw1 = f->d_tw1;
    a.tw2 = f->d_tw2;
    a.n = (long long)n;
    a.in_stride = (long long)in_stride;
    a.out_stride = (long long)out_stride;
    a.hl = f->ovl;
    a.advance = FFT_N - f->ovl;
    a.hist_len = f->hl;
    a.nblk = ((long long)n + a.advance - 1) / a.advance;
    a.ticket = f->d_ticket;
    a.total = 0;
    a.tgroups = 0;
    a.hs_stride = f->per_channel ? (long long)f->parts * 16 * 256 : 0;
    a.variant = FIR_VAR_AUTO;
    a.hs = f->d_hs;
    a.shift = 0;
    a.hist_out = nullptr;
}

// Behind a call's launches: the state the NEXT call starts from.  `fused`: the main launch already wrote
// it into d_hist[cur ^ 1].  A handle one of whose calls sits in a hipGraph keeps its state in d_hist[cur]
// for good -- the graph names that buffer -- so an eager call on such a handle copies the new state back
// instead of flipping (a replay after an eager call used to read the stale buffer).
static int fir_carry_state(Fir *f, const void *d_in, size_t n, size_t in_stride, bool fused, bool capturing, hipStream_t s)
{
    const int width = f->data_complex ? 2 : 1;
    if (capturing) {
        // in place, behind everything that read the old history: with n >= hl the kernel reads `in` only
        f->captured = true;
        return launch_history_update(d_in, (long long)n, (long long)in_stride, f->d_hist[f->cur], f->d_hist[f->cur], f->hl,
                                     width, f->n_channels, s, f->in_u8);
    }
    if (!fused) {
        int rc = launch_history_update(d_in, (long long)n, (long long)in_stride, f->d_hist[f->cur], f->d_hist[f->cur ^ 1],
                                       f->hl, width, f->n_channels, s, f->in_u8);
        if (rc != SFE_OK) return rc;
    }
    if (f->captured) SFE_HIP(hipMemcpyAsync(f->d_hist[f->cur], f->d_hist[f->cur ^ 1], f->hist_bytes(), hipMemcpyDeviceToDevice, s));
    else f->cur ^= 1;
    return SFE_OK;
}

int fir_run(Fir *f, const void *d_in, void *d_out, size_t n, size_t in_stride,
                   size_t out_stride, hipStream_t s)
{
    if (n == 0) return SFE_OK;
    f->started = true;
    int algo = f->algo;
    if (algo == SFE_FIR_ALGO_AUTO) algo = f->fft_ok ? SFE_FIR_ALGO_FFT : SFE_FIR_ALGO_DIRECT;
    int rc;
    bool hist_fused = false;
    const bool capturing = stream_is_capturing(s);
    if (capturing && n < (size_t)f->hl) {
        set_error("fir_process_stream: a call captured into a hipGraph must bring at least the history length (%d samples): "
                  "shorter calls carry state the replay cannot see", f->hl);
        return SFE_ESTATE;
    }
    if (algo == SFE_FIR_ALGO_FFT) {
        if (!f->fft_ok) {
            set_error("fir: %d taps exceed %d partitions of the 4096-point kernel", f->n_taps, FIR_MAX_PARTS);
            return SFE_EINVAL;
        }
        FirFftArgs a;
        fir_fill_args(f, a, d_in, d_out, n, in_stride, out_stride);
        hist_fused = n >= (size_t)f->hl && !capturing;   // else the old history still contributes (captured: in place, below)
        a.variant = fir_pick_variant(f, a);
        f->last_variant = a.variant;
        rc = SFE_OK;
        // one launch per tap partition: partition p filters the stream delayed by p*ovl samples and
        // (p > 0) adds to what the earlier ones wrote
        for (int p = 0; p < f->parts && rc == SFE_OK; p++) {
            a.hs = f->d_hs + (size_t)p * 16 * 256;
            a.shift = p * f->ovl;
            a.hist_out = (p == 0 && hist_fused) ? f->d_hist[f->cur ^ 1] : nullptr;
            rc = launch_fir_fft(a, f->data_complex, f->out_complex, f->in_u8, f->out_tx10, f->n_channels, s, p > 0);
        }
    } else {
        if (f->taps_complex || f->in_u8 || f->out_tx10 || f->per_channel) {
            set_error("fir: the direct kernel takes one set of real taps and float input/output; use SFE_FIR_ALGO_FFT");
            return SFE_EINVAL;
        }
        const PolyTiledPlan *pl = get_tiled_plan(f->plans, f->h_taps, 1, f->n_taps, 1, 0, &rc);
        if (rc != SFE_OK) return rc;
        if (pl) {
            PolyTiledArgs ta;
            ta.in = d_in;
            ta.out = d_out;
            ta.hist = f->d_hist[f->cur];
            ta.G = pl->d_G;
            ta.Gt = pl->d_Gt;
            ta.n_in = (long long)n;
            ta.in_stride = (long long)in_stride;
            ta.out_stride = (long long)out_stride;
            ta.n_out = (long long)n;
            ta.hl = f->hl;
            ta.Lp = pl->Lp;
            ta.e_max = pl->e_max;
            rc = launch_poly_tiled(*pl, ta, f->data_complex, 0, 0, f->n_channels, s);
        } else {
            PolyArgs a;
            memset(&a, 0, sizeof(a));
            a.in = d_in;
            a.out = d_out;
            a.hist = f->d_hist[f->cur];
            a.taps = f->d_taps;
            a.n_in = (long long)n;
            a.in_stride = (long long)in_stride;
            a.out_stride = (long long)out_stride;
            a.hl = f->hl;
            a.U = 1;
            a.plen = f->n_taps;
            a.pos0 = 0;
            a.step = 1;
            a.n_out = (long long)n;
            rc = launch_poly_int(a, f->data_complex, 0, 0, f->n_channels, s);
        }
    }
    if (rc != SFE_OK) return rc;
    return fir_carry_state(f, d_in, n, in_stride, hist_fused, capturing, s);
}

int fir_create_impl(const float *taps, int n_taps, int taps_complex, int data_complex,
                           int n_channels, int block_hint, int device, int per_channel, sfe_fir_t *out)
{
    if (!out) return SFE_EINVAL;
    *out = nullptr;
    if (!taps || n_taps < 1 || n_channels < 1) {
        set_error("fir_create: need taps, n_taps >= 1, n_channels >= 1");
        return SFE_EINVAL;
    }
    if (block_hint != 0 && block_hint + 1 - n_taps < 1) {
        set_error("fir_create: fft_len %d leaves no block for %d taps (blkconv.cxx:47)", block_hint, n_taps);
        return SFE_EINVAL;
    }
    int prev_dev = -1;
    (void)hipGetDevice(&prev_dev);
    int rc = use_device(device);
    if (rc != SFE_OK) return rc;
    struct Restore { int d; ~Restore() { if (d >= 0) (void)hipSetDevice(d); } } restore__{prev_dev};
    Fir *f = new (std::nothrow) Fir;
    if (!f) return SFE_ENOMEM;
    f->n_taps = n_taps;
    f->taps_complex = taps_complex ? 1 : 0;
    f->data_complex = data_complex ? 1 : 0;
    f->out_complex = (f->taps_complex || f->data_complex) ? 1 : 0;
    f->n_channels = n_channels;
    f->per_channel = per_channel ? 1 : 0;
    f->device = device;
    f->block_hint = block_hint;
    f->blk = block_hint ? block_hint + 1 - n_taps : 0;
    f->fft_ok = fir_choose_partition(n_taps, &f->ovl, &f->parts);
    if (f->per_channel && !f->fft_ok) {
        delete f;
        set_error("fir_create_per_channel: %d taps exceed %d partitions of the 4096-point kernel", n_taps, FIR_MAX_PARTS);
        return SFE_ERANGE;
    }
    if (f->fft_ok) f->hl = f->parts * f->ovl;                       // history the slowest partition reaches back to
    else f->hl = ((n_taps - 1 + 255) / 256) * 256;                  // beyond FIR_MAX_PARTS partitions: direct kernel only
    auto fail = [&](int code) { fir_free(f); return code; };
#define TRY(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return fail(hip_fail(e__, #call)); } while (0)
    TRY(hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking));
    f->h_taps_all.assign(taps, taps + (size_t)n_taps * (taps_complex ? 2 : 1) * (per_channel ? n_channels : 1));
    if (f->fft_ok) {
        rc = fir_build_tables(f, taps);
        if (rc != SFE_OK) return fail(rc);
    }
    if (!f->taps_complex && !f->per_channel) {
        f->h_taps.assign(taps, taps + n_taps);
        TRY(hipMalloc(&f->d_taps, (size_t)n_taps * sizeof(float)));
        TRY(hipMemcpy(f->d_taps, taps, (size_t)n_taps * sizeof(float), hipMemcpyHostToDevice));
    }
    for (int i = 0; i < 2; i++) {
        TRY(hipMalloc(&f->d_hist[i], f->hist_bytes()));
        TRY(hipMemset(f->d_hist[i], 0, f->hist_bytes()));
    }
    if (f->blk > 0) {
        if (n_channels != 1) {
            set_error("fir_create: the host block path (block_hint) is single-channel");
            return fail(SFE_EINVAL);
        }
        const size_t in_e = f->data_complex ? 2 : 1, out_e = f->out_complex ? 2 : 1;
        const size_t hb = ((size_t)block_hint + 2) * (out_e > in_e ? out_e : in_e) * sizeof(float);
        TRY(hipHostMalloc((void **)&f->h_buf, hb));
        memset(f->h_buf, 0, hb);
        TRY(hipMalloc(&f->d_blk_in, (size_t)f->blk * in_e * sizeof(float)));
        TRY(hipMalloc(&f->d_blk_out, (size_t)f->blk * out_e * sizeof(float)));
        TRY(hipHostMalloc(&f->h_blk_out, hb));
    }
    TRY(hipDeviceSynchronize());
#undef TRY
    *out = f;
    return SFE_OK;
}

}  // namespace sfe

using namespace sfe;

extern "C" {

// ---------------------------------------------------------------------------- FIR
int sfe_dsp_fir_create(const float *taps, int n_taps, int taps_complex, int data_complex,
                       int n_channels, int block_hint, int device, sfe_fir_t *out)
{
    return fir_create_impl(taps, n_taps, taps_complex, data_complex, n_channels, block_hint, device, 0, out);
}

int sfe_dsp_fir_create_per_channel(const float *taps, int n_taps, int taps_complex, int n_channels, int device,
                                   sfe_fir_t *out)
{
    return fir_create_impl(taps, n_taps, taps_complex, 1, n_channels, 0, device, 1, out);
}

int sfe_dsp_fir_plan(int n_taps, int *overlap, int *partitions, int *advance)
{
    if (n_taps < 1) return SFE_EINVAL;
    int o = 0, p = 0;
    if (!fir_choose_partition(n_taps, &o, &p)) {
        set_error("fir_plan: %d taps exceed %d partitions of the 4096-point kernel", n_taps, FIR_MAX_PARTS);
        return SFE_ERANGE;
    }
    if (overlap) *overlap = o;
    if (partitions) *partitions = p;
    if (advance) *advance = FFT_N - o;
    return SFE_OK;
}

int sfe_dsp_fir_host_buffer(sfe_fir_t h, float **buf, int *blk)
{
    Fir *f = as_fir(h);
    if (!f || !f->h_buf) {
        set_error("fir_host_buffer: handle was created without block_hint");
        return SFE_ESTATE;
    }
    if (buf) *buf = f->h_buf;
    if (blk) *blk = f->blk;
    return SFE_OK;
}

int sfe_dsp_fir_process_block(sfe_fir_t h)
{
    Fir *f = as_fir(h);
    if (!f || !f->h_buf) {
        set_error("fir_process_block: handle was created without block_hint");
        return SFE_ESTATE;
    }
    SFE_ON_DEVICE(f->device);
    const size_t in_b = (size_t)f->blk * (f->data_complex ? 8 : 4);
    const size_t out_b = (size_t)f->blk * (f->out_complex ? 8 : 4);
    if (f->parts == 1 && (size_t)f->blk <= f->zc_max) {
        int rc = fir_run(f, f->h_buf, f->h_blk_out, (size_t)f->blk, (size_t)f->blk, (size_t)f->blk, f->stream);
        if (rc != SFE_OK) return rc;
        SFE_HIP(hipStreamSynchronize(f->stream));
        memcpy(f->h_buf, f->h_blk_out, out_b);
        return SFE_OK;
    }
    SFE_HIP(hipMemcpyAsync(f->d_blk_in, f->h_buf, in_b, hipMemcpyHostToDevice, f->stream));
    int rc = fir_run(f, f->d_blk_in, f->d_blk_out, (size_t)f->blk, (size_t)f->blk, (size_t)f->blk, f->stream);
    if (rc != SFE_OK) return rc;
    SFE_HIP(hipMemcpyAsync(f->h_buf, f->d_blk_out, out_b, hipMemcpyDeviceToHost, f->stream));
    SFE_HIP(hipStreamSynchronize(f->stream));
    return SFE_OK;
}

int sfe_dsp_fir_process_stream(sfe_fir_t h, const void *d_in, void *d_out, size_t n,
                               size_t in_stride, size_t out_stride, sfe_stream_t stream)
{
    Fir *f = as_fir(h);
    if (!f || (n && (!d_in || !d_out))) {
        set_error("fir_process_stream: null handle or buffer");
        return SFE_EINVAL;
    }
    if (f->n_channels > 1 && (in_stride < n || out_stride < n)) {
        set_error("fir_process_stream: channel stride smaller than n");
        return SFE_EINVAL;
    }
    // bytes per element as the kernels address them
    const size_t isz = f->in_u8 ? (f->data_complex ? 2 : 1) : (f->data_complex ? 8 : 4);
    const size_t osz = f->out_complex ? 8 : 4;
    if ((reinterpret_cast<uintptr_t>(d_in) & (isz - 1)) ||
        (reinterpret_cast<uintptr_t>(d_out) & (f->out_tx10 ? 0 : osz - 1))) {
        set_error("fir_process_stream: buffers must be aligned to their element (cf32 8 B, f32 4 B, u8 (I,Q) pairs 2 B; 10-bit output: none)");
        return SFE_EINVAL;
    }
    if (f->out_tx10 && f->n_channels > 1 && ((out_stride * (f->out_complex ? 2 : 1)) & 3)) {
        set_error("fir_process_stream: 10-bit output packs 4 floats per group: out_stride must keep channels on group boundaries");
        return SFE_EINVAL;
    }
    {
        const size_t in_b = ((size_t)(f->n_channels - 1) * in_stride + n) * isz;
        const size_t out_b = f->out_tx10 ? (((size_t)(f->n_channels - 1) * out_stride + n) * (f->out_complex ? 2 : 1) / 4 + 1) * 5
                                         : ((size_t)(f->n_channels - 1) * out_stride + n) * osz;
        if (ranges_overlap(d_in, in_b, d_out, out_b)) {
            set_error("fir_process_stream: input and output ranges overlap (in-place operation is not supported)");
            return SFE_EINVAL;
        }
    }
    SFE_ON_DEVICE(f->device);
    return fir_run(f, d_in, d_out, n, in_stride, out_stride, (hipStream_t)stream);
}

int sfe_dsp_fir_process_host(sfe_fir_t h, const void *in, void *out, size_t n)
{
    Fir *f = as_fir(h);
    if (!f || (n && (!in || !out))) {
        set_error("fir_process_host: null handle or buffer");
        return SFE_EINVAL;
    }
    if (f->n_channels != 1) {
        set_error("fir_process_host: single-channel handles only");
        return SFE_EINVAL;
    }
    SFE_ON_DEVICE(f->device);
    const size_t in_e = f->data_complex ? 8 : 4, out_e = f->out_complex ? 8 : 4;
    const size_t CH = (size_t)1 << 20;            // samples per staged chunk
    if (!f->h_stage || !f->d_st_in || !f->d_st_out) {
        // allocate into locals and commit only when all three exist: a failed later allocation must
        // not leave a half-built staging set behind for the next call to trip over
        void *hs = nullptr, *di = nullptr, *dn = nullptr, *ho = nullptr;
        const size_t zc = f->zc_max < CH ? f->zc_max : CH;
        hipError_t e = hipHostMalloc(&hs, CH * (in_e > out_e ? in_e : out_e));
        if (e == hipSuccess) e = hipMalloc(&di, CH * in_e);
        if (e == hipSuccess) e = hipMalloc(&dn, CH * out_e);
        if (e == hipSuccess && zc) e = hipHostMalloc(&ho, zc * out_e);
        if (e != hipSuccess) {
            if (hs) (void)hipHostFree(hs);
            if (di) (void)hipFree(di);
            if (dn) (void)hipFree(dn);
            if (ho) (void)hipHostFree(ho);
            return hip_fail(e, "fir_process_host staging");
        }
        f->h_stage = hs;
        f->d_st_in = di;
        f->d_st_out = dn;
        f->h_stage_out = ho;
        f->stage_samples = CH;
    }
    const char *ip = static_cast<const char *>(in);
    char *op = static_cast<char *>(out);
    for (size_t off = 0; off < n; off += CH) {
        const size_t m = n - off < CH ? n - off : CH;
        memcpy(f->h_stage, ip + off * in_e, m * in_e);
        if (f->parts == 1 && f->h_stage_out && m <= f->zc_max) {          // small call: no DMA copies
            int rc = fir_run(f, f->h_stage, f->h_stage_out, m, m, m, f->stream);
            if (rc != SFE_OK) return rc;
            SFE_HIP(hipStreamSynchronize(f->stream));
            memcpy(op + off * out_e, f->h_stage_out, m * out_e);
            continue;
        }
        SFE_HIP(hipMemcpyAsync(f->d_st_in, f->h_stage, m * in_e, hipMemcpyHostToDevice, f->stream));
        int rc = fir_run(f, f->d_st_in, f->d_st_out, m, m, m, f->stream);
        if (rc != SFE_OK) return rc;
        SFE_HIP(hipMemcpyAsync(f->h_stage, f->d_st_out, m * out_e, hipMemcpyDeviceToHost, f->stream));
        SFE_HIP(hipStreamSynchronize(f->stream));
        memcpy(op + off * out_e, f->h_stage, m * out_e);
    }
    return SFE_OK;
}


// Carried state from a halo: the stream is about to continue at a sample whose predecessors are
// d_prev[0 .. n_prev) (float32, the handle's element type, per channel at `stride`) -- e.g. the
// first call of a span when one long stream is cut across GPUs (blkconv.cxx:105-109: what the
// reference carries in m_overlap is determined by exactly these n_taps-1 input samples).
int sfe_dsp_fir_load_history(sfe_fir_t h, const void *d_prev, size_t n_prev, size_t stride, sfe_stream_t stream)
{
    Fir *f = as_fir(h);
    if (!f || (n_prev && !d_prev)) {
        set_error("fir_load_history: null handle or buffer");
        return SFE_EINVAL;
    }
    if (f->n_channels > 1 && stride < n_prev) {
        set_error("fir_load_history: channel stride smaller than n_prev");
        return SFE_EINVAL;
    }
    if (reinterpret_cast<uintptr_t>(d_prev) & (f->data_complex ? 7 : 3)) {
        set_error("fir_load_history: buffer must be aligned to its element");
        return SFE_EINVAL;
    }
    SFE_ON_DEVICE(f->device);
    hipStream_t s = (hipStream_t)stream;
    SFE_HIP(hipMemsetAsync(f->d_hist[f->cur], 0, f->hist_bytes(), s));        // shorter halos: zeros in front
    if (n_prev) {
        int rc = launch_history_update(d_prev, (long long)n_prev, (long long)stride, f->d_hist[f->cur], f->d_hist[f->cur ^ 1],
                                       f->hl, f->data_complex ? 2 : 1, f->n_channels, s, 0);
        if (rc != SFE_OK) return rc;
        if (f->captured) SFE_HIP(hipMemcpyAsync(f->d_hist[f->cur], f->d_hist[f->cur ^ 1], f->hist_bytes(), hipMemcpyDeviceToDevice, s));
        else f->cur ^= 1;
    }
    return SFE_OK;
}

int sfe_dsp_fir_set_input_format(sfe_fir_t h, int fmt)
{
    Fir *f = as_fir(h);
    if (!f || (fmt != SFE_FMT_F32 && fmt != SFE_FMT_U8)) return SFE_EINVAL;
    if (f->piped && (fmt == SFE_FMT_U8) != (f->in_u8 != 0)) {
        // a pipe sized its pinned and device batches from the item format at create
        set_error("fir_set_input_format: a pipe over this handle has frozen its item format (destroy the pipe first)");
        return SFE_ESTATE;
    }
    if (fmt == SFE_FMT_U8 && (!f->fft_ok || f->taps_complex)) {
        set_error("fir_set_input_format: u8 input needs the FFT kernel with real taps");
        return SFE_ESTATE;
    }
    f->in_u8 = fmt == SFE_FMT_U8;
    return SFE_OK;
}

// A filter of ~2818..3841 taps is served fastest by TWO partitions (fir_choose_partition), but the
// 10-bit packed output exists for the single-launch kernel only (partitions after the first
// read-modify-write float32).  One transform can still overlap such a filter (hl1 < 4096), so the
// handle is re-planned as ONE partition: new spectrum table, new (zeroed) history.  ADVICE r2.
static int fir_replan_single(Fir *f)
{
    const int need = f->n_taps > 1 ? f->n_taps - 1 : 1;
    const int hl1 = ((need + 255) / 256) * 256;
    if (hl1 >= FFT_N) return SFE_ESTATE;
    if (f->started || f->captured) {
        // the re-plan zeroes the carried state and frees buffers a captured graph names
        set_error("fir_set_output_format: this filter must be re-planned as one launch for 10-bit output, which "
                  "restarts the stream: set the format before the first process call (or after sfe_dsp_fir_reset)");
        return SFE_ESTATE;
    }
    SFE_HIP(hipDeviceSynchronize());
    // build the new plan beside the old one and swap only when all of it exists: a failure leaves the handle as it was
    struct Saved {
        v2f *hs, *tw1, *tw2;
        unsigned *ticket;
        void *hist[2];
        int parts, ovl, hl, cur;
        bool fft_ok;
    } old = {f->d_hs, f->d_tw1, f->d_tw2, f->d_ticket, {f->d_hist[0], f->d_hist[1]}, f->parts, f->ovl, f->hl, f->cur, f->fft_ok};
    f->d_hs = f->d_tw1 = f->d_tw2 = nullptr;
    f->d_ticket = nullptr;
    f->d_hist[0] = f->d_hist[1] = nullptr;
    f->parts = 1;
    f->ovl = hl1;
    f->hl = hl1;
    f->cur = 0;
    int rc = fir_build_tables(f, f->h_taps_all.data());
    for (int i = 0; i < 2 && rc == SFE_OK; i++) {
        hipError_t e = hipMalloc(&f->d_hist[i], f->hist_bytes());
        if (e == hipSuccess) e = hipMemset(f->d_hist[i], 0, f->hist_bytes());
        if (e != hipSuccess) rc = hip_fail(e, "fir_replan_single: history");
    }
    if (rc == SFE_OK) {
        void *drop[6] = {old.hs, old.tw1, old.tw2, old.ticket, old.hist[0], old.hist[1]};
        for (void *q : drop)
            if (q) (void)hipFree(q);
        return SFE_OK;
    }
    void *drop[6] = {f->d_hs, f->d_tw1, f->d_tw2, f->d_ticket, f->d_hist[0], f->d_hist[1]};
    for (void *q : drop)
        if (q) (void)hipFree(q);
    f->d_hs = old.hs;
    f->d_tw1 = old.tw1;
    f->d_tw2 = old.tw2;
    f->d_ticket = old.ticket;
    f->d_hist[0] = old.hist[0];
    f->d_hist[1] = old.hist[1];
    f->parts = old.parts;
    f->ovl = old.ovl;
    f->hl = old.hl;
    f->cur = old.cur;
    f->fft_ok = old.fft_ok;
    return rc;
}

int sfe_dsp_fir_set_output_format(sfe_fir_t h, int fmt)
{
    Fir *f = as_fir(h);
    if (!f || (fmt != SFE_FMT_F32 && fmt != SFE_FMT_TX10)) return SFE_EINVAL;
    if (f->piped && (fmt == SFE_FMT_TX10) != (f->out_tx10 != 0)) {
        set_error("fir_set_output_format: a pipe over this handle has frozen its item format (destroy the pipe first)");
        return SFE_ESTATE;
    }
    if (fmt == SFE_FMT_TX10 && f->fft_ok && f->parts > 1 && f->data_complex == f->out_complex) {
        // up to 3841 taps one transform still overlaps the filter: re-plan as a single launch (the carried state is zeroed:
        // formats are set before a stream starts)
        SFE_ON_DEVICE(f->device);
        int rc = fir_replan_single(f);
        if (rc != SFE_OK && rc != SFE_ESTATE) return rc;
    }
    if (fmt == SFE_FMT_TX10 && (!f->fft_ok || f->parts > 1 || f->data_complex != f->out_complex)) {
        set_error("fir_set_output_format: 10-bit output needs the single-launch FFT kernel (a filter that one 4096-point "
                  "transform can overlap: up to 3841 taps) and a real->real or complex->complex stream");
        return SFE_ESTATE;
    }
    f->out_tx10 = fmt == SFE_FMT_TX10;
    return SFE_OK;
}

int sfe_dsp_fir_set_algo(sfe_fir_t h, int algo)
{
    Fir *f = as_fir(h);
    if (!f || algo < SFE_FIR_ALGO_AUTO || algo > SFE_FIR_ALGO_FFT) return SFE_EINVAL;
    f->algo = algo;
    return SFE_OK;
}

int sfe_dsp_fir_set_variant(sfe_fir_t h, int variant)
{
    Fir *f = as_fir(h);
    if (!f || variant < SFE_FIR_VARIANT_AUTO || variant > SFE_FIR_VARIANT_WAVE_PRIVATE) {
        set_error("fir_set_variant: -1 (measure) or 0..2");
        return SFE_EINVAL;
    }
    f->variant = variant;
    return SFE_OK;
}

int sfe_dsp_fir_get_variant(sfe_fir_t h, int *last_variant, int *calibrations, float *ms_by_variant)
{
    Fir *f = as_fir(h);
    if (!f) return SFE_EINVAL;
    if (last_variant) *last_variant = f->last_variant;
    if (calibrations) *calibrations = f->cal_runs;
    if (ms_by_variant)
        for (int v = 0; v < FIR_VAR_COUNT; v++) ms_by_variant[v] = f->cal_ms[v];
    return SFE_OK;
}

int sfe_dsp_fir_calibrate(sfe_fir_t h, const void *d_in, void *d_out, size_t n, size_t in_stride,
                          size_t out_stride, sfe_stream_t stream, int *chosen)
{
    Fir *f = as_fir(h);
    if (chosen) *chosen = SFE_FIR_VARIANT_REGISTER_LOADS;
    if (!f || !n || !d_in || !d_out) {
        set_error("fir_calibrate: null handle or buffer");
        return SFE_EINVAL;
    }
    if (f->n_channels > 1 && (in_stride < n || out_stride < n)) {
        set_error("fir_calibrate: channel stride smaller than n");
        return SFE_EINVAL;
    }
    if (!f->fft_ok || f->algo == SFE_FIR_ALGO_DIRECT) return SFE_OK;    // the direct kernel has one form
    SFE_ON_DEVICE(f->device);
    hipStream_t s = (hipStream_t)stream;
    if (stream_is_capturing(s)) {
        set_error("fir_calibrate: a measurement cannot be captured into a hipGraph");
        return SFE_ESTATE;
    }
    FirFftArgs a;
    fir_fill_args(f, a, d_in, d_out, n, in_stride, out_stride);      // hist_out stays null: the stream does not advance
    int best = FIR_VAR_REG;
    int rc = fir_calibrate(f, a, s, &best);
    if (rc == SFE_OK && chosen) *chosen = best;
    return rc;
}

int sfe_dsp_fir_forget_calibrations(void)
{
    std::lock_guard<std::mutex> lk(g_fir_var_mutex);
    g_fir_var_cache.clear();
    return SFE_OK;
}

int sfe_dsp_fir_set_zero_copy_max(sfe_fir_t h, size_t max_samples)
{
    Fir *f = as_fir(h);
    if (!f) return SFE_EINVAL;
    if (f->h_stage) {
        // the pinned output staging of sfe_dsp_fir_process_host was sized from the old limit
        set_error("fir_set_zero_copy_max: set it before the first sfe_dsp_fir_process_host call");
        return SFE_ESTATE;
    }
    f->zc_max = max_samples;
    return SFE_OK;
}

int sfe_dsp_fir_reset(sfe_fir_t h)
{
    Fir *f = as_fir(h);
    if (!f) return SFE_EINVAL;
    SFE_ON_DEVICE(f->device);
    SFE_HIP(hipDeviceSynchronize());
    for (int i = 0; i < 2; i++) SFE_HIP(hipMemset(f->d_hist[i], 0, f->hist_bytes()));
    if (f->d_ticket) SFE_HIP(hipMemset(f->d_ticket, 0, FIR_TICKET_GROUPS_MAX * 128));
    f->started = false;         // `captured` stays: a graph made before the reset still names d_hist[cur]
    return SFE_OK;
}

int sfe_dsp_fir_destroy(sfe_fir_t h)
{
    Fir *f = as_fir(h);
    if (!f) return SFE_OK;
    if (f->piped) {
        set_error("fir_destroy: a pipe still borrows this handle (sfe_dsp_pipe_destroy first)");
        return SFE_ESTATE;
    }
    DeviceGuard g(f->device);
    (void)hipDeviceSynchronize();
    fir_free(f);
    return SFE_OK;
}

}  // extern "C"
