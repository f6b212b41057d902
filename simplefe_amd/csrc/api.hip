// api.hip -- the C ABI of libsfe_dsp.so (include/sfe_dsp.h): handles, host-side logic,
// launches.  Host code only; kernels live in fir_fft.hip / polyphase.hip / util.hip.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <new>
#include <tuple>
#include <numeric>
#include <unordered_map>
#include <utility>
#include <vector>

#include "common.h"
#include "timelaw.h"

namespace sfe {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char *what)
{
    set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    return e == hipErrorOutOfMemory ? SFE_ENOMEM : (e == hipErrorNoDevice ? SFE_ENODEV : SFE_EHIP);
}

int device_cu_count()
{
    constexpr int MAXDEV = 64;
    static std::atomic<int> cache[MAXDEV];          // zero-initialised; 0 = not asked yet
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return 256;
    int c = cache[dev].load(std::memory_order_relaxed);
    if (c > 0) return c;
    if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = 256;
    cache[dev].store(c, std::memory_order_relaxed);   // racing threads store the same value
    return c;
}

// DeviceGuard: common.h
#define SFE_ON_DEVICE(dev)                                   \
    DeviceGuard guard__(dev);                                \
    if (!guard__.ok) {                                       \
        set_error("cannot select device %d", (int)(dev));    \
        return SFE_EHIP;                                     \
    }

// [a, a + an) and [b, b + bn) share a byte
static bool ranges_overlap(const void *a, size_t an, const void *b, size_t bn)
{
    const uintptr_t pa = reinterpret_cast<uintptr_t>(a), pb = reinterpret_cast<uintptr_t>(b);
    return an && bn && pa < pb + bn && pb < pa + an;
}

static int use_device(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no HIP device available (%s): libsfe_dsp has no CPU fallback",
                  e != hipSuccess ? hipGetErrorString(e) : "device count 0");
        return SFE_ENODEV;
    }
    if (device < 0 || device >= n) {
        set_error("device %d out of range (0..%d)", device, n - 1);
        return SFE_EINVAL;
    }
    SFE_HIP(hipSetDevice(device));
    return SFE_OK;
}

// ---- tiled polyphase plans (common.h: PolyTiledPlan) ---------------------------------
// Fold (U, step, pos0) into zero-padded per-output-phase tap rows of equal length.
//   taps_pm: [U][plen] phase-major host taps.
struct PlanCache {
    std::map<std::pair<int, long long>, PolyTiledPlan> plans;   // (step, pos0) -> plan
    void clear()
    {
        for (auto &kv : plans) {
            if (kv.second.d_G) (void)hipFree(kv.second.d_G);
            if (kv.second.d_Gt) (void)hipFree(kv.second.d_Gt);
        }
        plans.clear();
    }
};

static long long floordiv_ll(long long a, long long b)
{
    long long q = a / b;
    return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q;
}

// Folds (U, step, pos0) into UP zero-padded tap rows of equal length Lp (a multiple of `quantum`):
// output UP*m + r = sum_q G[r][q] x[SP*m + e_max - Lp + 1 + q].
struct FoldedRows {
    int SP = 0, UP = 0, Lp = 0, e_max = 0;
    std::vector<float> G;
};
static FoldedRows fold_rows(const std::vector<float> &taps_pm, int U, int plen, int step, long long pos0,
                            int quantum_in_SP)
{
    FoldedRows f;
    const int g = std::gcd(step, U);
    f.SP = step / g;
    f.UP = U / g;
    // drop trailing all-zero taps (decimate's odd-izing zero, resample's last-phase padding)
    int plen_eff = plen;
    while (plen_eff > 1) {
        bool any = false;
        for (int ph = 0; ph < U; ph++) any = any || taps_pm[(size_t)ph * plen + plen_eff - 1] != 0.0f;
        if (any) break;
        plen_eff--;
    }
    std::vector<long long> o(f.UP);
    std::vector<int> ph(f.UP);
    long long e_max = -(1LL << 60), e_min = (1LL << 60);
    for (int r = 0; r < f.UP; r++) {
        const long long A = pos0 + (long long)r * step;
        o[r] = floordiv_ll(A, U);
        ph[r] = (int)(A - o[r] * U);
        e_max = o[r] > e_max ? o[r] : e_max;
        e_min = o[r] < e_min ? o[r] : e_min;
    }
    const int L = plen_eff + (int)(e_max - e_min);
    const int quantum = quantum_in_SP * f.SP;
    f.Lp = ((L + quantum - 1) / quantum) * quantum;
    f.e_max = (int)e_max;
    f.G.assign((size_t)f.UP * f.Lp, 0.0f);
    for (int r = 0; r < f.UP; r++)
        for (int q = 0; q < f.Lp; q++) {
            const long long j = o[r] - e_max + f.Lp - 1 - q;     // tap index met at local time q
            if (j >= 0 && j < plen_eff) f.G[(size_t)r * f.Lp + q] = taps_pm[(size_t)ph[r] * plen + j];
        }
    return f;
}

// returns nullptr when the shape has no tiled kernel (caller uses the generic one)
static const PolyTiledPlan *get_tiled_plan(PlanCache &cache, const std::vector<float> &taps_pm, int U, int plen,
                                           int step, long long pos0, int *rc)
{
    *rc = SFE_OK;
    auto key = std::make_pair(step, pos0);
    auto it = cache.plans.find(key);
    if (it != cache.plans.end()) return it->second.d_G ? &it->second : nullptr;
    PolyTiledPlan pl;
    const FoldedRows f = fold_rows(taps_pm, U, plen, step, pos0, 2);   // whole pairs of SP-sample chunks
    pl.SP = f.SP;
    pl.UP = f.UP;
    pl.Lp = f.Lp;
    pl.e_max = f.e_max;
    if (!poly_tiled_supported(pl.SP, pl.UP, pl.Lp)) {
        cache.plans[key] = pl;          // d_G == nullptr marks "unsupported"
        return nullptr;
    }
    hipError_t e = hipMalloc(&pl.d_G, f.G.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(pl.d_G, f.G.data(), f.G.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess && pl.UP <= 8) {
        // the same taps transposed, [local time][phase] padded to 8 phases (the runtime-shape kernel reads a row per tap)
        std::vector<float> gt((size_t)pl.Lp * 8, 0.0f);
        for (int r = 0; r < pl.UP; r++)
            for (int q = 0; q < pl.Lp; q++) gt[(size_t)q * 8 + r] = f.G[(size_t)r * pl.Lp + q];
        e = hipMalloc(&pl.d_Gt, gt.size() * sizeof(float));
        if (e == hipSuccess) e = hipMemcpy(pl.d_Gt, gt.data(), gt.size() * sizeof(float), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        if (pl.d_G) (void)hipFree(pl.d_G);
        if (pl.d_Gt) (void)hipFree(pl.d_Gt);
        *rc = hip_fail(e, "tiled plan upload");
        return nullptr;
    }
    auto ins = cache.plans.emplace(key, pl);
    return &ins.first->second;
}

// ---- transform-domain plans (common.h: PolyFftPlan) -----------------------------------
struct FftPlanCache {
    std::map<std::pair<int, long long>, PolyFftPlan> plans;
    void clear()
    {
        for (auto &kv : plans) {
            if (kv.second.d_H) (void)hipFree(kv.second.d_H);
            if (kv.second.d_tw) (void)hipFree(kv.second.d_tw);
        }
        plans.clear();
    }
};

// nullptr when the shape is not worth (or not instantiated for) the transform-domain kernel
// fft_mode: 0 = choose by the calibrated rule, 1 = always when instantiated, -1 = never
static const PolyFftPlan *get_fft_plan(FftPlanCache &cache, const std::vector<float> &taps_pm, int U, int plen,
                                       int step, long long pos0, int fft_mode, int *rc)
{
    *rc = SFE_OK;
    auto key = std::make_pair(step, pos0);
    auto it = cache.plans.find(key);
    if (it != cache.plans.end()) return it->second.d_H ? &it->second : nullptr;
    PolyFftPlan pl;
    const FoldedRows f = fold_rows(taps_pm, U, plen, step, pos0, 1);
    pl.SP = f.SP;
    pl.UP = f.UP;
    pl.R = poly_fft_segments(f.SP, f.UP);
    pl.Li = f.Lp / f.SP;
    pl.e_max = f.e_max;
    // Selection (measured over a grid of shapes at 2^26 samples, scripts/calibrate_rs_fft.py):
    //  - the overlap must leave a useful block: Li <= 192 (V = 257 - Li >= 65 of 256 points);
    //  - shapes neither tiled kernel takes fall to the generic kernel, which is 3-11x slower than this
    //    one: take the transform whenever it exists;
    //  - otherwise the transform wins once the direct form costs more than ~230 flop per (complex)
    //    input sample, scaled by how much of each 256-point block is overlap, and 1.4x later for
    //    UP = 4 (one segment per pass fills only 9 of the 16 lane groups).
    const double direct_flops = 2.0 * 2.0 * f.Lp * f.UP / f.SP;    // per complex input sample (or per pair of real ones)
    const int V = 257 - pl.Li;
    const int Lp2 = ((f.Lp + 2 * f.SP - 1) / (2 * f.SP)) * (2 * f.SP);
    const bool tiled_ok = poly_tiled_supported(f.SP, f.UP, Lp2);
    // (round 4: shapes without a compile-time tiled instantiation now run poly_rt_kernel, whose loops are not
    // unrolled over SP and UP: the transform takes over at half the arithmetic)
    const double threshold = 230.0 * 231.0 / (V > 0 ? V : 1) * (f.UP >= 4 ? 1.4 : 1.0) *
                             (poly_tiled_is_compiled(f.SP, f.UP, Lp2) ? 1.0 : 0.5);
    const bool forced = fft_mode > 0;
    if (!pl.R || pl.Li > 192 || fft_mode < 0 || (!forced && tiled_ok && direct_flops < threshold)) {
        cache.plans[key] = pl;
        return nullptr;
    }
    const int M = 256, SP = f.SP, UP = f.UP;
    std::vector<float> H((size_t)UP * SP * M * 2);
    const double w0 = -2.0 * M_PI / M;
    for (int r = 0; r < UP; r++)
        for (int cp = 0; cp < SP; cp++) {
            const int c = SP - 1 - cp;
            for (int b = 0; b < M; b++) {
                double re = 0.0, im = 0.0;
                for (int i = 0; i < pl.Li; i++) {
                    const double h = f.G[(size_t)r * f.Lp + (f.Lp - 1 - SP * i - c)];
                    const double ang = w0 * (double)((b * i) % M);
                    re += h * cos(ang);
                    im += h * sin(ang);
                }
                H[((size_t)(r * SP + cp) * M + b) * 2 + 0] = (float)(re / M);
                H[((size_t)(r * SP + cp) * M + b) * 2 + 1] = (float)(im / M);
            }
        }
    std::vector<float> tw(6 * 16 * 2);
    for (int k = 1; k < 4; k++)
        for (int l = 0; l < 16; l++) {
            const double a1 = w0 * (l * k), a4 = w0 * (4 * l * k);
            tw[((k - 1) * 16 + l) * 2 + 0] = (float)cos(a1);
            tw[((k - 1) * 16 + l) * 2 + 1] = (float)sin(a1);
            tw[((k + 2) * 16 + l) * 2 + 0] = (float)cos(a4);
            tw[((k + 2) * 16 + l) * 2 + 1] = (float)sin(a4);
        }
    hipError_t e = hipMalloc(&pl.d_H, H.size() * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&pl.d_tw, tw.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(pl.d_H, H.data(), H.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(pl.d_tw, tw.data(), tw.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (pl.d_H) (void)hipFree(pl.d_H);
        if (pl.d_tw) (void)hipFree(pl.d_tw);
        *rc = hip_fail(e, "transform-domain plan upload");
        return nullptr;
    }
    auto ins = cache.plans.emplace(key, pl);
    return &ins.first->second;
}

// ---- f32-MFMA plans (common.h: PolyMfmaPlan) ------------------------------------------
struct MfmaCache {
    std::map<std::pair<int, long long>, PolyMfmaPlan> plans;
    void clear()
    {
        for (auto &kv : plans)
            if (kv.second.d_A) (void)hipFree(kv.second.d_A);
        plans.clear();
    }
};

static const PolyMfmaPlan *get_mfma_plan(MfmaCache &cache, const std::vector<float> &taps_pm, int U, int plen,
                                         int step, long long pos0, int *rc)
{
    *rc = SFE_OK;
    auto key = std::make_pair(step, pos0);
    auto it = cache.plans.find(key);
    if (it != cache.plans.end()) return it->second.d_A ? &it->second : nullptr;
    PolyMfmaPlan pl;
    const int g = std::gcd(step, U);
    const int SP = step / g, UP = U / g;
    if (UP > 16) {
        cache.plans[key] = pl;
        return nullptr;
    }
    const int DM = 16 / UP;
    pl.RG = UP * DM;
    pl.GS = SP * DM;
    int plen_eff = plen;
    while (plen_eff > 1) {
        bool any = false;
        for (int ph = 0; ph < U; ph++) any = any || taps_pm[(size_t)ph * plen + plen_eff - 1] != 0.0f;
        if (any) break;
        plen_eff--;
    }
    std::vector<long long> o(UP);
    std::vector<int> ph(UP);
    long long e_max = -(1LL << 60), e_min = (1LL << 60);
    for (int r = 0; r < UP; r++) {
        const long long A = pos0 + (long long)r * step;
        o[r] = floordiv_ll(A, U);
        ph[r] = (int)(A - o[r] * U);
        e_max = o[r] > e_max ? o[r] : e_max;
        e_min = o[r] < e_min ? o[r] : e_min;
    }
    const long long u_hi = (long long)SP * (DM - 1) + e_max;
    const long long K0 = u_hi - (e_min - (plen_eff - 1)) + 1;
    pl.Kp = (int)((K0 + 3) / 4 * 4);
    pl.u_lo = (int)(u_hi - pl.Kp + 1);
    pl.density = (float)((double)pl.RG * plen_eff / (16.0 * pl.Kp));
    if (!poly_mfma_fits(pl.GS, pl.RG, pl.Kp)) {
        cache.plans[key] = pl;          // d_A == nullptr marks "unsupported"
        return nullptr;
    }
    const int ksteps = pl.Kp / 4;
    std::vector<float> Af((size_t)ksteps * 64, 0.0f);
    for (int ks = 0; ks < ksteps; ks++)
        for (int lane = 0; lane < 64; lane++) {
            const int row = lane & 15, kk = 4 * ks + (lane >> 4);
            if (row >= pl.RG) continue;
            const int d = row / UP, r = row % UP;
            const long long jt = (long long)SP * d + o[r] - u_hi + kk;      // tap met at window pos u_hi - kk
            if (jt >= 0 && jt < plen_eff) Af[(size_t)ks * 64 + lane] = taps_pm[(size_t)ph[r] * plen + jt];
        }
    hipError_t e = hipMalloc(&pl.d_A, Af.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(pl.d_A, Af.data(), Af.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (pl.d_A) (void)hipFree(pl.d_A);
        *rc = hip_fail(e, "mfma plan upload");
        return nullptr;
    }
    auto ins = cache.plans.emplace(key, pl);
    return &ins.first->second;
}

// ------------------------------------------------------------------------------ FIR
struct Fir {
    uint32_t magic = 0x46495231u;   // 'FIR1': catches stale or foreign handles
    int n_taps = 0, taps_complex = 0, data_complex = 0, out_complex = 0, n_channels = 1;
    int device = 0, algo = SFE_FIR_ALGO_AUTO, in_u8 = 0, out_tx10 = 0;
    int blk = 0, block_hint = 0;
    int hl = 0;                 // carried history per channel, samples (multiple of 256)
    int ovl = 0;                // overlap of one transform (multiple of 256): each launch applies `ovl` (+1) taps
    int variant = FIR_VAR_AUTO; // data movement of the cf32 kernel: measured per device and shape unless sfe_dsp_fir_set_variant fixed it
    int last_variant = FIR_VAR_AUTO, cal_runs = 0;      // what the last bulk call ran; calibrations this handle made
    float cal_ms[FIR_VAR_COUNT] = {0.0f, 0.0f, 0.0f};   // medians of this handle's last calibration, by variant
    int piped = 0;              // pipes alive over this handle: they froze its item formats, so the format setters refuse
    int per_channel = 0;        // taps given per channel ([n_channels][n_taps]): one spectrum set per channel
    int parts = 1;              // partitions of the tap vector, one launch each (filters longer than one overlap)
    bool fft_ok = false;
    v2f *d_hs = nullptr, *d_tw1 = nullptr, *d_tw2 = nullptr;
    unsigned *d_ticket = nullptr;   // work counter of the persistent FFT kernel (zero between launches)
    float *d_taps = nullptr;    // real taps for the direct kernel
    std::vector<float> h_taps;  // host copy (direct-kernel plan)
    std::vector<float> h_taps_all;   // every tap as given at create (complex pairs / per-channel rows included): re-planning
    PlanCache plans;
    void *d_hist[2] = {nullptr, nullptr};
    int cur = 0;
    bool captured = false;      // a call of this handle sits in a hipGraph that names d_hist[cur]: the state stays there (fir_run)
    bool started = false;       // samples have gone through since create / reset
    // class-compatible host block path
    float *h_buf = nullptr;     // pinned, block_hint+2 floats (blkconv.cxx:44 sizes it so)
    void *d_blk_in = nullptr, *d_blk_out = nullptr;
    void *h_blk_out = nullptr;  // pinned: the kernel's output of a zero-copy block (then copied over h_buf)
    // host-pointer streaming path (sfe_dsp_fir_process_host): chunked pinned + device staging
    void *h_stage = nullptr, *d_st_in = nullptr, *d_st_out = nullptr;
    void *h_stage_out = nullptr;
    size_t stage_samples = 0;
    // Calls of at most zc_max samples skip the two DMA copies: the kernel reads the pinned host buffer
    // and writes a pinned host buffer itself (one launch + one wait instead of copy, launch, copy, wait).
    // Only where the kernel reads its input once (parts == 1).  sfe_dsp_fir_set_zero_copy_max; 0 disables.
    size_t zc_max = (size_t)1 << 20;
    hipStream_t stream = nullptr;
    size_t hist_bytes() const { return (size_t)n_channels * hl * (data_complex ? 8 : 4); }
};

static Fir *as_fir(void *h)
{
    Fir *f = static_cast<Fir *>(h);
    if (f && f->magic != 0x46495231u) {
        set_error("not a live FIR handle");
        return nullptr;
    }
    return f;
}

static void fir_free(Fir *f)
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
static bool stream_is_capturing(hipStream_t s)
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

static int fir_run(Fir *f, const void *d_in, void *d_out, size_t n, size_t in_stride,
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

// ------------------------------------------------------------------ resample / decimate
struct Rs {
    uint32_t magic = 0x52533031u;   // 'RS01'
    int U = 1, n_taps = 0, plen = 0, blksize = 0, data_complex = 0, n_channels = 1;
    int device = 0, mode = SFE_RS_RESAMPLE, exact_stream = 0, in_u8 = 0;
    int piped = 0;                         // pipes alive over this handle (they froze its input format)
    int fft_mode = 0;                      // sfe_dsp_rs_set_algo: 1 force the transform-domain kernel, -1 never, 0 the calibrated rule
    int use_mfma = 0;                      // sfe_dsp_rs_set_algo(SFE_RS_ALGO_MFMA): the matrix-pipe form (measured slower; opt-in)
    int hl = 0;
    float *d_taps = nullptr;               // [U][plen] phase-major
    std::vector<float> h_taps_pm;          // host copy of the same (tiled plans)
    PlanCache plans;
    MfmaCache mfma_plans;
    FftPlanCache fft_plans;
    unsigned *d_ticket = nullptr;          // work counters of the transform-domain kernel
    struct Fir *gen_tables = nullptr;      // general rate in the transform domain (poly_gen.hip): the U phases' spectra and the
                                           // twiddle bases, built by the FIR's own table builder (one "channel" per phase)
    bool gen_tried = false;
    void *d_hist[2] = {nullptr, nullptr};
    int cur = 0;
    bool captured = false;                 // a call of this handle sits in a hipGraph that names d_hist[cur] (see fir_carry_state)
    sfe_rs_timestate ts = {0, 0.0f, 0};
    // class-compatible host path staging (one channel)
    void *d_in = nullptr, *d_out = nullptr;
    long long *d_pos = nullptr;
    float *d_mu = nullptr;
    size_t out_cap = 0, sched_cap = 0;
    void *h_stage = nullptr;               // pinned: in/out staging
    size_t h_stage_bytes = 0;
    long long *h_pos = nullptr;            // pinned schedule staging
    float *h_mu = nullptr;
    void *d_segs = nullptr, *d_chunks = nullptr, *h_segs = nullptr, *h_chunks = nullptr;   // run-length plans
    size_t segs_cap = 0, chunks_cap = 0;
    hipEvent_t ev_plan = nullptr;          // recorded behind a call's plan uploads: the pinned staging is free again once it fires
    hipStream_t plan_stream = nullptr;     // ... on this stream (the device-side plan arrays are ordered by it)
    // General rate: the plan of one blksize-sample reference call depends only on the time state the call
    // starts in, and that state is a multiple of the float32 grid of the call's LAST binade inside
    // [-1, step) -- a few thousand possible values (blksize*U = 16384: 2^-10 apart) -- so plans are
    // memoised per start state: a 2^28-sample call replays 65 536 reference calls as table look-ups
    // instead of 65 536 x ~40 runs of float arithmetic (36 ms -> ~2 ms on the host), and the run table
    // lives on the device across calls (only what is new is uploaded).
    struct SegPlanRef {
        int seg_first, n_seg, n_out;
        sfe_rs_timestate after;
        int next = -1;                     // index of the plan for the state this call ends in, once it has been met:
                                           // a stream of full-size calls then walks the plans by index, no hashing
    };
    std::unordered_map<uint64_t, int> seg_memo;      // start state -> index into seg_refs
    std::vector<SegPlanRef> seg_refs;
    std::vector<TlSeg> seg_table;          // runs of the memoised calls, in the order they were first met
    size_t seg_uploaded = 0;               // leading entries of seg_table already in d_segs
    float memo_rate = 0.0f;                // the memo is for one (rate, blksize)
    int memo_m = 0;
    hipStream_t stream = nullptr;
    int esz() const { return data_complex ? 8 : 4; }
};

static Rs *as_rs(void *h)
{
    Rs *r = static_cast<Rs *>(h);
    if (r && r->magic != 0x52533031u) {
        set_error("not a live resample/decimate handle");
        return nullptr;
    }
    return r;
}

static void rs_free(Rs *r)
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

const char *sfe_dsp_version(void) { return "simplefe_amd 0.1 (gfx950)"; }
const char *sfe_dsp_last_error(void) { return g_err; }

int sfe_dsp_device_count(int *count)
{
    if (!count) return SFE_EINVAL;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    *count = (e == hipSuccess) ? n : 0;
    return SFE_OK;
}

int sfe_dsp_set_device(int device) { return use_device(device); }

int sfe_dsp_get_device(int *device)
{
    if (!device) return SFE_EINVAL;
    *device = 0;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        set_error("no HIP device available: libsfe_dsp has no CPU fallback");
        return SFE_ENODEV;
    }
    SFE_HIP(hipGetDevice(device));
    return SFE_OK;
}

int sfe_dsp_sync(sfe_stream_t stream)
{
    SFE_HIP(hipStreamSynchronize((hipStream_t)stream));
    return SFE_OK;
}

int sfe_dsp_malloc(void **dptr, size_t bytes)
{
    if (!dptr) return SFE_EINVAL;
    SFE_HIP(hipMalloc(dptr, bytes ? bytes : 16));
    return SFE_OK;
}
int sfe_dsp_free(void *dptr)
{
    if (dptr) SFE_HIP(hipFree(dptr));
    return SFE_OK;
}
int sfe_dsp_host_alloc(void **hptr, size_t bytes)
{
    if (!hptr) return SFE_EINVAL;
    SFE_HIP(hipHostMalloc(hptr, bytes ? bytes : 16));
    return SFE_OK;
}
int sfe_dsp_host_free(void *hptr)
{
    if (hptr) SFE_HIP(hipHostFree(hptr));
    return SFE_OK;
}
int sfe_dsp_memcpy_h2d(void *dptr, const void *hptr, size_t bytes, sfe_stream_t stream)
{
    SFE_HIP(hipMemcpyAsync(dptr, hptr, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return SFE_OK;
}
int sfe_dsp_memcpy_d2h(void *hptr, const void *dptr, size_t bytes, sfe_stream_t stream)
{
    SFE_HIP(hipMemcpyAsync(hptr, dptr, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return SFE_OK;
}
int sfe_dsp_memset(void *dptr, int value, size_t bytes, sfe_stream_t stream)
{
    SFE_HIP(hipMemsetAsync(dptr, value, bytes, (hipStream_t)stream));
    return SFE_OK;
}

struct Timer {
    hipEvent_t a, b;
};
int sfe_dsp_timer_create(sfe_timer_t *t)
{
    if (!t) return SFE_EINVAL;
    *t = nullptr;
    Timer *x = new (std::nothrow) Timer;
    if (!x) return SFE_ENOMEM;
    hipError_t e = hipEventCreate(&x->a);
    if (e != hipSuccess) {
        delete x;
        return hip_fail(e, "hipEventCreate");
    }
    e = hipEventCreate(&x->b);
    if (e != hipSuccess) {
        (void)hipEventDestroy(x->a);
        delete x;
        return hip_fail(e, "hipEventCreate");
    }
    *t = x;
    return SFE_OK;
}
int sfe_dsp_timer_start(sfe_timer_t t, sfe_stream_t stream)
{
    SFE_HIP(hipEventRecord(static_cast<Timer *>(t)->a, (hipStream_t)stream));
    return SFE_OK;
}
int sfe_dsp_timer_stop(sfe_timer_t t, sfe_stream_t stream)
{
    SFE_HIP(hipEventRecord(static_cast<Timer *>(t)->b, (hipStream_t)stream));
    return SFE_OK;
}
int sfe_dsp_timer_elapsed_ms(sfe_timer_t t, float *ms)
{
    Timer *x = static_cast<Timer *>(t);
    SFE_HIP(hipEventSynchronize(x->b));
    SFE_HIP(hipEventElapsedTime(ms, x->a, x->b));
    return SFE_OK;
}
int sfe_dsp_timer_destroy(sfe_timer_t t)
{
    Timer *x = static_cast<Timer *>(t);
    if (!x) return SFE_OK;
    (void)hipEventDestroy(x->a);
    (void)hipEventDestroy(x->b);
    delete x;
    return SFE_OK;
}

int sfe_dsp_synth_fill(void *dptr, uint64_t n_floats, uint32_t seed, uint32_t channel,
                       uint64_t first, sfe_stream_t stream)
{
    if (!dptr && n_floats) return SFE_EINVAL;
    return launch_synth_fill(static_cast<float *>(dptr), n_floats, seed, channel, first,
                             (hipStream_t)stream);
}

// ---------------------------------------------------------------------------- FIR
static int fir_create_impl(const float *taps, int n_taps, int taps_complex, int data_complex,
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

int sfe_dsp_rs_process_stream(sfe_rs_t h, const void *d_in, size_t n_in, size_t in_stride,
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
            rc = launch_poly_tiled(*pl, ta, r->data_complex, r->exact_stream, r->in_u8, r->n_channels, s);
        } else {
            if (r->in_u8) {
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
        if (r->in_u8) {
            set_error("rs_process_stream: u8 input is supported for integer-valued steps only");
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
        for (size_t off = 0; off < n_in; off += (size_t)r->blksize) {
            const int m = (int)((n_in - off) < (size_t)r->blksize ? (n_in - off) : (size_t)r->blksize);
            const int cap = (int)ceilf((float)m / rate) + 2;
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
        // Bulk calls of complex streams at rate >= 1 in fused arithmetic: the transform-domain kernel (poly_gen.hip) --
        // all U phases of every input by one forward and U inverse 4096-point transforms, outputs picked and blended from
        // LDS by the same runs.  sfe_dsp_rs_set_algo(SFE_RS_ALGO_DIRECT) and the exact mode keep poly_seg_kernel.
        rc = SFE_ESTATE;
        if (!r->exact_stream && r->data_complex && !r->in_u8 && r->fft_mode >= 0 && stepf >= (float)r->U &&
            (r->fft_mode > 0 || (n_in >= ((size_t)1 << 16) && r->plen >= 12))) {
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
                ga.ovl = r->gen_tables->ovl;
                ga.blksize = r->blksize;
                ga.n_chunks = (int)chunks.size();
                int max_runs = 0;
                for (size_t i = 0; i < chunks.size(); i++) {
                    const int two = chunks[i].n_seg + (i + 1 < chunks.size() ? chunks[i + 1].n_seg : 0);
                    max_runs = two > max_runs ? two : max_runs;
                }
                rc = launch_poly_gen(ga, max_runs, stepf, r->n_channels, s);
            }
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

// ------------------------------------------------------------------------ converters
int sfe_dsp_rx_u8_to_f32(const void *d_bytes, void *d_floats, size_t n_bytes, sfe_stream_t stream)
{
    if (n_bytes && (!d_bytes || !d_floats)) return SFE_EINVAL;
    return launch_rx_u8_to_f32(static_cast<const uint8_t *>(d_bytes), static_cast<float *>(d_floats),
                               n_bytes, (hipStream_t)stream);
}

int sfe_dsp_tx_f32_to_10bit(const void *d_floats, void *d_bytes, size_t n_floats, sfe_stream_t stream)
{
    if (n_floats && (!d_bytes || !d_floats)) return SFE_EINVAL;
    return launch_tx_f32_to_10bit(static_cast<const float *>(d_floats), static_cast<uint8_t *>(d_bytes),
                                  n_floats, (hipStream_t)stream);
}

}  // extern "C"

// ------------------------------------------------------------ pipelined host streaming
// A GNU Radio scheduler hands a block a few thousand items per work() call
// (gr-simplefe/lib/sink_c_impl.cc:157-174, source_c_impl.cc:134-153); one synchronous H2D ->
// kernel -> D2H round trip per call is launch/sync bound (27 us per 3841 samples).  The pipe
// collects pushed items in pinned batches and keeps up to SFE_PIPE_SLOTS batches in flight on
// three streams (copy in / filter / copy out overlap, PCIe is full duplex); pull hands out finished
// items in order.  Sample alignment is untouched: item k out is the filter's output for item k in.
#include <immintrin.h>
namespace sfe {
// Host copies into / out of the pinned batches are what bounds the pipe (the GPU side of a batch
// is ~30 us, the two copies ~60): stream them past the cache -- the pinned side is touched next by
// the DMA engine, not by this core.  Falls back to memcpy without AVX2 or for small / odd pieces.
__attribute__((target("avx2"))) static void copy_stream_avx2(char *dst, const char *src, size_t n)
{
    while (n && (reinterpret_cast<uintptr_t>(dst) & 31u)) {
        *dst++ = *src++;
        n--;
    }
    for (; n >= 128; n -= 128, dst += 128, src += 128) {
        const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src));
        const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + 32));
        const __m256i c = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + 64));
        const __m256i d = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + 96));
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst), a);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + 32), b);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + 64), c);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + 96), d);
    }
    _mm_sfence();
    if (n) memcpy(dst, src, n);
}
static void copy_stream(void *dst, const void *src, size_t n)
{
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2 && n >= 4096) copy_stream_avx2(static_cast<char *>(dst), static_cast<const char *>(src), n);
    else memcpy(dst, src, n);
}

constexpr int PIPE_SLOTS = 4;
struct FirPipe {
    uint32_t magic = 0x50495031u;   // 'PIP1'
    Fir *f = nullptr;               // the filter behind the pipe ...
    void *rs = nullptr;             // ... or the resampler / decimator (sfe_rs_t), at `rate`
    float rate = 1.0f;
    int device = 0;
    size_t batch = 0, out_cap = 0, in_e = 0, out_e = 0;     // batch: input items per slot; out_cap: output items a slot can hold
    size_t quantum = 1;             // a partly filled batch sent on its way early is cut on a multiple of this (rs pipes at a
                                    // non-integer step: blksize, so the cut falls where a reference call ends; else 1)
    size_t tx_gs = 0;               // 10-bit packed output: samples per 5-byte group (2 complex / 4 real); an output ITEM is one group.
                                    // Cuts then fall on whole groups ONLY (less than a group is never sent: the converter emits whole groups)
    struct Slot {
        char *h_in = nullptr, *h_out = nullptr;
        void *d_in = nullptr, *d_out = nullptr;
        size_t n = 0;               // items submitted in this slot
        size_t n_out = 0;           // items it produces (== n behind a filter)
        hipEvent_t ev_in = nullptr, ev_k = nullptr, ev_out = nullptr;
        bool busy = false;          // submitted and not yet fully pulled
    } slot[PIPE_SLOTS];
    hipStream_t s_in = nullptr, s_k = nullptr, s_out = nullptr;
    int head = 0;                   // slot being filled
    size_t fill = 0;                // items in it
    int tail = 0;                   // oldest busy slot
    size_t out_off = 0;             // items already pulled from it
    bool tail_ready = false;        // its ev_out has been seen complete
};

static FirPipe *as_pipe(void *h)
{
    FirPipe *p = static_cast<FirPipe *>(h);
    if (p && p->magic != 0x50495031u) {
        set_error("not a live pipe handle");
        return nullptr;
    }
    return p;
}

static void pipe_free(FirPipe *p)
{
    if (!p) return;
    p->magic = 0;
    DeviceGuard g(p->device);
    for (auto &sl : p->slot) {
        if (sl.h_in) (void)hipHostFree(sl.h_in);
        if (sl.h_out) (void)hipHostFree(sl.h_out);
        if (sl.d_in) (void)hipFree(sl.d_in);
        if (sl.d_out) (void)hipFree(sl.d_out);
        if (sl.ev_in) (void)hipEventDestroy(sl.ev_in);
        if (sl.ev_k) (void)hipEventDestroy(sl.ev_k);
        if (sl.ev_out) (void)hipEventDestroy(sl.ev_out);
    }
    if (p->s_in) (void)hipStreamDestroy(p->s_in);
    if (p->s_k) (void)hipStreamDestroy(p->s_k);
    if (p->s_out) (void)hipStreamDestroy(p->s_out);
    delete p;
}

// submits the first `count` items of the batch being filled (all of it when count == fill); what is left
// moves to the front of the next slot (free: a partial submit only happens when nothing is in flight)
static int pipe_submit(FirPipe *p, size_t count)
{
    FirPipe::Slot &sl = p->slot[p->head];
    const size_t rem = p->fill - count;
    sl.n = count;
    SFE_HIP(hipMemcpyAsync(sl.d_in, sl.h_in, sl.n * p->in_e, hipMemcpyHostToDevice, p->s_in));
    SFE_HIP(hipEventRecord(sl.ev_in, p->s_in));
    SFE_HIP(hipStreamWaitEvent(p->s_k, sl.ev_in, 0));
    // a slot's device buffers are reused PIPE_SLOTS batches later: by then its copy-out has been
    // waited for (the slot was pulled), so no further ordering is needed on s_k
    int rc;
    if (p->f) {
        rc = fir_run(p->f, sl.d_in, sl.d_out, sl.n, sl.n, sl.n, p->s_k);
        sl.n_out = p->tx_gs ? sl.n / p->tx_gs : sl.n;
    } else {
        // the output count is known on the host as soon as the launch is made (closed form for
        // integer-valued steps, the replayed float32 recurrence otherwise): the copy-out is sized by it
        rc = sfe_dsp_rs_process_stream(p->rs, sl.d_in, sl.n, sl.n, sl.d_out, p->out_cap, p->out_cap, p->rate, &sl.n_out, p->s_k);
    }
    if (rc != SFE_OK) return rc;
    SFE_HIP(hipEventRecord(sl.ev_k, p->s_k));
    SFE_HIP(hipStreamWaitEvent(p->s_out, sl.ev_k, 0));
    if (sl.n_out) SFE_HIP(hipMemcpyAsync(sl.h_out, sl.d_out, sl.n_out * p->out_e, hipMemcpyDeviceToHost, p->s_out));
    SFE_HIP(hipEventRecord(sl.ev_out, p->s_out));
    sl.busy = true;
    p->head = (p->head + 1) % PIPE_SLOTS;
    if (rem) memcpy(p->slot[p->head].h_in, sl.h_in + count * p->in_e, rem * p->in_e);
    p->fill = rem;
    return SFE_OK;
}
}  // namespace sfe

extern "C" {

static int pipe_alloc(FirPipe *p, sfe_pipe_t *out)
{
    auto fail = [&](hipError_t e, const char *what) { int rc = hip_fail(e, what); pipe_free(p); return rc; };
#define TRY(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return fail(e__, #call); } while (0)
    TRY(hipStreamCreateWithFlags(&p->s_in, hipStreamNonBlocking));
    TRY(hipStreamCreateWithFlags(&p->s_k, hipStreamNonBlocking));
    TRY(hipStreamCreateWithFlags(&p->s_out, hipStreamNonBlocking));
    for (auto &sl : p->slot) {
        TRY(hipHostMalloc((void **)&sl.h_in, p->batch * p->in_e));
        TRY(hipHostMalloc((void **)&sl.h_out, p->out_cap * p->out_e));
        TRY(hipMalloc(&sl.d_in, p->batch * p->in_e));
        TRY(hipMalloc(&sl.d_out, p->out_cap * p->out_e));
        TRY(hipEventCreateWithFlags(&sl.ev_in, hipEventDisableTiming));
        TRY(hipEventCreateWithFlags(&sl.ev_k, hipEventDisableTiming));
        TRY(hipEventCreateWithFlags(&sl.ev_out, hipEventDisableTiming));
    }
#undef TRY
    *out = p;
    return SFE_OK;
}

int sfe_dsp_fir_pipe_create(sfe_fir_t fir, size_t batch_items, sfe_pipe_t *out)
{
    if (!out) return SFE_EINVAL;
    *out = nullptr;
    Fir *f = as_fir(fir);
    if (!f || f->n_channels != 1) {
        set_error("fir_pipe_create: needs a single-channel FIR handle");
        return SFE_EINVAL;
    }
    if (batch_items == 0) batch_items = (size_t)1 << 18;
    batch_items = (batch_items + 3) & ~(size_t)3;          // whole 10-bit groups per batch, whatever the output format
    if (batch_items < 256 || batch_items > ((size_t)1 << 26)) {
        set_error("fir_pipe_create: batch of %zu items out of range (256 .. 2^26)", batch_items);
        return SFE_EINVAL;
    }
    SFE_ON_DEVICE(f->device);
    FirPipe *p = new (std::nothrow) FirPipe;
    if (!p) return SFE_ENOMEM;
    p->f = f;
    p->device = f->device;
    p->batch = p->out_cap = batch_items;
    // items in: float32 samples, or the u8 wire format when the handle converts on load (SFE_FMT_U8:
    // 2 bytes per complex item, 1 per real one -- a receive chain hands the device's bytes straight in)
    p->in_e = f->in_u8 ? (f->data_complex ? 2 : 1) : (f->data_complex ? 8 : 4);
    p->out_e = f->out_complex ? 8 : 4;
    if (f->out_tx10) {
        // the transmit wire format out (sink_c_impl.cc:118-144 / sink_f_impl.cc:117-143): an output item is one 5-byte group
        p->tx_gs = f->out_complex ? 2 : 4;
        p->quantum = p->tx_gs;
        p->out_e = 5;
        p->out_cap = p->batch / p->tx_gs;
    }
    const int rc = pipe_alloc(p, out);
    if (rc == SFE_OK) f->piped++;          // the handle's formats are frozen while the pipe lives
    return rc;
}

int sfe_dsp_rs_pipe_create(sfe_rs_t rs, size_t batch_items, float rate, sfe_pipe_t *out)
{
    if (!out) return SFE_EINVAL;
    *out = nullptr;
    Rs *r = as_rs(rs);
    if (!r || r->n_channels != 1) {
        set_error("rs_pipe_create: needs a single-channel resample/decimate handle");
        return SFE_EINVAL;
    }
    if (r->mode == SFE_RS_RESAMPLE ? (rate < 1.0 / r->U) : (rate < 1.0)) {
        set_error("rs_pipe_create: rate %g not accepted by this mode", (double)rate);
        return SFE_EINVAL;
    }
    if (batch_items == 0) batch_items = (size_t)1 << 18;
    // whole reference calls per batch: for a non-integer step the result depends on where the
    // blksize-sample calls fall (resample.cxx:85-153), and they must fall where they would without the pipe
    batch_items = (batch_items + (size_t)r->blksize - 1) / (size_t)r->blksize * (size_t)r->blksize;
    if (batch_items < 256 || batch_items > ((size_t)1 << 26)) {
        set_error("rs_pipe_create: batch of %zu items out of range (256 .. 2^26)", batch_items);
        return SFE_EINVAL;
    }
    SFE_ON_DEVICE(r->device);
    FirPipe *p = new (std::nothrow) FirPipe;
    if (!p) return SFE_ENOMEM;
    p->rs = rs;
    p->rate = rate;
    p->device = r->device;
    p->batch = batch_items;
    p->out_cap = (size_t)ceil((double)batch_items / (double)rate) + 8;
    p->out_e = (size_t)r->esz();
    p->in_e = r->in_u8 ? (r->data_complex ? 2 : 1) : p->out_e;      // u8 wire-format items in (integer-valued steps)
    {
        // an integer-valued step gives the same items wherever the calls are cut; any other step does not
        const float stepf = rate * (float)r->U;
        p->quantum = (stepf >= 1.0f && stepf == floorf(stepf)) ? 1 : (size_t)r->blksize;
    }
    const int rc = pipe_alloc(p, out);
    if (rc == SFE_OK) r->piped++;
    return rc;
}

int sfe_dsp_pipe_push(sfe_pipe_t h, const void *in, size_t n_items, size_t *n_taken)
{
    FirPipe *p = as_pipe(h);
    if (!p || !n_taken || (n_items && !in)) return SFE_EINVAL;
    *n_taken = 0;
    SFE_ON_DEVICE(p->device);
    const char *src = static_cast<const char *>(in);
    while (n_items) {
        FirPipe::Slot &sl = p->slot[p->head];
        if (sl.busy) break;                                   // every slot in flight: pull first (backpressure)
        size_t m = p->batch - p->fill;
        if (m > n_items) m = n_items;
        copy_stream(sl.h_in + p->fill * p->in_e, src, m * p->in_e);
        p->fill += m;
        src += m * p->in_e;
        n_items -= m;
        *n_taken += m;
        if (p->fill == p->batch) {
            int rc = pipe_submit(p, p->fill);
            if (rc != SFE_OK) return rc;
        }
    }
    return SFE_OK;
}

// The oldest finished items, in place: *ptr / *count describe what is left of the oldest batch in flight once its
// copy-out has completed (count 0: nothing is ready).  wait: 0 = never block, 1 = block for that batch, 2 = also send a
// partly filled batch on its way when nothing else is in flight (end of stream / drain) and block for it.
static int pipe_front(FirPipe *p, int wait, const char **ptr, size_t *count)
{
    *ptr = nullptr;
    *count = 0;
    for (;;) {
        FirPipe::Slot &sl = p->slot[p->tail];
        if (!sl.busy) {
            if (wait == 2 && p->fill > 0 && p->tail == p->head) {
                // whole reference calls first: the remainder -- less than one call -- goes out only when
                // it is all there is, as the short last call a reference caller would make
                const size_t whole = p->fill / p->quantum * p->quantum;
                if (!whole && p->tx_gs) return SFE_OK;   // less than one 10-bit group: nothing the converter would emit
                int rc = pipe_submit(p, whole ? whole : p->fill);
                if (rc != SFE_OK) return rc;
                continue;
            }
            return SFE_OK;
        }
        if (!p->tail_ready) {
            if (wait) {
                SFE_HIP(hipEventSynchronize(sl.ev_out));
            } else {
                hipError_t e = hipEventQuery(sl.ev_out);
                if (e == hipErrorNotReady) return SFE_OK;
                if (e != hipSuccess) return hip_fail(e, "hipEventQuery");
            }
            p->tail_ready = true;
        }
        if (sl.n_out == p->out_off) {        // a batch that produced nothing (a decimator fed less than one step): retire it
            sl.busy = false;
            p->tail = (p->tail + 1) % PIPE_SLOTS;
            p->out_off = 0;
            p->tail_ready = false;
            continue;
        }
        *ptr = sl.h_out + p->out_off * p->out_e;
        *count = sl.n_out - p->out_off;
        return SFE_OK;
    }
}

// `m` of the items pipe_front described have been consumed
static void pipe_advance(FirPipe *p, size_t m)
{
    FirPipe::Slot &sl = p->slot[p->tail];
    p->out_off += m;
    if (p->out_off == sl.n_out) {
        sl.busy = false;
        p->tail = (p->tail + 1) % PIPE_SLOTS;
        p->out_off = 0;
        p->tail_ready = false;
    }
}

int sfe_dsp_pipe_pull(sfe_pipe_t h, void *out, size_t max_items, int wait, size_t *n_got)
{
    FirPipe *p = as_pipe(h);
    if (!p || !n_got || (max_items && !out)) return SFE_EINVAL;
    *n_got = 0;
    SFE_ON_DEVICE(p->device);
    char *dst = static_cast<char *>(out);
    int w = wait;                        // wait == 1 blocks for the oldest batch only, wait == 2 for all of them
    while (max_items) {
        const char *src;
        size_t m;
        int rc = pipe_front(p, w, &src, &m);
        if (rc != SFE_OK) return rc;
        if (!m) break;
        if (wait == 1) w = 0;
        if (m > max_items) m = max_items;
        copy_stream(dst, src, m * p->out_e);
        dst += m * p->out_e;
        max_items -= m;
        *n_got += m;
        pipe_advance(p, m);
    }
    return SFE_OK;
}

int sfe_dsp_pipe_acquire(sfe_pipe_t h, void **buf, size_t *room_items)
{
    FirPipe *p = as_pipe(h);
    if (!p || !buf || !room_items) return SFE_EINVAL;
    *buf = nullptr;
    *room_items = 0;
    FirPipe::Slot &sl = p->slot[p->head];
    if (sl.busy) return SFE_OK;          // every batch in flight: take finished items out first
    *buf = sl.h_in + p->fill * p->in_e;
    *room_items = p->batch - p->fill;
    return SFE_OK;
}

int sfe_dsp_pipe_commit(sfe_pipe_t h, size_t n_items)
{
    FirPipe *p = as_pipe(h);
    if (!p) return SFE_EINVAL;
    if (p->slot[p->head].busy ? n_items != 0 : n_items > p->batch - p->fill) {
        set_error("pipe_commit: %zu items exceed the room the last acquire reported", n_items);
        return SFE_EINVAL;
    }
    SFE_ON_DEVICE(p->device);
    p->fill += n_items;
    if (p->fill == p->batch) return pipe_submit(p, p->fill);
    return SFE_OK;
}

int sfe_dsp_pipe_peek(sfe_pipe_t h, const void **out, size_t *n_items, int wait)
{
    FirPipe *p = as_pipe(h);
    if (!p || !out || !n_items) return SFE_EINVAL;
    SFE_ON_DEVICE(p->device);
    const char *src;
    int rc = pipe_front(p, wait, &src, n_items);
    *out = src;
    return rc;
}

int sfe_dsp_pipe_release(sfe_pipe_t h, size_t n_items)
{
    FirPipe *p = as_pipe(h);
    if (!p) return SFE_EINVAL;
    FirPipe::Slot &sl = p->slot[p->tail];
    if (n_items && (!sl.busy || !p->tail_ready || n_items > sl.n_out - p->out_off)) {
        set_error("pipe_release: %zu items exceed what the last peek reported", n_items);
        return SFE_EINVAL;
    }
    if (n_items) pipe_advance(p, n_items);
    return SFE_OK;
}

int sfe_dsp_pipe_pending(sfe_pipe_t h, size_t *items)
{
    FirPipe *p = as_pipe(h);
    if (!p || !items) return SFE_EINVAL;
    size_t n = p->tx_gs ? p->fill / p->tx_gs : p->fill;         // (10-bit output: in groups, the unit pull hands out)
    for (int i = 0; i < PIPE_SLOTS; i++)
        if (p->slot[i].busy) n += p->slot[i].n_out - (i == p->tail ? p->out_off : 0);
    *items = n;
    return SFE_OK;
}

int sfe_dsp_pipe_destroy(sfe_pipe_t h)
{
    FirPipe *p = as_pipe(h);
    if (!p) return SFE_OK;
    {
        DeviceGuard g(p->device);
        (void)hipStreamSynchronize(p->s_in);
        (void)hipStreamSynchronize(p->s_k);
        (void)hipStreamSynchronize(p->s_out);
    }
    // the handle is alive: its destroy call refuses while a pipe borrows it
    if (p->f && p->f->piped > 0) p->f->piped--;
    if (p->rs && static_cast<Rs *>(p->rs)->piped > 0) static_cast<Rs *>(p->rs)->piped--;
    pipe_free(p);
    return SFE_OK;
}

}  // extern "C"
