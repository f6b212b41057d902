// api_plans.hip -- folding (U, step, pos0) into what the polyphase kernels read: zero-padded tap rows (poly_tiled_kernel,
// poly_rt_kernel), sub-filter spectra (poly_fft256_kernel), f32-MFMA fragments (poly_mfma_kernel).  Host code only.
#include "host.h"

namespace sfe {

// ---- tiled polyphase plans (common.h: PolyTiledPlan) ---------------------------------
// Fold (U, step, pos0) into zero-padded per-output-phase tap rows of equal length.
//   taps_pm: [U][plen] phase-major host taps.
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
const PolyTiledPlan *get_tiled_plan(PlanCache &cache, const std::vector<float> &taps_pm, int U, int plen,
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
    // (round 5: 9 ... 256 outputs per period -- 10/9, 25/24, x32, 147/160 -- have a tiled form too, poly_rt_dma.hip's poly_rt_dma_many_kernel; where its
    // launcher declines -- exact mode, unaligned channels -- the caller runs the generic kernel)
    const bool many = pl.UP > 8 && pl.UP <= 256 && pl.SP >= 1 && pl.SP <= 256 && pl.Lp > 0;      // (147/160, 160/147: 44.1 <-> 48 kHz)
    if (!poly_tiled_supported(pl.SP, pl.UP, pl.Lp) && !many) {
        cache.plans[key] = pl;          // d_G == nullptr marks "unsupported"
        return nullptr;
    }
    hipError_t e = hipMalloc(&pl.d_G, f.G.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(pl.d_G, f.G.data(), f.G.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess && pl.UP <= 256) {
        // the same taps transposed, [local time][phase] padded to whole groups of 8 phases (the runtime-shape kernels read a row per tap)
        pl.gt_pitch = (pl.UP + 7) / 8 * 8;
        std::vector<float> gt((size_t)pl.Lp * pl.gt_pitch, 0.0f);
        for (int r = 0; r < pl.UP; r++)
            for (int q = 0; q < pl.Lp; q++) gt[(size_t)q * pl.gt_pitch + r] = f.G[(size_t)r * pl.Lp + q];
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

// nullptr when the shape is not worth (or not instantiated for) the transform-domain kernel
// fft_mode: 0 = choose by the calibrated rule, 1 = always when instantiated, -1 = never
const PolyFftPlan *get_fft_plan(FftPlanCache &cache, const std::vector<float> &taps_pm, int U, int plen,
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
    // the 256 roots once (the same arguments as before, so the same spectra): the plan of one phase offset used to cost ~1.3 ms of cos / sin -- a stream cut
    // into equal bulk calls meets up to `step` offsets, each built on first use (profiles/r05/speed_sweep.txt)
    struct Roots {
        double c[256], s[256];
        Roots() { for (int k = 0; k < 256; k++) { c[k] = cos(-2.0 * M_PI / 256 * (double)k); s[k] = sin(-2.0 * M_PI / 256 * (double)k); } }
    };
    static const Roots roots;
    for (int r = 0; r < UP; r++)
        for (int cp = 0; cp < SP; cp++) {
            const int c = SP - 1 - cp;
            for (int b = 0; b < M; b++) {
                double re = 0.0, im = 0.0;
                for (int i = 0; i < pl.Li; i++) {
                    const double h = f.G[(size_t)r * f.Lp + (f.Lp - 1 - SP * i - c)];
                    const int k = (b * i) % M;
                    re += h * roots.c[k];
                    im += h * roots.s[k];
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

const PolyMfmaPlan *get_mfma_plan(MfmaCache &cache, const std::vector<float> &taps_pm, int U, int plen,
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

}  // namespace sfe
