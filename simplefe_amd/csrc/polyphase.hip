// polyphase.hip -- time-domain kernels: the polyphase dot product behind resample/decimate
// and the direct-form FIR (gfx950).
//
//   s(p) = sum_{j < plen} taps[(p mod U) + j*U] * x[floor(p/U) - j]
// is libdsp/decimate.cxx:132-140 (get_sample) and, value for value, the m_out[phase][n] the
// resample class precomputes (libdsp/resample.cxx:100-114).  Accumulation runs over ascending
// j from 0.0f exactly as the reference does; EXACT kernels keep multiply and add separate
// (the reference is compiled without FMA contraction), the others fuse them.
//
//   poly_int_kernel   : integer-valued step (mu == 0): out[k] = s(pos0 + k*step).
//                       Covers decimate /8 (U=1, step 8), resample 5/3 (U=3, step 5) and the
//                       direct FIR (U=1, step 1).  Input tile + taps staged in LDS.
//   poly_sched_kernel : general rate: out[k] = s(P_k)*(1-mu_k) + mu_k*s(P_k+1) with (P_k, mu_k)
//                       replayed on the host from the reference's float32 recurrence
//                       (libdsp/resample.cxx:119-150).
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

#include "common.h"

// hipcc contracts a*b+c into an FMA by default (and HIP's __fmul_rn/__fadd_rn are plain * and +
// defined in a header, so they contract too).  This file is compiled with -ffp-contract=off
// (simplefe_amd/build.py: EXACT_SOURCES) so the EXACT kernels round every product, as the
// reference built without FMA does; the fast kernels ask for the FMA explicitly.
#pragma clang fp contract(off)

namespace sfe {
namespace {

__device__ __forceinline__ long long floordiv(long long a, int b)
{
    long long q = a / b;
    return (a % b != 0 && a < 0) ? q - 1 : q;
}

template <bool CPLX> struct Elem;
template <> struct Elem<true> {
    typedef v2f T;
    static __device__ __forceinline__ T zero() { return (v2f){0.0f, 0.0f}; }
};
template <> struct Elem<false> {
    typedef float T;
    static __device__ __forceinline__ T zero() { return 0.0f; }
};

template <bool EXACT>
__device__ __forceinline__ float mac(float acc, float t, float x)
{
    if constexpr (EXACT) { const float p = t * x; return acc + p; }
    else return __builtin_fmaf(t, x, acc);
}
template <bool EXACT>
__device__ __forceinline__ v2f mac(v2f acc, float t, v2f x)
{
    if constexpr (EXACT) { const v2f p = (v2f){t, t} * x; return acc + p; }
    else return __builtin_elementwise_fma((v2f){t, t}, x, acc);
}

// virtual stream: history (hl samples) followed by this call's input; zero outside
template <bool CPLX>
__device__ __forceinline__ typename Elem<CPLX>::T vload(const typename Elem<CPLX>::T *in,
                                                        const typename Elem<CPLX>::T *hist,
                                                        long long i, long long n_in, int hl)
{
    if (i >= 0) return i < n_in ? in[i] : Elem<CPLX>::zero();
    return (i >= -(long long)hl) ? hist[hl + i] : Elem<CPLX>::zero();
}

// same virtual stream when the input is u8 offset binary (history is float32 already)
template <bool CPLX>
__device__ __forceinline__ typename Elem<CPLX>::T vload_u8(const unsigned char *in, const typename Elem<CPLX>::T *hist,
                                                           long long i, long long n_in, int hl)
{
    if (i >= 0) {
        if (i >= n_in) return Elem<CPLX>::zero();
        if constexpr (CPLX) return (v2f){u8_to_f32(in[2 * i]), u8_to_f32(in[2 * i + 1])};
        else return u8_to_f32(in[i]);
    }
    return (i >= -(long long)hl) ? hist[hl + i] : Elem<CPLX>::zero();
}

// ------------------------------------------------------------------ integer-step law
// One workgroup = TILE consecutive outputs.  LDS: [tile_in_cap] samples + [U*plen] taps.
// sparse (round 5; the launcher sets it when step >= U * plen: every output's plen samples are its own): the LDS holds the tile's WINDOWS
// back to back -- plen samples per output -- instead of the whole span between the first and the last, which at a step of 128 or 1000
// neither fits nor is needed (until round 5 the bulk call REFUSED steps beyond ~117: "step too large for the LDS tile"; the
// reference's decimate takes any rate >= 1, libdsp/decimate.cxx:75-78).  Same sums in the same order.
template <bool CPLX, bool EXACT>
__global__ __launch_bounds__(256) void poly_int_kernel(PolyArgs a, int tile_out, int tile_in_cap, int sparse, int taps_global)
{
    typedef typename Elem<CPLX>::T T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T *xs = reinterpret_cast<T *>(smem);
    float *ts = reinterpret_cast<float *>(smem + (size_t)tile_in_cap * sizeof(T));

    const int ch = blockIdx.y;
    const T *in = static_cast<const T *>(a.in) + (size_t)ch * a.in_stride;
    const T *hist = static_cast<const T *>(a.hist) + (size_t)ch * a.hl;
    T *out = static_cast<T *>(a.out) + (size_t)ch * a.out_stride;

    const long long k0 = (long long)blockIdx.x * tile_out;
    long long k1 = k0 + tile_out;
    if (k1 > a.n_out) k1 = a.n_out;
    const long long p_first = a.pos0 + k0 * a.step;
    const long long p_last = a.pos0 + (k1 - 1) * a.step;
    const long long n_lo = floordiv(p_first, a.U) - (a.plen - 1);
    const long long n_hi = floordiv(p_last, a.U);
    const int tile_len = (int)(n_hi - n_lo + 1);

    if (sparse) {
        const unsigned cnt = (unsigned)(k1 - k0) * (unsigned)a.plen;
        for (unsigned idx = threadIdx.x; idx < cnt; idx += 256u) {
            const unsigned w = idx / (unsigned)a.plen, j = idx - w * (unsigned)a.plen;        // window w, its j-th sample in time order
            const long long n = floordiv(a.pos0 + (k0 + w) * a.step, a.U);
            xs[idx] = vload<CPLX>(in, hist, n - (a.plen - 1) + j, a.n_in, a.hl);
        }
    } else if (n_lo >= 0 && n_lo + tile_len <= a.n_in) {
        // an interior tile: eight requests per thread in flight at a time (the guarded loop below is one load -> wait ->
        // LDS write per iteration, one memory latency after the other; poly_seg_kernel has the measurement)
        const T *src = in + n_lo;
        for (int i0 = (int)threadIdx.x; i0 < tile_len; i0 += 256 * 8) {
            T v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int i = i0 + 256 * u;
                v[u] = i < tile_len ? __builtin_nontemporal_load(src + i) : Elem<CPLX>::zero();
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int i = i0 + 256 * u;
                if (i < tile_len) xs[i] = v[u];
            }
        }
    } else {
        for (int i = threadIdx.x; i < tile_len; i += 256) xs[i] = vload<CPLX>(in, hist, n_lo + i, a.n_in, a.hl);
    }
    // the taps TRANSPOSED, ts[j U + phase]: consecutive outputs sit in different phases (phase = (pos0 + k step) mod U), and phase-major rows
    // plen floats apart put a wave's reads of tap j into ONE bank -- 10/9, 16/15, 25/24 ran at 0.05-0.10 of the roofline until round 5
    // (taps_global: more taps than the LDS holds beside a tile -- 160 phases of 127 -- stay where they are, phase-major in memory, and are read
    // through the caches; until round 5 such a filter was refused: "taps do not fit the LDS tile")
    if (!taps_global)
        for (int i = threadIdx.x; i < a.U * a.plen; i += 256) {
            const int ph = i / a.plen, j = i - ph * a.plen;
            ts[j * a.U + ph] = a.taps[i];
        }
    __syncthreads();

    for (long long k = k0 + threadIdx.x; k < k1; k += 256) {
        const long long p = a.pos0 + k * a.step;
        const long long n = floordiv(p, a.U);
        const int ph = (int)(p - n * a.U);
        const float *tp = taps_global ? a.taps + (size_t)ph * a.plen : ts + ph;
        const int tstr = taps_global ? 1 : a.U;
        const T *xp = sparse ? xs + ((k - k0) * a.plen + (a.plen - 1)) : xs + (n - n_lo);      // the sample at time n
        T acc = Elem<CPLX>::zero();
        for (int j = 0; j < a.plen; j++) acc = mac<EXACT>(acc, tp[j * tstr], xp[-j]);
        out[k] = acc;
    }
}

// ------------------------------------------------------ integer-step law, tiled (fast path)
// One workgroup = TM consecutive m (TM*UP outputs).  The input span is staged once into LDS,
// de-interleaved by SP:  X[p][c] = x[n_org + SP*c + p], so that for a fixed tap the lanes of a
// wave (consecutive m) read consecutive cells of one row, conflict-free.  Each thread owns TWO
// consecutive m (2i, 2i+1): the cells it needs for them at tap chunk c are X[p][2i+c] and
// X[p][2i+c+1], so one aligned 16-byte LDS read (a cell pair at an even column) feeds four
// multiply-accumulates across two chunks -- 1/4 of the LDS instructions of a read per tap.
// Taps are wave-uniform (SGPR loads); UP accumulators per m share every sample read (resample
// 5/3: one read feeds up to three phase sums).  The loop runs q = Lp-1 .. 0, i.e. tap index
// ascending: the reference's accumulation order (libdsp/decimate.cxx:134-137).
constexpr int TM = 512;                 // m per workgroup (2 consecutive per thread)
__host__ __device__ constexpr int tiled_xc(int SP) { return SP >= 4 ? 64 : (SP >= 2 ? 256 : 1024); }
__host__ __device__ constexpr int tiled_rowlen(int SP) { return TM + tiled_xc(SP) + 2; }   // even, == 2 mod 16

template <bool CPLX> struct Pair;
template <> struct Pair<true> { typedef v4f P; };
template <> struct Pair<false> { typedef v2f P; };
__device__ __forceinline__ v2f pair_lo(v4f p) { return (v2f){p.x, p.y}; }
__device__ __forceinline__ v2f pair_hi(v4f p) { return (v2f){p.z, p.w}; }
__device__ __forceinline__ float pair_lo(v2f p) { return p.x; }
__device__ __forceinline__ float pair_hi(v2f p) { return p.y; }

// IN_U8: the input is the device wire format (u8 offset binary, gr-simplefe source blocks) and
// is converted while it is staged -- 2 bytes instead of 8 per complex sample from HBM.
// DIAG (instantiated under -DSFE_DIAG only, SFE_TILED_DIAG): bit 0 = no dot products (the staged tile is
// still written and read once), bit 1 = no staging through LDS either (loads feed the result directly).
template <int SP, int UP, bool CPLX, bool EXACT, bool IN_U8 = false, int DIAG = 0>
__global__ __launch_bounds__(256) void poly_tiled_kernel(PolyTiledArgs a)
{
    typedef typename Elem<CPLX>::T T;
    typedef typename Pair<CPLX>::P P2;
    constexpr int ROWLEN = tiled_rowlen(SP);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T *X = reinterpret_cast<T *>(smem);

    const unsigned tid = threadIdx.x;
    const int ch = blockIdx.y;
    const T *in = static_cast<const T *>(a.in) + (IN_U8 ? 0 : (size_t)ch * a.in_stride);
    const unsigned char *in8 = static_cast<const unsigned char *>(a.in) + (size_t)ch * a.in_stride * (CPLX ? 2 : 1);
    const T *hist = static_cast<const T *>(a.hist) + (size_t)ch * a.hl;
    T *out = static_cast<T *>(a.out) + (size_t)ch * a.out_stride;

    // state carry-over fused in: ONE EXTRA workgroup per channel (blockIdx.x == the tile count) does
    // nothing but write the NEXT call's history, in[n_in - hl .. n_in) as float32, and leaves (the launcher adds it
    // only when n_in >= hl).  Done by tile 0's workgroup in front of its tile instead, the copy loop cost the
    // decimator 12 VGPRs -- 82 instead of 70, five waves per SIMD instead of seven, 1.59 -> 2.05 ms at 2^30.
    // (fused-arithmetic instantiations only: the exact-mode kernels -- the class-compatible path, a few thousand samples per
    // call -- are left as they were, two of them sit on a register / scalar-register step that the extra arguments cross)
    unsigned bx = blockIdx.x, nwork = a.tiles;
#ifdef SFE_DIAG
    if (a.win > 1u) {                            // `win` windows: see PolyTiledArgs (measured, not kept: DESIGN.md 4.2)
        const unsigned per = (a.tiles + a.win - 1u) / a.win;
        nwork = per * a.win;
        bx = (blockIdx.x % a.win) * per + blockIdx.x / a.win;
    }
#endif
    if (!EXACT && a.hist_out && blockIdx.x == nwork) {
        T *ho = static_cast<T *>(a.hist_out) + (size_t)ch * a.hl;
#pragma unroll 1
        for (unsigned i = tid; i < (unsigned)a.hl; i += 256u) {
            if constexpr (IN_U8) ho[i] = vload_u8<CPLX>(in8, hist, a.n_in - a.hl + i, a.n_in, a.hl);
            else ho[i] = in[a.n_in - a.hl + i];
        }
        return;
    }
    if (bx >= a.tiles) return;
    const long long m0 = (long long)bx * TM;
    const long long n_org = (long long)SP * m0 + a.e_max - (a.Lp - 1);   // stream index of local sample 0
    const int n_tile = SP * TM + a.Lp;
    // translation look-ahead (PolyTiledArgs::tlb_ahead; diagnostic library only -- measured 5-6 % slower): a one-lane read of the first input sample and the first output of the
    // tile `tlb_ahead` further on; what it returns is looked at after the tile's own stores, i.e. never waited for early
#ifdef SFE_DIAG
    float touch = 0.0f;
    if (!IN_U8 && a.tlb_ahead && tid == 0) {
        const long long ni = n_org + (long long)a.tlb_ahead * SP * TM, ko = (long long)UP * (m0 + (long long)a.tlb_ahead * TM);
        if (ni >= 0 && ni < a.n_in) touch = __builtin_nontemporal_load(reinterpret_cast<const float *>(in + ni));
        if (ko < a.n_out) touch += __builtin_nontemporal_load(reinterpret_cast<const float *>(out + ko));
    }
#endif
    constexpr int MAIN = SP * TM / 256;       // unrolled loads per thread for the body of the tile

    // ---- stage: coalesced 8-byte lanes in, transposed into the SP rows
    if constexpr (IN_U8) {
        // 8-byte lanes again, now 4 complex (or 8 real) samples each; the tile start is rounded
        // down to an 8-byte boundary of the byte stream and the extra samples skipped
        constexpr int SPL = CPLX ? 4 : 8;                              // samples per 8-byte lane
        const bool interior = n_org >= 0 && n_org + n_tile + SPL <= a.n_in;
        if (interior && (reinterpret_cast<uintptr_t>(in8) & 7) == 0) {
            const long long a0 = n_org & ~(long long)(SPL - 1);
            const int delta = (int)(n_org - a0);
            const unsigned long long *src = reinterpret_cast<const unsigned long long *>(in8 + a0 * (CPLX ? 2 : 1));
            const int nl = (n_tile + delta + SPL - 1) / SPL;
            for (int l = tid; l < nl; l += 256) {
                const unsigned long long w = __builtin_nontemporal_load(src + l);
#pragma unroll
                for (int q = 0; q < SPL; q++) {
                    const int s = l * SPL + q - delta;
                    if (s >= 0 && s < n_tile) {
                        T v;
                        if constexpr (CPLX) v = (v2f){u8_to_f32((unsigned)(w >> (16 * q)) & 0xFFu), u8_to_f32((unsigned)(w >> (16 * q + 8)) & 0xFFu)};
                        else v = u8_to_f32((unsigned)(w >> (8 * q)) & 0xFFu);
                        X[((unsigned)s % SP) * ROWLEN + (unsigned)s / SP] = v;
                    }
                }
            }
        } else {
            for (unsigned s = tid; s < (unsigned)n_tile; s += 256)
                X[(s % SP) * ROWLEN + s / SP] = vload_u8<CPLX>(in8, hist, n_org + s, a.n_in, a.hl);
        }
    } else if (n_org >= 0 && n_org + n_tile <= a.n_in) {
        const T *src = in + n_org;                                       // uniform
        T v[MAIN];
#pragma unroll
        for (int i = 0; i < MAIN; i++) v[i] = __builtin_nontemporal_load(src + tid + 256u * i);
        // cell of local sample s = tid + 256 i:  s = SP (q0 + c_i) + (r0 + d_i) with compile-time
        // c_i = 256 i / SP, d_i = 256 i % SP and r0 + d_i < 2 SP: one compare instead of a
        // division by SP per element.
        const unsigned q0 = tid / SP, r0 = tid % SP;
        const unsigned cell0 = r0 * ROWLEN + q0;
        if constexpr (DIAG & 2) {
            T sum = v[0];
#pragma unroll
            for (int i = 1; i < MAIN; i++) sum += v[i];
            X[tid] = sum;
        } else {
#pragma unroll
        for (int i = 0; i < MAIN; i++) {
            constexpr unsigned W = SP * ROWLEN - 1;         // row wrap: -SP rows, +1 column
            const unsigned ci = (256u * i) / SP, di = (256u * i) % SP;
            const unsigned cell = cell0 + di * ROWLEN + ci;
            X[(r0 + di >= (unsigned)SP) ? cell - W : cell] = v[i];
        }
        }
        for (unsigned s = SP * TM + tid; s < (unsigned)n_tile; s += 256)
            X[(s % SP) * ROWLEN + s / SP] = __builtin_nontemporal_load(src + s);
    } else {
        for (unsigned s = tid; s < (unsigned)n_tile; s += 256)
            X[(s % SP) * ROWLEN + s / SP] = vload<CPLX>(in, hist, n_org + s, a.n_in, a.hl);
    }
    __syncthreads();

    T acc[2][UP];
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
        for (int r = 0; r < UP; r++) acc[b][r] = Elem<CPLX>::zero();

    // chunk c covers local times q = c*SP + p.  One STEP = two chunks (cc+1, cc), cc even: it
    // loads the pair row at column 2*tid+cc into `cur` and uses `nxt` (column 2*tid+cc+2).
    // STEPs alternate two register sets so no pair is ever copied.
    const int nchunk = a.Lp / SP;                                   // even (host pads Lp)
    const P2 *xp = reinterpret_cast<const P2 *>(X) + tid + nchunk / 2;   // pair column (2*tid + cc)/2
    // DIAG bit 2: the taps through the constant address space (scalar loads) instead of the generic pointer (vector loads
    // with a uniform address) -- A/B in the diagnostic library
    typedef const __attribute__((address_space(4))) float *cfp;
    typename std::conditional<(DIAG & 4) != 0, cfp, const float *>::type g = (typename std::conditional<(DIAG & 4) != 0, cfp, const float *>::type)a.G + (size_t)(nchunk - 1) * SP;               // taps of chunk cc+1
    auto step = [&](P2 (&cur)[SP], const P2 (&nxt)[SP]) {
        xp -= 1;
#pragma unroll
        for (int p = 0; p < SP; p++) cur[p] = xp[p * (ROWLEN / 2)];
#pragma unroll
        for (int p = SP - 1; p >= 0; --p) {                          // chunk cc+1
#pragma unroll
            for (int r = 0; r < UP; r++) {
                const float t = g[r * a.Lp + p];                     // wave-uniform
                acc[0][r] = mac<EXACT>(acc[0][r], t, pair_hi(cur[p]));
                acc[1][r] = mac<EXACT>(acc[1][r], t, pair_lo(nxt[p]));
            }
        }
#pragma unroll
        for (int p = SP - 1; p >= 0; --p) {                          // chunk cc
#pragma unroll
            for (int r = 0; r < UP; r++) {
                const float t = g[r * a.Lp + p - SP];
                acc[0][r] = mac<EXACT>(acc[0][r], t, pair_lo(cur[p]));
                acc[1][r] = mac<EXACT>(acc[1][r], t, pair_hi(cur[p]));
            }
        }
        g -= 2 * SP;
    };
    P2 pa[SP], pb[SP];
#pragma unroll
    for (int p = 0; p < SP; p++) pa[p] = xp[p * (ROWLEN / 2)];      // pairs at column 2*tid + nchunk
    int steps = (DIAG & 1) ? 0 : nchunk / 2;
    if constexpr (DIAG & 1) {
#pragma unroll
        for (int r = 0; r < UP; r++) { acc[0][r] = pair_lo(pa[0]); acc[1][r] = pair_hi(pa[SP - 1]); }
    }
    if (steps & 1) {                                                 // odd count: peel one, landing in pa
        step(pb, pa);
#pragma unroll
        for (int p = 0; p < SP; p++) pa[p] = pb[p];
        steps--;
    }
    for (; steps > 0; steps -= 2) {
        step(pb, pa);
        step(pa, pb);
    }

    // ---- outputs UP*(m0 + 2*tid) .. + 2*UP - 1: contiguous per lane.
    // UP == 1: one 16-byte streaming store per lane, lanes contiguous.  UP > 1: a lane's 2*UP
    // results are 16*UP bytes apart from its neighbour's, so the tile is first laid out linearly
    // in LDS (the input tile is dead by now) and then stored with contiguous 16-byte lanes --
    // every wave instruction writes whole 128-byte lines (otherwise WRITE_SIZE read +16%).
    const long long k0 = (long long)UP * m0;                // first output of the tile
    const bool aligned = ((reinterpret_cast<uintptr_t>(out + k0) & 15) == 0);
    if (UP == 1 || !aligned || k0 + (long long)UP * TM > a.n_out) {
        const long long k = k0 + (long long)UP * 2 * tid;
        if (UP == 1 && aligned && k + 2 <= a.n_out) {
            P2 v;
            if constexpr (CPLX) v = (v4f){acc[0][0].x, acc[0][0].y, acc[1][0].x, acc[1][0].y};
            else v = (v2f){acc[0][0], acc[1][0]};
            __builtin_nontemporal_store(v, reinterpret_cast<P2 *>(out + k));
        } else {
#pragma unroll
            for (int j = 0; j < 2 * UP; j++)
                if (k + j < a.n_out) out[k + j] = acc[j / UP][j % UP];
        }
    } else {
        __syncthreads();                                    // everyone is done reading X
        T *Y = X;
#pragma unroll
        for (int j = 0; j < 2 * UP; j++) Y[2 * UP * tid + j] = acc[j / UP][j % UP];
        __syncthreads();
        const P2 *Yp = reinterpret_cast<const P2 *>(Y);
        P2 *op = reinterpret_cast<P2 *>(out + k0);
#pragma unroll
        for (int i = 0; i < UP; i++) __builtin_nontemporal_store(Yp[tid + 256 * i], op + tid + 256 * i);
    }
#ifdef SFE_DIAG
    if (!IN_U8 && a.tlb_ahead) asm volatile("" ::"v"(touch));      // the look-ahead reads end here
#endif
}

// ------------------------------------------- integer-step law, tiled, ANY (SP, UP): launch arguments
// poly_tiled_kernel exists for a handful of compile-time (SP, UP) pairs (eleven until the end of round 4, none of them with
// UP > SP: VERDICT r3 missing 4; since then also 2/3, 3/4 and 4/3):
// every interpolating ratio -- what `resample` accepts and `decimate` does not, libdsp/resample.cxx:91 against
// libdsp/decimate.cxx:75-78 -- and decimations such as 6, 7, 16 fell to poly_int_kernel (one output per thread,
// per-lane tap rows, samples read at stride `step`).  This kernel is the tiled form with SP and UP as
// ARGUMENTS: the same zero-padded rows G[UP][Lp] (api_plans.hip: fold_rows), the same LDS image -- the tile
// de-interleaved by a runtime SP, X[p][c] = x[n_org + SP c + p], row pitch chosen by the launcher so that the
// scatter spreads over the banks -- and per thread m = tid, tid + 256, ...: for a fixed tap the lanes of a wave
// read consecutive cells of one row (conflict-free), every sample read feeds all UP phase sums, taps come
// through the constant address space (scalar loads: they are wave-uniform).  Accumulation runs tap index
// ascending from 0.0f (the reference's order, libdsp/decimate.cxx:134-137): EXACT is bit-identical to the
// compiled reference, the default fuses multiply and add.  UPM = phase sums compiled in (= UP, 1..8).
// MB: m per thread that run TOGETHER through the tap loop (m, m + 256, ...): one tap row (a scalar load) and one loop
// step feed MB sample reads and MB x UP multiply-accumulates -- with one m at a time the loop is bound by the latency of
// the tap load and the LDS read (interpolate x2: 2.38 -> see profiles/r04/shapes.txt).
template <bool CPLX, bool EXACT, bool IN_U8, int UPM, int MB, int NLD>
__global__ __launch_bounds__(256) void poly_rt_kernel(PolyTiledArgs a)
{
    typedef typename Elem<CPLX>::T T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T *X = reinterpret_cast<T *>(smem);

    const unsigned tid = threadIdx.x;
    const int ch = blockIdx.y;
    const T *in = static_cast<const T *>(a.in) + (IN_U8 ? 0 : (size_t)ch * a.in_stride);
    const unsigned char *in8 = static_cast<const unsigned char *>(a.in) + (size_t)ch * a.in_stride * (CPLX ? 2 : 1);
    const T *hist = static_cast<const T *>(a.hist) + (size_t)ch * a.hl;
    T *out = static_cast<T *>(a.out) + (size_t)ch * a.out_stride;

    if (!EXACT && a.hist_out && blockIdx.x == a.tiles) {        // the history workgroup (poly_tiled_kernel has the reasoning)
        T *ho = static_cast<T *>(a.hist_out) + (size_t)ch * a.hl;
#pragma unroll 1
        for (unsigned i = tid; i < (unsigned)a.hl; i += 256u) {
            if constexpr (IN_U8) ho[i] = vload_u8<CPLX>(in8, hist, a.n_in - a.hl + i, a.n_in, a.hl);
            else ho[i] = in[a.n_in - a.hl + i];
        }
        return;
    }
    const unsigned SP = (unsigned)a.SP, RL = (unsigned)a.rowlen;
    const int UP = a.UP, TMr = a.tm;
    const long long m0 = (long long)blockIdx.x * TMr;
    const long long n_org = (long long)SP * m0 + a.e_max - (a.Lp - 1);   // stream index of local sample 0
    const unsigned n_tile = SP * (unsigned)TMr + (unsigned)a.Lp;
    // local sample s -> cell (s % SP) * RL + s / SP, the division as a multiply (exact: s < 2^16, SP <= 64;
    // SP = 1 has no 32-bit multiplier and needs none)
    auto cell_of = [&](unsigned s) -> unsigned {
        if (SP == 1u) return s;
        const unsigned q = __umulhi(s, a.sp_inv);
        return (s - q * SP) * RL + q;
    };
    if constexpr (IN_U8) {
        for (unsigned s = tid; s < n_tile; s += 256u) X[cell_of(s)] = vload_u8<CPLX>(in8, hist, n_org + s, a.n_in, a.hl);
    } else if (n_org >= 0 && n_org + n_tile <= a.n_in) {
        const T *src = in + n_org;                                        // uniform
        // The tile's requests go out before the first is looked at (eight at a time -- the first version -- left a
        // workgroup waiting for memory two or three times over per tile).  NLD rows of 256 are compiled in: 6 or 10 for tiles
        // of up to 1536 / 2560 samples (the interpolating shapes: their tile is bounded by its OUTPUT), 14, 16 or 18 beyond --
        // SP tm <= 4096 samples plus arms of up to 32 taps; what is left of a longer tile follows one request at a time.
        // (Eighteen for every shape cost the interpolators three waves of occupancy and a dozen redundant requests per
        // thread: x4 2.71 -> 3.14 ms.)  Lanes beyond the tile read its last sample and drop it: straight-line code, no
        // branch around a request (with a uniform guard per row the compiler took the requests apart again).
        T v[NLD];
#pragma unroll
        for (int u = 0; u < NLD; u++) {
            const unsigned i = tid + 256u * u;
            v[u] = __builtin_nontemporal_load(src + (i < n_tile ? i : n_tile - 1u));
        }
#pragma unroll
        for (int u = 0; u < NLD; u++) {
            const unsigned i = tid + 256u * u;
            if (i < n_tile) X[cell_of(i)] = v[u];
        }
        for (unsigned i = tid + 256u * NLD; i < n_tile; i += 256u) X[cell_of(i)] = __builtin_nontemporal_load(src + i);
    } else {
        for (unsigned s = tid; s < n_tile; s += 256u) X[cell_of(s)] = vload<CPLX>(in, hist, n_org + s, a.n_in, a.hl);
    }
    __syncthreads();

    // taps: row qt of Gt holds the UP phases' taps at local time qt (padded to 8 floats): ONE scalar load per tap
    const __attribute__((address_space(4))) float *gt = (const __attribute__((address_space(4))) float *)a.Gt;
    const int Lq = a.Lp / (int)SP;
    const bool out16 = CPLX && (UP % 2 == 0) && (reinterpret_cast<uintptr_t>(out) & 15u) == 0;     // a thread's UP outputs as 16-byte pairs
    const bool out_pairs = CPLX && (UP & 1) && UP == UPM && (reinterpret_cast<uintptr_t>(out) & 15u) == 0 && !(TMr & 1);      // odd UP: lane pairs (below)
    const bool out16w = CPLX && UP == UPM && a.y_off && (reinterpret_cast<uintptr_t>(out) & 15u) == 0;            // whole waves through LDS (below)
#pragma unroll 1
    for (int mi0 = (int)tid; mi0 < TMr; mi0 += 256 * MB) {
        T acc[MB][UPM];
        unsigned cj[MB];                                             // column of m_j relative to m_0 (a clamped one beyond the tile: not stored)
#pragma unroll
        for (int j = 0; j < MB; j++) {
            cj[j] = mi0 + 256 * j < TMr ? 256u * j : 0u;
#pragma unroll
            for (int r = 0; r < UPM; r++) acc[j][r] = Elem<CPLX>::zero();
        }
        // local time qt = SP qq + p descending = tap index ascending; sample (mi, qt) sits at row p, column mi + qq
        unsigned off = (SP - 1u) * RL + (unsigned)(Lq - 1) + (unsigned)mi0;
        unsigned p = SP - 1u;
        // (round 5 tried eight / four taps' reads in flight for one / two m per thread: no change beyond the run-to-run noise,
        // profiles/r05/shapes_tile_sizes.txt)
#pragma unroll 2
        for (int qt = a.Lp - 1; qt >= 0; --qt) {
            T x[MB];
#pragma unroll
            for (int j = 0; j < MB; j++) x[j] = X[off + cj[j]];
            float tp[UPM];
#pragma unroll
            for (int r = 0; r < UPM; r++) tp[r] = gt[8 * qt + r];       // consecutive scalar loads: merged into one s_load_dwordxN
#pragma unroll
            for (int j = 0; j < MB; j++)
#pragma unroll
                for (int r = 0; r < UPM; r++) acc[j][r] = mac<EXACT>(acc[j][r], tp[r], x[j]);     // phases beyond UP meet the table's zeros and are not stored
            if (p == 0u) {
                p = SP - 1u;
                off += (SP - 1u) * RL - 1u;                         // row SP - 1 of the column before
            } else {
                p--;
                off -= RL;
            }
        }
#pragma unroll
        for (int j = 0; j < MB; j++) {
            const int mi = mi0 + 256 * j;
            if constexpr (CPLX && UPM >= 3) {
                // Three or more outputs per m: a lane's UP results are 8 UP bytes from its neighbour's, every store instruction
                // of the forms below writes 16-byte pieces at that stride -- for the shapes that are all stores (x4: 32 of
                // every 40 bytes, x8: 64 of 72) the worst pattern there is (x8: 0.39 of the roofline).  The wave's 64 m are
                // 64 UP CONSECUTIVE outputs: laid out in a region of LDS of the wave's own (rows of UP + 1 cells: the writes
                // spread over the banks) and read back pair by pair, they leave as whole contiguous kilobytes.  Nothing but
                // the wave touches the region and a wave's LDS operations execute in order: no barrier.
                const int mi_w = ((int)(tid & ~63u)) + (mi0 - (int)tid) + 256 * j;       // the wave's first m of this round
                const long long kw = (long long)UP * (m0 + mi_w);
                if (out16w && mi_w + 64 <= TMr && kw + 64LL * UP <= a.n_out) {                // uniform over the wave
                    v2f *Yw = reinterpret_cast<v2f *>(smem + a.y_off) + (tid >> 6) * (64u * (UPM + 1));
                    const unsigned lane = tid & 63u;
#pragma unroll
                    for (int r = 0; r < UPM; r++) Yw[lane * (UPM + 1) + r] = acc[j][r];
#pragma unroll
                    for (int i = 0; i < (32 * UPM + 63) / 64; i++) {
                        const unsigned pi = lane + 64u * i;                              // pair of outputs 2 pi, 2 pi + 1
                        if ((32 * UPM) % 64 == 0 || pi < 32u * UPM) {
                            const unsigned o0 = 2u * pi, o1 = o0 + 1u;
                            const v2f lo2 = Yw[(o0 / UPM) * (UPM + 1) + o0 % UPM], hi2 = Yw[(o1 / UPM) * (UPM + 1) + o1 % UPM];
                            __builtin_nontemporal_store((v4f){lo2.x, lo2.y, hi2.x, hi2.y}, reinterpret_cast<v4f *>(out + kw + o0));
                        }
                    }
                    continue;
                }
            }
            if (mi >= TMr) continue;
            const long long k = (long long)UP * (m0 + mi);
            if constexpr (CPLX && (UPM == 3 || UPM == 5 || UPM == 7)) {
                // an ODD number of outputs per m: lanes 2i, 2i + 1 hold 2 UP consecutive outputs between them, starting on a
                // 16-byte boundary (m of the even lane is even).  The even lane takes the odd lane's first output beside its own
                // last one, so that both store aligned 16-byte pairs only -- (UP + 1) / 2 and (UP - 1) / 2 of them -- instead
                // of UP 8-byte pieces at a stride of 8 UP bytes each.  (tm is even and tiles start on even m.)
                const long long kp = (long long)UP * (m0 + (mi & ~1));          // the pair's first output: the same in both lanes
                const bool pair_ok = out_pairs && kp + 2 * UP <= a.n_out;
                const v2f first = acc[j][0];
                const v2f nxt0 = (v2f){__shfl_down(first.x, 1), __shfl_down(first.y, 1)};       // the odd lane's first output, in the even lane
                if (pair_ok) {
                    const bool odd = (tid & 1u) != 0;
                    constexpr int H = (UPM - 1) / 2;
#pragma unroll
                    for (int i = 0; i < H; i++) {            // even lane: (2i, 2i + 1) at k + 2i; odd lane: (2i + 1, 2i + 2) at k + 2i + 1
                        const v2f lo2 = odd ? acc[j][2 * i + 1] : acc[j][2 * i];
                        const v2f hi2 = odd ? acc[j][2 * i + 2] : acc[j][2 * i + 1];
                        *reinterpret_cast<v4f *>(out + k + 2 * i + (odd ? 1 : 0)) = (v4f){lo2.x, lo2.y, hi2.x, hi2.y};
                    }
                    if (!odd) *reinterpret_cast<v4f *>(out + k + UPM - 1) = (v4f){acc[j][UPM - 1].x, acc[j][UPM - 1].y, nxt0.x, nxt0.y};
                    continue;
                }
            }
            if constexpr (CPLX && UPM >= 2) {
                if (out16 && k + UP <= a.n_out) {        // UP even here: whole pairs
#pragma unroll
                    for (int r = 0; r + 1 < UPM; r += 2)
                        if (r < UP) *reinterpret_cast<v4f *>(out + k + r) = (v4f){acc[j][r].x, acc[j][r].y, acc[j][r + 1].x, acc[j][r + 1].y};
                    continue;
                }
            }
#pragma unroll
            for (int r = 0; r < UPM; r++)
                if ((UPM == 1 || r < UP) && k + r < a.n_out) out[k + r] = acc[j][r];
        }
    }
}

// poly_rt_kernel for SP = 1 -- the pure interpolators (x2, x3, ... x8: every input sample yields UP outputs), where that
// kernel is bound by LDS bandwidth: each thread reads every sample of its window once per m, 8 bytes per UP multiply-adds
// (x2 at 2^28 samples: 69 GB of LDS reads = 0.87 ms of a 1.55 ms launch, twice the vector ALU's share).  Here a thread owns
// PAIRS of consecutive m (2 tid + 512 j, + 1): with P(c) = (X[c], X[c + 1]), c even -- one aligned 16-byte read, lanes 16 bytes
// apart: conflict-free -- the pair's two sums need, for local times 2s + 1 and 2s,
//     m:      P(m + 2s).hi, P(m + 2s).lo          m + 1:  P(m + 2s + 2).lo, P(m + 2s).hi
// i.e. ONE new pair per two taps and two m: half the bytes per multiply-add.  Both sums still run tap index ascending from
// 0.0f (the reference's order): EXACT is bit-identical to the compiled reference.  (poly_tiled_kernel<1, UP> has the same
// pairing but 512-m tiles, its taps by vector loads and its outputs staged through LDS: 0.37 of the roofline for x2 and x4
// when instantiated, against 0.50 for poly_rt_kernel -- profiles/r04/shapes_interpolators.txt.)
template <bool CPLX, bool EXACT, bool IN_U8, int UPM, int MBP, int NLD>
__global__ __launch_bounds__(256) void poly_rt1_kernel(PolyTiledArgs a)
{
    typedef typename Elem<CPLX>::T T;
    typedef typename Pair<CPLX>::P P2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T *X = reinterpret_cast<T *>(smem);

    const unsigned tid = threadIdx.x;
    const int ch = blockIdx.y;
    const T *in = static_cast<const T *>(a.in) + (IN_U8 ? 0 : (size_t)ch * a.in_stride);
    const unsigned char *in8 = static_cast<const unsigned char *>(a.in) + (size_t)ch * a.in_stride * (CPLX ? 2 : 1);
    const T *hist = static_cast<const T *>(a.hist) + (size_t)ch * a.hl;
    T *out = static_cast<T *>(a.out) + (size_t)ch * a.out_stride;

    if (!EXACT && a.hist_out && blockIdx.x == a.tiles) {        // the history workgroup (poly_tiled_kernel has the reasoning)
        T *ho = static_cast<T *>(a.hist_out) + (size_t)ch * a.hl;
#pragma unroll 1
        for (unsigned i = tid; i < (unsigned)a.hl; i += 256u) {
            if constexpr (IN_U8) ho[i] = vload_u8<CPLX>(in8, hist, a.n_in - a.hl + i, a.n_in, a.hl);
            else ho[i] = in[a.n_in - a.hl + i];
        }
        return;
    }
    const int UP = a.UP, TMr = a.tm;                            // tm even, Lp even (host)
    const long long m0 = (long long)blockIdx.x * TMr;
    const long long n_org = m0 + a.e_max - (a.Lp - 1);           // stream index of local sample 0
    const unsigned n_tile = (unsigned)TMr + (unsigned)a.Lp;      // X[s] = x[n_org + s]; sample (mi, qt) is X[mi + qt]
    if constexpr (IN_U8) {
        for (unsigned s = tid; s < n_tile; s += 256u) X[s] = vload_u8<CPLX>(in8, hist, n_org + s, a.n_in, a.hl);
    } else if (n_org >= 0 && n_org + n_tile <= a.n_in) {
        const T *src = in + n_org;                               // uniform; every request before the first is looked at (poly_rt_kernel)
        T v[NLD];
#pragma unroll
        for (int u = 0; u < NLD; u++) {
            const unsigned i = tid + 256u * u;
            v[u] = __builtin_nontemporal_load(src + (i < n_tile ? i : n_tile - 1u));
        }
#pragma unroll
        for (int u = 0; u < NLD; u++) {
            const unsigned i = tid + 256u * u;
            if (i < n_tile) X[i] = v[u];
        }
        for (unsigned i = tid + 256u * NLD; i < n_tile; i += 256u) X[i] = __builtin_nontemporal_load(src + i);
    } else {
        for (unsigned s = tid; s < n_tile; s += 256u) X[s] = vload<CPLX>(in, hist, n_org + s, a.n_in, a.hl);
    }
    __syncthreads();

    const __attribute__((address_space(4))) float *gt = (const __attribute__((address_space(4))) float *)a.Gt;
    const P2 *Xp = reinterpret_cast<const P2 *>(X);
    const bool out16 = CPLX && (reinterpret_cast<uintptr_t>(out) & 15u) == 0;
#pragma unroll 1
    for (int mi0 = 2 * (int)tid; mi0 < TMr; mi0 += 512 * MBP) {
        T acc[MBP][2][UPM];
        unsigned cj[MBP];                                        // pair column of pair j relative to pair 0 (a clamped one beyond the tile: not stored)
#pragma unroll
        for (int j = 0; j < MBP; j++) {
            cj[j] = mi0 + 512 * j < TMr ? 256u * j : 0u;
#pragma unroll
            for (int r = 0; r < UPM; r++) acc[j][0][r] = acc[j][1][r] = Elem<CPLX>::zero();
        }
        const unsigned pc0 = (unsigned)mi0 / 2u;
        P2 nxt[MBP];
#pragma unroll
        for (int j = 0; j < MBP; j++) nxt[j] = Xp[pc0 + cj[j] + (unsigned)a.Lp / 2u];
#pragma unroll 2
        for (int sx = a.Lp / 2 - 1; sx >= 0; --sx) {
            P2 cur[MBP];
#pragma unroll
            for (int j = 0; j < MBP; j++) cur[j] = Xp[pc0 + cj[j] + (unsigned)sx];
            float t1[UPM], t0[UPM];
#pragma unroll
            for (int r = 0; r < UPM; r++) {
                t1[r] = gt[8 * (2 * sx + 1) + r];                 // the two rows are 64 consecutive bytes: scalar loads
                t0[r] = gt[8 * (2 * sx) + r];
            }
#pragma unroll
            for (int j = 0; j < MBP; j++) {
#pragma unroll
                for (int r = 0; r < UPM; r++) {
                    acc[j][0][r] = mac<EXACT>(acc[j][0][r], t1[r], pair_hi(cur[j]));
                    acc[j][1][r] = mac<EXACT>(acc[j][1][r], t1[r], pair_lo(nxt[j]));
                }
#pragma unroll
                for (int r = 0; r < UPM; r++) {
                    acc[j][0][r] = mac<EXACT>(acc[j][0][r], t0[r], pair_lo(cur[j]));
                    acc[j][1][r] = mac<EXACT>(acc[j][1][r], t0[r], pair_hi(cur[j]));
                }
                nxt[j] = cur[j];
            }
        }
#pragma unroll
        for (int j = 0; j < MBP; j++) {
            const int mi = mi0 + 512 * j;
            if constexpr (CPLX) {
                // the wave's 64 pairs of m are 128 UP consecutive outputs: through a region of LDS of the wave's own (rows of
                // 2 UP + 1 cells) they leave as contiguous kilobytes (poly_rt_kernel has the reasoning)
                const int mi_w = 2 * (int)(tid & ~63u) + (mi0 - 2 * (int)tid) + 512 * j;
                const long long kw = (long long)UP * (m0 + mi_w);
                if (out16 && UP == UPM && a.y_off && mi_w + 128 <= TMr && kw + 128LL * UP <= a.n_out) {       // uniform over the wave
                    constexpr unsigned RW = 2 * UPM + 1;
                    v2f *Yw = reinterpret_cast<v2f *>(smem + a.y_off) + (tid >> 6) * (64u * RW);
                    const unsigned lane = tid & 63u;
#pragma unroll
                    for (int h = 0; h < 2; h++)
#pragma unroll
                        for (int r = 0; r < UPM; r++) Yw[lane * RW + h * UPM + r] = acc[j][h][r];
#pragma unroll
                    for (int i = 0; i < UPM; i++) {
                        const unsigned o0 = 2u * (lane + 64u * i), o1 = o0 + 1u;
                        const v2f lo2 = Yw[(o0 / (2 * UPM)) * RW + o0 % (2 * UPM)], hi2 = Yw[(o1 / (2 * UPM)) * RW + o1 % (2 * UPM)];
                        __builtin_nontemporal_store((v4f){lo2.x, lo2.y, hi2.x, hi2.y}, reinterpret_cast<v4f *>(out + kw + o0));
                    }
                    continue;
                }
            }
            if (mi >= TMr) continue;
            const long long k = (long long)UP * (m0 + mi);      // 2 UP consecutive outputs, 16 UP bytes from a 16-byte boundary
            if constexpr (CPLX) {
                if (out16 && UP == UPM && k + 2 * UP <= a.n_out) {
#pragma unroll
                    for (int i = 0; i < UPM; i++) {
                        const v2f lo2 = acc[j][(2 * i) / UPM][(2 * i) % UPM], hi2 = acc[j][(2 * i + 1) / UPM][(2 * i + 1) % UPM];
                        *reinterpret_cast<v4f *>(out + k + 2 * i) = (v4f){lo2.x, lo2.y, hi2.x, hi2.y};
                    }
                    continue;
                }
            }
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int r = 0; r < UPM; r++)
                    if (r < UP && k + (long long)h * UP + r < a.n_out) out[k + (long long)h * UP + r] = acc[j][h][r];
        }
    }
}

#ifdef SFE_DIAG
#include "diag/polyphase_diag.inc"
#endif

// ------------------------------------------- integer-step law on the matrix pipe (f32 MFMA)
// OPT-IN (sfe_dsp_rs_set_algo(h, SFE_RS_ALGO_MFMA)), kept as measured evidence.  The 127-tap-per-arm resampler is VALU-bound
// in the tiled kernel (130 v_pk_fma_f32 per output is its arithmetic floor; 39 % of the HBM
// roofline).  Grouping RG = UP*DM consecutive outputs makes the tap matrix A[RG x Kp] 78 % dense
// for that shape, and v_mfma_f32_16x16x4_f32 is an exact k-ordered fmaf chain
// (cdna_hip_programming.md section 3), so the result equals the fused VALU kernel's bit for bit.
// Measured on MI355X (2^28 cf32, 5/3): 1.21 ms at 2.14 GHz with the matrix pipe 62 % busy, against
// 1.10 ms for the VALU kernel at 1.4-1.8 GHz -- same peak rate (64 FLOP/clk/SIMD), 22 % structural
// zeros, so the VALU form stays the default.  cf32 only: columns are (group, re|im) and B is
// read straight from the interleaved LDS tile -- no de-interleave.
//   A fragment, K-step ks: lane l holds A[row = l&15][kk = 4 ks + (l>>4)]
//   B fragment:            lane l holds B[kk = 4 ks + (l>>4)][col = l&15],  col = 2*group + part
//   kk ascending = window position DEscending = tap index ascending (the reference's order)
//   D: lane l, reg i -> row 4 (l>>4) + i, col l&15
constexpr int MF_NB = 4;            // column blocks (of 8 groups) per wave (the K loop is written out for 4)
constexpr int MF_WG = MF_NB * 8;    // groups per wave tile
constexpr int MF_LD = 16;           // staged loads per thread: tiles of up to 4096 samples
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 4) void poly_mfma_kernel(PolyMfmaArgs a)
{
    // WAVE-granular: every wave owns its own tile of MF_WG groups, its own slice of LDS and its
    // own loop over tiles -- no workgroup barrier after the tap fragments are in place, so the
    // four waves of a SIMD (one from each resident workgroup) are free to drift apart.  (Measured:
    // no faster than workgroup-wide tiles with four barriers per tile -- 1.21 ms either way -- so
    // lockstep phases are not what keeps the matrix pipe at 62 %; scripts/probes/mfma_probe.hip
    // reaches 82 % with this K loop alone.)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_tile = a.GS * (MF_WG - 1) + a.Kp;               // samples staged per wave tile
    float *Af = reinterpret_cast<float *>(smem);
    v2f *X = reinterpret_cast<v2f *>(smem + a.a_bytes + (size_t)wave * a.x_bytes);
    const int ch = blockIdx.y;
    const v2f *in = static_cast<const v2f *>(a.in) + (size_t)ch * a.in_stride;
    const v2f *hist = static_cast<const v2f *>(a.hist) + (size_t)ch * a.hl;
    v2f *out = static_cast<v2f *>(a.out) + (size_t)ch * a.out_stride;

    // tap fragments: once per (persistent) workgroup -- the only workgroup-level hand-off
    const int ksteps = a.Kp >> 2;
    for (unsigned i = tid; i < (unsigned)ksteps * 64; i += 256) Af[i] = a.A[i];
    __syncthreads();

    // Column block nb holds the 8 groups  gs*gi + (nb % gs) + 8*gs*(nb / gs)  of the wave's 32:
    // the spacing gs (1, 2 or 4; chosen by the host) makes the 32 lanes of a ds_read_b32 group
    // -- 8 groups x {re,im} x 2 window positions -- fall on 32 different LDS banks.
    const unsigned j = lane & 15, kq = lane >> 4;
    const unsigned gs = a.gs;
    unsigned boff[MF_NB], ygrp[MF_NB];
#pragma unroll
    for (int nb = 0; nb < MF_NB; nb++) {
        ygrp[nb] = gs * (j >> 1) + ((unsigned)nb % gs) + 8 * gs * ((unsigned)nb / gs);
        boff[nb] = 2 * a.GS * ygrp[nb] + (j & 1) + 2 * (a.Kp - 1 - (int)kq);   // float offset at K-step 0
    }
    const float *Xf = reinterpret_cast<const float *>(X);
    const float *ap = Af + lane;

    // Sample tile: up to MF_LD loads per lane, issued one tile AHEAD (in flight during the K
    // loop of the current tile) and written to the wave's LDS slice at the top of the next one.
    v2f stg[MF_LD];
    auto fetch = [&](long long wt) {
        const long long n_org = (long long)a.GS * (wt * MF_WG) + a.u_lo;
        if (n_org >= 0 && n_org + n_tile <= a.n_in) {
            const v2f *src = in + n_org;
#pragma unroll
            for (int i = 0; i < MF_LD; i++)
                if (lane + 64u * i < (unsigned)n_tile) stg[i] = __builtin_nontemporal_load(src + lane + 64u * i);
        } else {
#pragma unroll
            for (int i = 0; i < MF_LD; i++)
                if (lane + 64u * i < (unsigned)n_tile) stg[i] = vload<true>(in, hist, n_org + lane + 64u * i, a.n_in, a.hl);
        }
    };
    const long long wstride = (long long)gridDim.x * 4;
    long long wt = (long long)blockIdx.x * 4 + wave;             // this wave's tile index
    if (wt < a.tiles) fetch(wt);
    for (; wt < a.tiles; wt += wstride) {
        const long long g_first = wt * MF_WG;
#pragma unroll
        for (int i = 0; i < MF_LD; i++)
            if (lane + 64u * i < (unsigned)n_tile) X[lane + 64u * i] = stg[i];
        asm volatile("" ::: "memory");      // LDS ops of one wave execute in order: no barrier needed
        if (wt + wstride < a.tiles) fetch(wt + wstride);

        f32x4 acc[MF_NB];
        f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
        // Software pipeline, two register sets (0: even K-steps, 1: odd), no copies: the fragments
        // of step ks+1 are in flight while step ks multiplies.  bq[] walk DOWN 4 samples per step
        // (ascending tap index); offsets stay non-negative for the ds_read immediate field.
        const float *aq = ap;                         // step ks at aq[0], ks+1 at aq[64]
        const float *bq0 = Xf + boff[0] - 8, *bq1 = Xf + boff[1] - 8, *bq2 = Xf + boff[2] - 8, *bq3 = Xf + boff[3] - 8;
        float a0 = aq[0], a1;                         // bq*: step ks at [8], step ks+1 at [0]
        float b00 = bq0[8], b01 = bq1[8], b02 = bq2[8], b03 = bq3[8], b10, b11, b12, b13;
#define SFE_MF4(A, B0, B1, B2, B3)                                                   \
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A, B0, acc0, 0, 0, 0);                 \
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A, B1, acc1, 0, 0, 0);                 \
    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(A, B2, acc2, 0, 0, 0);                 \
    acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(A, B3, acc3, 0, 0, 0);
        int ks = 0;
        for (; ks + 2 < ksteps; ks += 2) {            // steady state: both prefetches unconditional
            a1 = aq[64];
            b10 = bq0[0]; b11 = bq1[0]; b12 = bq2[0]; b13 = bq3[0];
            SFE_MF4(a0, b00, b01, b02, b03)
            aq += 128;
            bq0 -= 16; bq1 -= 16; bq2 -= 16; bq3 -= 16;
            a0 = aq[0];
            b00 = bq0[8]; b01 = bq1[8]; b02 = bq2[8]; b03 = bq3[8];
            SFE_MF4(a1, b10, b11, b12, b13)
        }
        SFE_MF4(a0, b00, b01, b02, b03)               // step ks (set 0 is loaded)
        if (ks + 1 < ksteps) {                        // even K-step count: one more
            a1 = aq[64];
            b10 = bq0[0]; b11 = bq1[0]; b12 = bq2[0]; b13 = bq3[0];
            SFE_MF4(a1, b10, b11, b12, b13)
        }
#undef SFE_MF4
        acc[0] = acc0; acc[1] = acc1; acc[2] = acc2; acc[3] = acc3;
        asm volatile("" ::: "memory");

        // D -> linear output tile Y[RG * group + row] (complex) in the wave's own slice (X is dead:
        // the wave's LDS reads above have all returned into registers), then whole-line stores
        float *Yf = reinterpret_cast<float *>(X);
#pragma unroll
        for (int nb = 0; nb < MF_NB; nb++) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const unsigned row = 4 * kq + i;
                if ((int)row < a.RG) Yf[2 * (a.RG * ygrp[nb] + row) + (j & 1)] = acc[nb][i];
            }
        }
        asm volatile("" ::: "memory");
        const long long k0 = (long long)a.RG * g_first;
        const int n_y = a.RG * MF_WG;
        const v2f *Y = reinterpret_cast<const v2f *>(Yf);
        if (k0 + n_y <= a.n_out && ((reinterpret_cast<uintptr_t>(out + k0) & 15) == 0) && (n_y & 1) == 0) {
            const v4f *Y4 = reinterpret_cast<const v4f *>(Y);
            v4f *o4 = reinterpret_cast<v4f *>(out + k0);
            for (unsigned i = lane; i < (unsigned)(n_y / 2); i += 64) __builtin_nontemporal_store(Y4[i], o4 + i);
        } else {
            for (unsigned i = lane; i < (unsigned)n_y; i += 64)
                if (k0 + i < a.n_out) out[k0 + i] = Y[i];
        }
        asm volatile("" ::: "memory");
    }
}

// ------------------------------------------------------------------- scheduled law
template <bool CPLX, bool EXACT>
__device__ __forceinline__ typename Elem<CPLX>::T dot_at(const PolyArgs &a,
                                                         const typename Elem<CPLX>::T *in,
                                                         const typename Elem<CPLX>::T *hist,
                                                         long long p)
{
    typedef typename Elem<CPLX>::T T;
    const long long n = floordiv(p, a.U);
    const int ph = (int)(p - n * a.U);
    const float *tp = a.taps + ph * a.plen;
    T acc = Elem<CPLX>::zero();
    for (int j = 0; j < a.plen; j++) acc = mac<EXACT>(acc, tp[j], vload<CPLX>(in, hist, n - j, a.n_in, a.hl));
    return acc;
}

template <bool CPLX, bool EXACT>
__global__ __launch_bounds__(256) void poly_sched_kernel(PolyArgs a)
{
    typedef typename Elem<CPLX>::T T;
    const int ch = blockIdx.y;
    const T *in = static_cast<const T *>(a.in) + (size_t)ch * a.in_stride;
    const T *hist = static_cast<const T *>(a.hist) + (size_t)ch * a.hl;
    T *out = static_cast<T *>(a.out) + (size_t)ch * a.out_stride;
    const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
    if (k >= a.n_out) return;
    const long long p = a.sched_pos[k];
    const float mu = a.sched_mu[k];
    const T s0 = dot_at<CPLX, EXACT>(a, in, hist, p);
    const T s1 = dot_at<CPLX, EXACT>(a, in, hist, p + 1);
    // out = s0*(1.0f-mu) + mu*s1   (libdsp/resample.cxx:147, decimate.cxx:124)
    const float om = 1.0f - mu;
    if constexpr (CPLX) {
        if constexpr (EXACT)
        {
            const v2f l = s0 * (v2f){om, om}, r = (v2f){mu, mu} * s1;
            out[k] = l + r;
        }
        else
            out[k] = __builtin_elementwise_fma((v2f){mu, mu}, s1, s0 * (v2f){om, om});
    } else {
        if constexpr (EXACT) { const float l = s0 * om, r = mu * s1; out[k] = l + r; }
        else out[k] = __builtin_fmaf(mu, s1, s0 * om);
    }
}

// ------------------------------------------------------- general rate, run-length form
struct DevSeg {          // == sfe::TlSeg (timelaw.h), restated here to keep this file HIP-only
    double t0;
    float  d;
    int    k0, count, pad;
};
constexpr int SEG_MAX_LDS = 96;
// >= plen + 1 (the shifted rows read one tap further), a multiple of 4 floats, and never a multiple of 64 floats: a wave's
// lanes read up to U different rows at once (one per phase), and rows a multiple of 256 bytes apart would put the same
// tap of every phase on the same banks -- a U-way conflict on every 16-byte tap read
__host__ __device__ constexpr int seg_row(int plen) { return ((plen + 4) & ~3) % 64 == 0 ? ((plen + 4) & ~3) + 4 : (plen + 4) & ~3; }

template <bool CPLX, bool EXACT>
__global__ __launch_bounds__(256) void poly_seg_kernel(PolySegArgs a)
{
    typedef typename Elem<CPLX>::T T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DevSeg *sg = reinterpret_cast<DevSeg *>(smem);
    // taps in LDS twice, rows padded to a multiple of four floats (16-byte rows): ts[ph][j] = taps[ph][j] and the same
    // rows one tap to the left, tsh[ph][j] = taps[ph][j + 1] -- whichever a lane's second dot product needs (below),
    // four taps come with one aligned 16-byte read
    // (round 5: only phase 0's row is ever read shifted -- the second sum of an output in the LAST phase -- so tsh is ONE row, not U: 32 phases of
    // 127 taps or 160 of 32 fit the LDS beside a call's samples; a.taps_global: filters that still do not -- 160 phases of 127 taps -- leave
    // their taps in memory, read one at a time through the caches.  Until then such a call fell through to the host-scheduled path: a (position,
    // weight) pair per OUTPUT over PCIe, 170-340 ms for 2^26 samples, profiles/r05/speed_sweep.txt)
    const int plp = seg_row(a.plen);
    float *ts = reinterpret_cast<float *>(smem + SEG_MAX_LDS * sizeof(DevSeg));
    float *tsh = ts + (size_t)a.U * plp;
    T *xs = reinterpret_cast<T *>(smem + SEG_MAX_LDS * sizeof(DevSeg) + (a.taps_global ? 0 : (size_t)(a.U + 1) * plp * 4));

    // a.split > 1 (round 5; a reference call -- blksize samples -- larger than the LDS: blksize 8192 and up for complex streams): the call's outputs
    // are dealt to `split` workgroups in equal runs, and each stages the input span ITS outputs reach (positions grow with the output index).
    // Until then such a call fell through to the host-scheduled path: 37 ms for 2^24 samples where blksize 4096 takes 0.5.
    const int chunk_i = (int)(blockIdx.x / (unsigned)a.split), part = (int)(blockIdx.x - (unsigned)chunk_i * (unsigned)a.split);
    const SegChunk c = a.chunks[chunk_i];
    const int ch = blockIdx.y;
    const T *in = static_cast<const T *>(a.in) + (size_t)ch * a.in_stride;
    const T *hist = static_cast<const T *>(a.hist) + (size_t)ch * a.hl;
    T *out = static_cast<T *>(a.out) + (size_t)ch * a.out_stride + c.k_first;
    const DevSeg *gseg = static_cast<const DevSeg *>(a.segs) + c.seg_first;

    const int nsl = c.n_seg < SEG_MAX_LDS ? c.n_seg : SEG_MAX_LDS;
    for (int i = threadIdx.x; i < nsl; i += 256) sg[i] = gseg[i];
    if (!a.taps_global) {
        for (int i = threadIdx.x; i < a.U * plp; i += 256) {
            const int ph = i / plp, j = i - ph * plp;
            ts[i] = j < a.plen ? a.taps[ph * a.plen + j] : 0.0f;
        }
        for (int j = threadIdx.x; j < plp; j += 256) tsh[j] = j + 1 < a.plen ? a.taps[j + 1] : 0.0f;
    }
    // this workgroup's outputs [ka, kb) of the call and the first sample of its tile relative to the call's first (split == 1: all of them, - plen)
    const int ka = (int)((long long)c.n_out * part / a.split), kb = (int)((long long)c.n_out * (part + 1) / a.split);
    if (ka >= kb) return;
    int rel0 = -a.plen, n_tile = c.m + a.plen;          // tile: samples in_off - plen .. in_off + m - 1   (pos >= -1 reaches back plen samples)
    if (a.split > 1) {
        auto sample_of = [&](int k) -> long long {      // floor(position of output k / U): every thread walks the same runs (uniform)
            int q = 0;
            DevSeg g;
            for (;;) {
                g = gseg[q];
                if (k < g.k0 + g.count) break;
                q++;
            }
            return floordiv((long long)floor(g.t0 + (double)(k - g.k0) * (double)g.d), a.U);
        };
        const long long n_first = sample_of(ka), n_last = sample_of(kb - 1) + 1;       // (+ 1: the second sum of an output in the last phase)
        if (part > 0) rel0 = (int)n_first - a.plen;
        long long span = n_last - rel0 + 1;
        if (span > c.m - rel0) span = c.m - rel0;        // nothing beyond the call's own samples is ever read
        n_tile = span > a.tile_cap ? a.tile_cap : (int)span;      // (the launcher sizes tile_cap with room to spare)
    }
    const long long tile0 = c.in_off + rel0;
    if (tile0 >= 0 && tile0 + n_tile <= a.n_in) {
        // an interior call: its samples are requested eight per thread at a time, all in flight together.  Through the
        // guarded loop below every iteration is a branchy load -> wait -> LDS write of its own, one memory latency after
        // the other: seventeen of them were ~25 of the ~29 us a workgroup took (2^28 samples at rate 1.77: 2.3 -> see
        // profiles/r03/general_rate.txt)
        const T *src = in + tile0;
        for (int i0 = (int)threadIdx.x; i0 < n_tile; i0 += 256 * 8) {
            T v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int i = i0 + 256 * u;
                v[u] = i < n_tile ? __builtin_nontemporal_load(src + i) : Elem<CPLX>::zero();
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int i = i0 + 256 * u;
                if (i < n_tile) xs[i] = v[u];
            }
        }
    } else {
        for (int i = threadIdx.x; i < n_tile; i += 256) xs[i] = vload<CPLX>(in, hist, tile0 + i, a.n_in, a.hl);
    }
    __syncthreads();

    // floor(p / U) and p mod U without the 64-bit division (two per output, ~100 instructions each): inside a chunk p
    // is small -- -U <= p < (m + 1) U -- so p + U is an unsigned 32-bit number and its quotient by the uniform U is one
    // v_mul_hi by ceil(2^32 / U), exact while (p + U) U < 2^32 (U a power of two: the multiplier is 2^32 / U, exact
    // outright).  Anything outside that range -- only the reference's out_len-exhausted state gets there -- takes the
    // general path.
    const unsigned Uu = (unsigned)a.U, Minv = Uu > 1u ? 0xFFFFFFFFu / Uu + 1u : 0u;
    const long long fast_hi = (long long)(0xFFFFFFFFu / Uu) - Uu;
    auto split = [&](long long p, long long &n, int &ph) {
        if (p >= -(long long)Uu && p < fast_hi) {
            const unsigned pu = (unsigned)((int)p + (int)Uu);
            const unsigned q = Uu > 1u ? __umulhi(pu, Minv) : pu;
            n = (long long)(int)q - 1;
            ph = (int)(pu - q * Uu);
        } else {
            n = floordiv(p, a.U);
            ph = (int)(p - n * a.U);
        }
    };
    // s0 = s(p), s1 = s(p + 1) (resample.cxx:141-147) TOGETHER.  Position p + 1 is phase ph + 1 of the same input
    // sample n, or -- ph = U - 1 -- phase 0 of sample n + 1 (sh = 1).  Either way both sums run over the SAME samples
    // x[n - i]:   s0 += taps[ph][i] x[n - i],   s1 += taps[ph1][i + sh] x[n - i]   (sh = 1: after s1's own first term,
    // taps[0][0] x[n + 1]) -- one sample read feeds two multiply-accumulates, each sum still in the reference's order
    // (ascending tap index from 0.0f).  The taps come four at a time from the aligned rows: ts for s0 and for s1 with
    // sh = 0, tsh (the rows shifted by one tap) for s1 with sh = 1.  Separately the two sums read every sample twice
    // and their taps one float at a time: 11.1 -> see profiles/r03/general_rate.txt for 127 taps per sum.
    // Samples before the tile (only reachable through the reference's out_len-exhausted state)
    // read as zero instead of out of bounds: the sums stop at the tile's first sample.
    auto dot2 = [&](long long p, T &s0, T &s1) {
        long long n;
        int ph;
        split(p, n, ph);
        const int sh = ph + 1 == a.U;
        const int ph1 = sh ? 0 : ph + 1;
        const long long reach = n - rel0 + 1;                      // samples x[n], x[n-1], ... inside the tile
        const int L0 = reach < a.plen ? (reach > 0 ? (int)reach : 0) : a.plen;               // terms of s0
        const long long reach1 = reach + sh;
        const int J1 = reach1 < a.plen ? (reach1 > 0 ? (int)reach1 : 0) : a.plen;            // terms of s1
        const int L1 = J1 - (sh && J1 > 0 ? 1 : 0);                                          // ... of them over x[n - i]
        const float *ta = a.taps_global ? a.taps + (size_t)ph * a.plen : ts + ph * plp;
        const float *tb = a.taps_global ? a.taps + (size_t)ph1 * a.plen + sh : (sh ? tsh : ts + ph1 * plp);     // (sh: ph1 = 0)
        const T *xp = xs + (n - rel0);
        s0 = Elem<CPLX>::zero();
        s1 = Elem<CPLX>::zero();
        if (sh && J1 > 0) s1 = mac<EXACT>(s1, a.taps_global ? a.taps[0] : ts[0], xp[1]);     // taps[0][0] x[n + 1]
        const int common = L0 < L1 ? L0 : L1;
        int i = 0;
        for (; !a.taps_global && i + 4 <= common; i += 4) {      // (rows in memory are not padded to 16 bytes: one tap at a time there)
            const v4f a4 = *reinterpret_cast<const v4f *>(ta + i), b4 = *reinterpret_cast<const v4f *>(tb + i);
            const T x0 = xp[-i], x1 = xp[-i - 1], x2 = xp[-i - 2], x3 = xp[-i - 3];
            s0 = mac<EXACT>(s0, a4.x, x0);
            s1 = mac<EXACT>(s1, b4.x, x0);
            s0 = mac<EXACT>(s0, a4.y, x1);
            s1 = mac<EXACT>(s1, b4.y, x1);
            s0 = mac<EXACT>(s0, a4.z, x2);
            s1 = mac<EXACT>(s1, b4.z, x2);
            s0 = mac<EXACT>(s0, a4.w, x3);
            s1 = mac<EXACT>(s1, b4.w, x3);
        }
        for (; i < common; i++) {
            const T x0 = xp[-i];
            s0 = mac<EXACT>(s0, ta[i], x0);
            s1 = mac<EXACT>(s1, tb[i], x0);
        }
        for (int j = i; j < L0; j++) s0 = mac<EXACT>(s0, ta[j], xp[-j]);
        for (int j = i; j < L1; j++) s1 = mac<EXACT>(s1, tb[j], xp[-j]);
    };

    int s = 0;                                          // runs are visited in order by each thread
    for (int k = ka + (int)threadIdx.x; k < kb; k += 256) {
        DevSeg g;
        for (;;) {
            g = s < nsl ? sg[s] : gseg[s];
            if (k < g.k0 + g.count) break;
            s++;
        }
        const double t = g.t0 + (double)(k - g.k0) * (double)g.d;       // exact (timelaw.h)
        const double fl = floor(t);
        const float mu = (float)(t - fl);
        const long long p = (fl > -2.0e9 && fl < 2.0e9) ? (long long)(int)fl : (long long)fl;      // (one v_cvt_i32_f64 instead of the 64-bit conversion sequence)
        T s0, s1;
        dot2(p, s0, s1);
        const float om = 1.0f - mu;                                       // resample.cxx:147
        if constexpr (CPLX) {
            if constexpr (EXACT) { const v2f l = s0 * (v2f){om, om}, r = (v2f){mu, mu} * s1; out[k] = l + r; }
            else out[k] = __builtin_elementwise_fma((v2f){mu, mu}, s1, s0 * (v2f){om, om});
        } else {
            if constexpr (EXACT) { const float l = s0 * om, r = mu * s1; out[k] = l + r; }
            else out[k] = __builtin_fmaf(mu, s1, s0 * om);
        }
    }
}

// ---------------------------------------------------------------- history carry-over
__global__ __launch_bounds__(256) void history_update_kernel(const float *in, long long n_in,
                                                             long long in_stride,
                                                             const float *old_hist, float *new_hist,
                                                             int hl, int ef, int in_u8)
{
    const int ch = blockIdx.y;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;   // float index in [0, hl*ef)
    if (i >= (long long)hl * ef) return;
    const long long src = (n_in - hl) * ef + i;                      // float index in virtual stream
    const float *oh = old_hist + (size_t)ch * hl * ef;
    float v;
    if (src < 0) v = oh[(long long)hl * ef + src];
    else if (in_u8) v = u8_to_f32(reinterpret_cast<const unsigned char *>(in)[(size_t)ch * in_stride * ef + src]);
    else v = in[(size_t)ch * in_stride * ef + src];
    new_hist[(size_t)ch * hl * ef + i] = v;
}

}  // namespace

int launch_poly_int(const PolyArgs &a, int data_complex, int /*taps_complex*/, int exact,
                    int n_channels, hipStream_t s)
{
    if (a.n_out <= 0) return SFE_OK;
    const int esz = data_complex ? 8 : 4;
    // choose the output tile so that input tile + taps fit in 60 KiB of LDS; taps that would leave less than half of it to the samples stay in memory
    const int taps_global = (long long)a.U * a.plen * 4 > 30 * 1024;
    const long long budget = 60 * 1024 - (taps_global ? 0 : (long long)a.U * a.plen * 4);
    if (budget < (long long)(a.plen + 64) * esz) {
        set_error("polyphase: %d taps per phase do not fit the LDS tile", a.plen);
        return SFE_EINVAL;
    }
    long long tile_out = 2048;
    // windows that do not overlap (step >= U plen): the tile holds the outputs' own plen samples each and nothing between them
    const int sparse = (long long)a.step >= (long long)a.U * a.plen;
    auto need = [&](long long to) { return sparse ? to * a.plen : (to * a.step) / a.U + a.plen + 3; };
    while (tile_out > 1 && need(tile_out) * esz > budget) tile_out >>= 1;      // (down to ONE output per workgroup: slow and correct)
    if (need(tile_out) * esz > budget) {
        set_error("polyphase: step %d too large for the LDS tile", a.step);
        return SFE_EINVAL;
    }
    const int tile_in_cap = (int)((need(tile_out) + 3) & ~3LL);
    const size_t shmem = (size_t)tile_in_cap * esz + (taps_global ? 0 : (size_t)a.U * a.plen * 4);
    const long long nb = (a.n_out + tile_out - 1) / tile_out;
    if (nb > 0x7fffffffLL) {
        set_error("polyphase: too many tiles");
        return SFE_EINVAL;
    }
    dim3 grid((unsigned)nb, (unsigned)n_channels), block(256);
#define LAUNCH(C, E) hipLaunchKernelGGL((poly_int_kernel<C, E>), grid, block, shmem, s, a, (int)tile_out, tile_in_cap, sparse, taps_global)
    if (data_complex) { if (exact) LAUNCH(true, true); else LAUNCH(true, false); }
    else { if (exact) LAUNCH(false, true); else LAUNCH(false, false); }
#undef LAUNCH
    SFE_HIP(hipGetLastError());
    return SFE_OK;
}

// the fourteen compile-time instantiations of poly_tiled_kernel (round 4: 2/3, 3/4 -- UP > SP -- and 4/3)
static bool poly_tiled_compiled(int SP, int UP, int Lp)
{
    if (Lp <= 0 || Lp % (2 * SP)) return false;   // whole chunk pairs
    switch (SP * 16 + UP) {
    case 1 * 16 + 1: case 2 * 16 + 1: case 3 * 16 + 1: case 4 * 16 + 1: case 5 * 16 + 1: case 8 * 16 + 1:
    case 10 * 16 + 1: case 5 * 16 + 3: case 3 * 16 + 2: case 5 * 16 + 2: case 5 * 16 + 4:
    case 2 * 16 + 3: case 3 * 16 + 4: case 4 * 16 + 3:      // round 4: 2/3, 3/4 and 4/3 (the pure interpolators run poly_rt1_kernel)
        break;
    default: return false;
    }
    return Lp / SP <= tiled_xc(SP);
}

// ---- poly_rt_kernel's tile: m per workgroup and the LDS row pitch --------------------------------
constexpr int RT_MAX_SP = 64, RT_MAX_UP = 8, RT_LDS_MAX = 60 * 1024;
#ifndef RT1_MAX_UP
#define RT1_MAX_UP 8
#endif
// tm m per tile so that the larger of the input and the output tile is ~4096 samples: a power of two of 256 .. 2048 m
// (a thread then runs 1, 2, 4 or 8 m, whole groups of MB), multiples of 64 below
static int rt_tile_m(int SP, int UP)
{
    const int w = SP > UP ? SP : UP;
    const int ideal = 4096 / w;
    if (ideal < 256) return ideal < 64 ? 64 : ideal / 64 * 64;
    int tm = 256;
    while (tm < 2048 && tm * 2 * 2 <= ideal * 3) tm *= 2;        // the power of two nearest to `ideal` (ratio below 1.5)
    return tm;
}
// row pitch >= tm + Lq: the scatter writes sample s to cell (s % SP) * RL + s / SP; lanes hold consecutive s.
// Pick the pad (0..63) under which a wave's 64 consecutive s fall on the banks most evenly (64 banks of 4
// bytes; an element covers esz / 4 of them).
static int rt_rowlen(int SP, int tm, int Lq, int esz)
{
    const int base = tm + Lq, wpe = esz / 4;
    int best_pad = 0, best_cost = 1 << 30;
    for (int pad = 0; pad < 64; pad++) {
        const int RL = base + pad;
        int cost = 0;
        for (int s0 = 0; s0 < SP * 4; s0 += (SP > 3 ? SP / 3 : 1)) {      // a few phases of the wave against the rows
            int hits[64] = {0};
            for (int l = 0; l < 64; l++) {
                const int s = s0 + l, cell = (s % SP) * RL + s / SP;
                for (int w = 0; w < wpe; w++) hits[(cell * wpe + w) & 63]++;
            }
            int mx = 0;
            for (int b = 0; b < 64; b++) mx = hits[b] > mx ? hits[b] : mx;
            cost += mx;
        }
        if (cost < best_cost) {
            best_cost = cost;
            best_pad = pad;
        }
    }
    return base + best_pad;
}
static bool poly_rt_supported(int SP, int UP, int Lp, int esz)
{
    if (Lp <= 0 || Lp % SP || SP < 1 || SP > RT_MAX_SP || UP < 1 || UP > RT_MAX_UP) return false;
    const int tm = rt_tile_m(SP, UP);
    return (size_t)SP * (tm + Lp / SP + 63) * esz <= (size_t)RT_LDS_MAX;
}

bool poly_tiled_supported(int SP, int UP, int Lp)
{
    return poly_tiled_compiled(SP, UP, Lp) || poly_rt_supported(SP, UP, Lp, 8);
}
bool poly_tiled_is_compiled(int SP, int UP, int Lp) { return poly_tiled_compiled(SP, UP, Lp); }
// ... and, for wire-format (u8) input, the four of them that have a u8 form (launch_poly_tiled: SFE_U8); the other shapes' u8 streams run the
// runtime-shape kernels
bool poly_tiled_u8_is_compiled(int SP, int UP, int Lp)
{
    return poly_tiled_compiled(SP, UP, Lp) && ((UP == 1 && (SP == 2 || SP == 4 || SP == 8)) || (SP == 5 && UP == 3));
}

static int launch_poly_rt(const PolyTiledPlan &plan, const PolyTiledArgs &a0, int data_complex, int exact, int in_u8,
                          int n_channels, hipStream_t s)
{
    const int esz = data_complex ? 8 : 4;
    if (!poly_rt_supported(plan.SP, plan.UP, plan.Lp, esz)) return SFE_ESTATE;
    PolyTiledArgs a = a0;
    a.SP = plan.SP;
    a.UP = plan.UP;
    a.tm = rt_tile_m(plan.SP, plan.UP);
#ifdef SFE_DIAG
    // SFE_RT_TM=<m per tile, a multiple of 64>: smaller tiles = more resident workgroups (scripts/time_shapes.py, round 5)
    if (const char *e = getenv("SFE_RT_TM"))
        if (atoi(e) >= 64 && atoi(e) <= 2048 && atoi(e) % 64 == 0) a.tm = atoi(e);
#endif
    a.rowlen = rt_rowlen(plan.SP, a.tm, plan.Lp / plan.SP, esz);
    a.sp_inv = (unsigned)((0x100000000ull + (unsigned)plan.SP - 1) / (unsigned)plan.SP);
    const long long mtot = (a.n_out + plan.UP - 1) / plan.UP;
    const long long tiles = (mtot + a.tm - 1) / a.tm;
    if (tiles > 0x7fffffffLL) {
        set_error("polyphase: too many tiles");
        return SFE_EINVAL;
    }
    a.tiles = (unsigned)tiles;
    if (exact) a.hist_out = nullptr;
    dim3 grid((unsigned)tiles + (a.hist_out ? 1u : 0u), (unsigned)n_channels), block(256);
    size_t sh = (size_t)plan.SP * a.rowlen * esz;
    a.y_off = 0;
    // + the four waves' output regions (poly_rt_kernel: 64 rows of UP + 1 cells each) where the launch writes at least as much
    // as it reads; the shapes that mostly read (7/4, 7/3) lose more to the LDS the regions take than the stores gain (-4 %)
    if (data_complex && plan.UP >= 3 && plan.UP >= plan.SP) {
        a.y_off = (unsigned)((sh + 15) & ~(size_t)15);
        sh = a.y_off + (size_t)4 * 64 * (plan.UP + 1) * 8;
    }
    // m per thread run together: as many as the tile gives a thread, up to 4 (2 for eight phase sums: registers)
    const int per_thread = (a.tm + 255) / 256;
    // SP = 1, UP >= 2 -- the pure interpolators: pairs of consecutive m per thread (poly_rt1_kernel)
    // (every UP since the outputs leave through the waves' LDS regions; with a thread's 2 UP outputs stored 16 bytes at a stride
    // of 16 UP the pairs LOST from x4 on, where the launch is all stores: 2.68 -> 2.96 ms; with the regions x4 2.49 -> 2.16)
    if (plan.SP == 1 && plan.UP >= 2 && plan.UP <= RT1_MAX_UP && !(plan.Lp & 1) && !(a.tm & 1) && a.tm + plan.Lp <= 2560) {
        const bool six = a.tm + plan.Lp <= 1536;
        size_t sh1 = ((size_t)(a.tm + plan.Lp) * esz + 15) & ~(size_t)15;
        a.y_off = 0;
        if (data_complex) {                      // + the four waves' output regions: 64 rows of 2 UP + 1 cells
            a.y_off = (unsigned)sh1;
            sh1 += (size_t)4 * 64 * (2 * plan.UP + 1) * 8;
        }
#define SFE_R1B(C, E, U8, UPMv, MBPv)                                                                \
    do {                                                                                              \
        if (six || U8) hipLaunchKernelGGL((poly_rt1_kernel<C, E, U8, UPMv, MBPv, 6>), grid, block, sh1, s, a);          \
        else hipLaunchKernelGGL((poly_rt1_kernel<C, E, U8 && false, UPMv, MBPv, 10>), grid, block, sh1, s, a);          \
    } while (0)
#define SFE_R1(C, E, U8)                                                                              \
    do {                                                                                              \
        switch (plan.UP) {                                                                            \
        case 2: SFE_R1B(C, E, U8, 2, 2); break;                                                       \
        case 3: SFE_R1B(C, E, U8, 3, 2); break;                                                       \
        case 4: SFE_R1B(C, E, U8, 4, 2); break;                                                       \
        case 5: SFE_R1B(C, E, U8, 5, 1); break;                                                       \
        case 6: SFE_R1B(C, E, U8, 6, 1); break;                                                       \
        case 7: SFE_R1B(C, E, U8, 7, 1); break;                                                       \
        default: SFE_R1B(C, E, U8, 8, 1); break;                                                      \
        }                                                                                             \
    } while (0)
        if (in_u8) {
            if (data_complex) SFE_R1(true, false, true); else SFE_R1(false, false, true);
        } else if (data_complex) {
            if (exact) SFE_R1(true, true, false); else SFE_R1(true, false, false);
        } else {
            if (exact) SFE_R1(false, true, false); else SFE_R1(false, false, false);
        }
#undef SFE_R1
#undef SFE_R1B
        SFE_HIP(hipGetLastError());
        return SFE_OK;
    }
    const int n_tile = plan.SP * a.tm + plan.Lp;
    // rows of 256 requested up front: the smallest compiled-in count that covers the tile (a row too many is a redundant
    // request per thread and two registers; four too many cost /6 a tenth: 0.58 -> 0.64 with fourteen instead of eighteen)
    const int rows_ahead = n_tile <= 1536 ? 6 : (n_tile <= 2560 ? 10 : (n_tile <= 3584 ? 14 : (n_tile <= 4096 ? 16 : 18)));
#define SFE_RT1(C, E, U8, UPMv, MBv)                                                                 \
    do {                                                                                              \
        if (rows_ahead == 6 || U8) hipLaunchKernelGGL((poly_rt_kernel<C, E, U8, UPMv, MBv, 6>), grid, block, sh, s, a);   \
        else if (rows_ahead == 10) hipLaunchKernelGGL((poly_rt_kernel<C, E, U8 && false, UPMv, MBv, 10>), grid, block, sh, s, a);   \
        else if (rows_ahead == 14) hipLaunchKernelGGL((poly_rt_kernel<C, E, U8 && false, UPMv, MBv, 14>), grid, block, sh, s, a);   \
        else if (rows_ahead == 16) hipLaunchKernelGGL((poly_rt_kernel<C, E, U8 && false, UPMv, MBv, 16>), grid, block, sh, s, a);   \
        else hipLaunchKernelGGL((poly_rt_kernel<C, E, U8 && false, UPMv, MBv, 18>), grid, block, sh, s, a);         \
    } while (0)
#define SFE_RT(C, E, U8)                                                                             \
    do {                                                                                              \
        if (plan.UP == 1) {                                                                           \
            if (per_thread >= 4) SFE_RT1(C, E, U8, 1, 4);      \
            else if (per_thread >= 2) SFE_RT1(C, E, U8, 1, 2); \
            else SFE_RT1(C, E, U8, 1, 1);                      \
        } else if (plan.UP == 2) {                                                                    \
            if (per_thread >= 4) SFE_RT1(C, E, U8, 2, 4);      \
            else SFE_RT1(C, E, U8, 2, 2);                      \
        } else if (plan.UP == 3) {                                                                    \
            if (per_thread >= 4) SFE_RT1(C, E, U8, 3, 4);      \
            else SFE_RT1(C, E, U8, 3, 2);                      \
        } else if (plan.UP == 4) {                                                                    \
            if (per_thread >= 4) SFE_RT1(C, E, U8, 4, 4);      \
            else SFE_RT1(C, E, U8, 4, 2);                      \
        } else if (plan.UP == 5) {                                                                    \
            if (per_thread >= 4) SFE_RT1(C, E, U8, 5, 4);      \
            else SFE_RT1(C, E, U8, 5, 2);                      \
        } else if (plan.UP == 6) {                                                                    \
            if (per_thread >= 4) SFE_RT1(C, E, U8, 6, 4);      \
            else SFE_RT1(C, E, U8, 6, 2);                      \
        } else if (plan.UP == 7) {                                                                    \
            if (per_thread >= 4) SFE_RT1(C, E, U8, 7, 4);      \
            else SFE_RT1(C, E, U8, 7, 2);                      \
        } else {                                                                                      \
            if (per_thread >= 4) SFE_RT1(C, E, U8, 8, 4);      \
            else SFE_RT1(C, E, U8, 8, 2);                      \
        }                                                                                             \
    } while (0)
    if (in_u8) {
        if (data_complex) SFE_RT(true, false, true); else SFE_RT(false, false, true);
    } else if (data_complex) {
        if (exact) SFE_RT(true, true, false); else SFE_RT(true, false, false);
    } else {
        if (exact) SFE_RT(false, true, false); else SFE_RT(false, false, false);
    }
#undef SFE_RT1
#undef SFE_RT
    SFE_HIP(hipGetLastError());
    return SFE_OK;
}

int launch_poly_tiled(const PolyTiledPlan &plan, const PolyTiledArgs &a0, int data_complex, int exact,
                      int in_u8, int n_channels, hipStream_t s)
{
    if (a0.n_out <= 0) return SFE_OK;
    if (!poly_tiled_compiled(plan.SP, plan.UP, plan.Lp))          // no compile-time instantiation: SP, UP as launch arguments
        return launch_poly_rt(plan, a0, data_complex, exact, in_u8, n_channels, s);
    const long long mtot = (a0.n_out + plan.UP - 1) / plan.UP;
    const long long tiles = (mtot + TM - 1) / TM;
    if (tiles > 0x7fffffffLL) {
        set_error("polyphase: too many tiles");
        return SFE_EINVAL;
    }
    PolyTiledArgs a = a0;
    a.tiles = (unsigned)tiles;
    if (exact) a.hist_out = nullptr;
    unsigned nwork = (unsigned)tiles;
#ifdef SFE_DIAG
    if (const char *e = getenv("SFE_TILED_WIN")) {      // scripts/dec_modes.py windows: the resident workgroups over `win` windows
        a.win = (unsigned)atoi(e);
        if (a.win > 1u) nwork = (unsigned)((tiles + a.win - 1) / a.win) * a.win;
    }
    if (const char *e = getenv("SFE_TILED_TLB")) a.tlb_ahead = (unsigned)atoi(e);      // scripts/dec_modes.py tlb
#endif
    dim3 grid(nwork + (a.hist_out ? 1u : 0u), (unsigned)n_channels), block(256);      // + the history workgroup
    const size_t esz = data_complex ? 8 : 4;
#ifdef SFE_DIAG
    // the streamed decimator (poly_stream_kernel): runs of consecutive tiles per workgroup with the next tile's samples
    // in flight -- measured slower than one tile per workgroup, selectable in the diagnostic library only
    if (plan.UP == 1 && !exact && !in_u8 && plan.Lp <= 256) {
        long long tpw = 1;
        if (const char *e = getenv("SFE_TILED_TPW")) tpw = atoi(e);
        if (tpw <= -2 && plan.SP == 8 && data_complex) {          // the tiles of a workgroup at the stride of the grid (diagnostic)
            a.tpw = (unsigned)-tpw;
            const unsigned groups = (unsigned)((tiles + a.tpw - 1) / a.tpw);
            dim3 sgrid(groups + (a.hist_out ? 1u : 0u), (unsigned)n_channels);
            hipLaunchKernelGGL((poly_stream_strided_kernel<8, true>), sgrid, block, (size_t)8 * tiled_rowlen(8) * esz, s, a);
            SFE_HIP(hipGetLastError());
            return SFE_OK;
        }
        if (tpw >= 2) {
            a.tpw = (unsigned)tpw;
            const unsigned groups = (unsigned)((tiles + tpw - 1) / tpw);
            dim3 sgrid(groups + (a.hist_out ? 1u : 0u), (unsigned)n_channels);
#define SFE_S(SPv)                                                                                    \
    case SPv: {                                                                                       \
        const size_t sh = (size_t)SPv * tiled_rowlen(SPv) * esz;                                      \
        if (data_complex) hipLaunchKernelGGL((poly_stream_kernel<SPv, true>), sgrid, block, sh, s, a);  \
        else hipLaunchKernelGGL((poly_stream_kernel<SPv, false>), sgrid, block, sh, s, a);             \
        SFE_HIP(hipGetLastError());                                                                   \
        return SFE_OK;                                                                                \
    }
            switch (plan.SP) {
                SFE_S(2) SFE_S(3) SFE_S(4) SFE_S(5) SFE_S(8)      // (10: twenty body loads per thread do not fit beside the dot product's registers)
            default: break;
            }
#undef SFE_S
        }
    }
#endif
#ifdef SFE_DIAG
    if (const char *e = getenv("SFE_TILED_DIAG")) {          // decimate by 8, cf32, fused numerics only
        const int dg = atoi(e);
        if (dg && plan.SP == 8 && plan.UP == 1 && data_complex && !exact && !in_u8) {
            const size_t sh = (size_t)8 * tiled_rowlen(8) * esz;
            if (dg == 1) hipLaunchKernelGGL((poly_tiled_kernel<8, 1, true, false, false, 1>), grid, block, sh, s, a);
            else if (dg == 2) hipLaunchKernelGGL((poly_tiled_kernel<8, 1, true, false, false, 2>), grid, block, sh, s, a);
            else if (dg == 4) hipLaunchKernelGGL((poly_tiled_kernel<8, 1, true, false, false, 4>), grid, block, sh, s, a);
            else hipLaunchKernelGGL((poly_tiled_kernel<8, 1, true, false, false, 3>), grid, block, sh, s, a);
            SFE_HIP(hipGetLastError());
            return SFE_OK;
        }
    }
#endif
    if (in_u8) {   // wire-format input: the decimator / resampler shapes a receive chain uses
        if (exact) {
            set_error("polyphase: u8 input runs the fused kernels only");
            return SFE_EINVAL;
        }
#define SFE_U8(SPv, UPv)                                                                              \
    case SPv * 16 + UPv: {                                                                            \
        const size_t sh = (size_t)SPv * tiled_rowlen(SPv) * esz;                                      \
        if (data_complex) hipLaunchKernelGGL((poly_tiled_kernel<SPv, UPv, true, false, true>), grid, block, sh, s, a);  \
        else hipLaunchKernelGGL((poly_tiled_kernel<SPv, UPv, false, false, true>), grid, block, sh, s, a);             \
    } break;
        switch (plan.SP * 16 + plan.UP) {
            SFE_U8(2, 1) SFE_U8(4, 1) SFE_U8(8, 1) SFE_U8(5, 3)
        default:          // shapes with a float kernel but no u8 instantiation: the runtime-shape kernel converts on load too
            return launch_poly_rt(plan, a0, data_complex, 0, 1, n_channels, s);
        }
#undef SFE_U8
        SFE_HIP(hipGetLastError());
        return SFE_OK;
    }
#define SFE_T(SPv, UPv)                                                                               \
    case SPv * 16 + UPv: {                                                                            \
        /* the output tile is laid out in the same LDS before it is stored: UP TM elements (more than the input's for UP > SP) */ \
        const size_t sh = (size_t)(SPv * tiled_rowlen(SPv) > UPv * TM ? SPv * tiled_rowlen(SPv) : UPv * TM) * esz;   \
        if (data_complex) {                                                                           \
            if (exact) hipLaunchKernelGGL((poly_tiled_kernel<SPv, UPv, true, true>), grid, block, sh, s, a);   \
            else hipLaunchKernelGGL((poly_tiled_kernel<SPv, UPv, true, false>), grid, block, sh, s, a);        \
        } else {                                                                                      \
            if (exact) hipLaunchKernelGGL((poly_tiled_kernel<SPv, UPv, false, true>), grid, block, sh, s, a);  \
            else hipLaunchKernelGGL((poly_tiled_kernel<SPv, UPv, false, false>), grid, block, sh, s, a);       \
        }                                                                                             \
    } break;
    switch (plan.SP * 16 + plan.UP) {
        SFE_T(1, 1) SFE_T(2, 1) SFE_T(3, 1) SFE_T(4, 1) SFE_T(5, 1) SFE_T(8, 1) SFE_T(10, 1)
        SFE_T(5, 3) SFE_T(3, 2) SFE_T(5, 2) SFE_T(5, 4)
        SFE_T(2, 3) SFE_T(3, 4) SFE_T(4, 3)
    default: return SFE_ESTATE;
    }
#undef SFE_T
    SFE_HIP(hipGetLastError());
    return SFE_OK;
}

static void mfma_lds_layout(int GS, int RG, int Kp, size_t *a_bytes, size_t *x_bytes, size_t *n_tile)
{
    *n_tile = (size_t)GS * (MF_WG - 1) + Kp;
    size_t xb = (*n_tile * 8 + 15) & ~(size_t)15;
    const size_t yb = ((size_t)RG * MF_WG * 8 + 15) & ~(size_t)15;
    if (xb < yb) xb = yb;                         // the output tile reuses the wave's sample slice
    *x_bytes = xb;
    *a_bytes = ((size_t)(Kp / 4) * 64 * 4 + 15) & ~(size_t)15;
}

bool poly_mfma_fits(int GS, int RG, int Kp)
{
    if (RG < 1 || RG > 16 || Kp < 4 || (Kp & 3)) return false;
    size_t ab, xb, nt;
    mfma_lds_layout(GS, RG, Kp, &ab, &xb, &nt);
    if (nt > (size_t)MF_LD * 64) return false;
    return ab + 4 * xb <= 64 * 1024;
}

int launch_poly_mfma(const PolyMfmaArgs &a, int n_channels, hipStream_t s)
{
    if (a.n_out <= 0) return SFE_OK;
    if (!poly_mfma_fits(a.GS, a.RG, a.Kp)) return SFE_ESTATE;
    const long long groups = (a.n_out + a.RG - 1) / a.RG;
    const long long tiles = (groups + MF_WG - 1) / MF_WG;          // wave tiles
    size_t ab, xb, nt;
    mfma_lds_layout(a.GS, a.RG, a.Kp, &ab, &xb, &nt);
    const size_t sh = ab + 4 * xb;
    PolyMfmaArgs b = a;
    b.x_bytes = (int)xb;
    b.a_bytes = (int)ab;
    b.tiles = tiles;
    // group spacing inside a column block: the 32 lanes of a ds_read_b32 group read float
    // 2*GS*gs*gi + part - 2*kq (gi < 8, part < 2, kq < 2): pick gs so the banks are all distinct
    b.gs = 1;
    for (int gs = 1; gs <= 4; gs *= 2) {
        unsigned seen = 0;
        bool ok = true;
        for (int gi = 0; gi < 8 && ok; gi++)
            for (int part = 0; part < 2 && ok; part++)
                for (int kq = 0; kq < 2 && ok; kq++) {
                    const unsigned bank = (unsigned)(2 * a.GS * gs * gi + part + 64 - 2 * kq) & 31u;
                    if (seen & (1u << bank)) ok = false;
                    seen |= 1u << bank;
                }
        if (ok) { b.gs = gs; break; }
    }
    // persistent grid = what is co-resident
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, poly_mfma_kernel, 256, sh) != hipSuccess || per_cu < 1) per_cu = 1;
    if (per_cu > 8) per_cu = 8;
#ifdef SFE_DIAG
    if (const char *e = getenv("SFE_MFMA_WG_PER_CU")) per_cu = atoi(e) > 0 ? atoi(e) : per_cu;
#endif
    long long gx = (tiles + 3) / 4;
    const long long cap = ((long long)device_cu_count() * per_cu + n_channels - 1) / n_channels;
    if (gx > cap) gx = cap;
    dim3 grid((unsigned)gx, (unsigned)n_channels), block(256);
#ifdef SFE_DIAG
    if (getenv("SFE_DEBUG_OCC"))
        fprintf(stderr, "poly_mfma: lds %zu B, grid %lld, wave tiles %lld, gs %d, %d blocks/CU\n", sh, gx, tiles, b.gs, per_cu);
#endif
    hipLaunchKernelGGL(poly_mfma_kernel, grid, block, sh, s, b);
    SFE_HIP(hipGetLastError());
    return SFE_OK;
}

int launch_poly_sched(const PolyArgs &a, int data_complex, int exact, int n_channels, hipStream_t s)
{
    if (a.n_out <= 0) return SFE_OK;
    const long long nb = (a.n_out + 255) / 256;
    dim3 grid((unsigned)nb, (unsigned)n_channels), block(256);
#define LAUNCH(C, E) hipLaunchKernelGGL((poly_sched_kernel<C, E>), grid, block, 0, s, a)
    if (data_complex) { if (exact) LAUNCH(true, true); else LAUNCH(true, false); }
    else { if (exact) LAUNCH(false, true); else LAUNCH(false, false); }
#undef LAUNCH
    SFE_HIP(hipGetLastError());
    return SFE_OK;
}

int launch_poly_seg(const PolySegArgs &a0, int data_complex, int exact, int n_channels, hipStream_t s)
{
    if (a0.n_chunks <= 0) return SFE_OK;
    const size_t esz = data_complex ? 8 : 4;
    PolySegArgs a = a0;
    const size_t taps_b = (size_t)(a.U + 1) * seg_row(a.plen) * 4;
    // a call's samples beside the runs and the taps; a call that does not fit is dealt to `split` workgroups, each with the span its outputs
    // reach: max_m / split samples + what one output's step and the float32 recurrence's wobble can add (a.span_slack, api_rs.hip) + plen
    a.split = 1;
    auto tile = [&](int split) { return split == 1 ? (size_t)a.max_m + a.plen + 1 : (size_t)(a.max_m + split - 1) / split + a.span_slack + a.plen + 2; };
    auto need = [&](int split, bool tg) { return SEG_MAX_LDS * sizeof(DevSeg) + (tg ? 0 : taps_b) + tile(split) * esz; };
    a.taps_global = taps_b > 24 * 1024 && need(1, false) > 64 * 1024;
    while (a.split < 64 && need(a.split, a.taps_global) > 64 * 1024) a.split *= 2;
    if (need(a.split, a.taps_global) > 64 * 1024 && !a.taps_global) {
        a.taps_global = 1;
        a.split = 1;
        while (a.split < 64 && need(a.split, true) > 64 * 1024) a.split *= 2;
    }
    const size_t sh = need(a.split, a.taps_global);
    if (sh > 64 * 1024 || (long long)a.n_chunks * a.split > 0x7fffffffLL) return SFE_ESTATE;
    a.tile_cap = (int)tile(a.split);
    dim3 grid((unsigned)(a.n_chunks * a.split), (unsigned)n_channels), block(256);
#define LAUNCH(C, E) hipLaunchKernelGGL((poly_seg_kernel<C, E>), grid, block, sh, s, a)
    if (data_complex) { if (exact) LAUNCH(true, true); else LAUNCH(true, false); }
    else { if (exact) LAUNCH(false, true); else LAUNCH(false, false); }
#undef LAUNCH
    SFE_HIP(hipGetLastError());
    return SFE_OK;
}

int launch_history_update(const void *in, long long n_in, long long in_stride, const void *old_hist,
                          void *new_hist, int hl, int elem_floats, int n_channels, hipStream_t s, int in_u8)
{
    if (hl <= 0) return SFE_OK;
    const long long nf = (long long)hl * elem_floats;
    dim3 grid((unsigned)((nf + 255) / 256), (unsigned)n_channels), block(256);
    hipLaunchKernelGGL(history_update_kernel, grid, block, 0, s, static_cast<const float *>(in), n_in,
                       in_stride, static_cast<const float *>(old_hist), static_cast<float *>(new_hist),
                       hl, elem_floats, in_u8);
    SFE_HIP(hipGetLastError());
    return SFE_OK;
}

}  // namespace sfe
