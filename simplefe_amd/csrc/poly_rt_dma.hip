// poly_rt_dma.hip -- the runtime-shape tiled polyphase kernels with their tile fetched by LDS-DMA and read in place (round 5).
//
// The law is poly_rt_kernel's (polyphase.hip): s(p) = sum_j taps[p % U + j U] x[p / U - j] (libdsp/decimate.cxx:132-140; value for
// value the m_out[phase][n] of libdsp/resample.cxx:100-114) at positions p = pos0 + k step with an integer-valued step; with
// g = gcd(step, U): UP = U / g outputs per SP = step / g input samples, the taps folded by the host into rows G[UP][Lp]
// (api_plans.hip: get_tiled_plan; Gt = the same rows transposed, one scalar load per local time).  Same accumulation order (tap
// index ascending from 0.0f, fused multiply-add): the same bits as poly_rt_kernel's default mode.
//
// What differs is how a tile reaches the LDS.  poly_rt_kernel requests the tile's rows into registers and scatters them
// de-interleaved by SP (X[p][c] = x[n_org + SP c + p]) so that for a fixed tap the lanes of a wave -- consecutive m -- read
// consecutive cells.  The scatter is ~8 vector instructions and an LDS write per staged sample, about as many instructions as the
// dot products of the shapes that mostly read (/7: 2-3 outputs of 32 taps per thread and tile), and the rows cost two registers
// each.  None of it is needed: with the tile CONTIGUOUS in the LDS lane m reads sample SP m + qt, and
//   * for an ODD SP that stride visits every bank once per group of lanes (gcd(SP, 32) = 1): conflict-free single-sample reads;
//   * for an EVEN SP the low bits of a sample's LDS index, (sh + qt) mod W, are the same in every lane (W = the power of two in SP,
//     capped by what 16 bytes hold: 2 complex or 4 real samples), so W consecutive taps' samples are ONE aligned wide read at an
//     element stride of SP / W per lane -- conflict-free while SP / W is odd, 2-way, 4-way ... beyond (complex /32 meets in one bank
//     group and still gains).  Taps in front of / behind the aligned groups are read one by one.  The same order: the same bits.
// A contiguous tile is exactly what LDS-DMA lands (global_load_lds_dwordx4: 64 lanes x 16 bytes = 1 KiB per instruction, no VGPR
// destination, no scatter): interior tiles are fetched that way -- from the 16-byte boundary at or below the tile's first sample --
// tiles at a stream's ends by guarded loads into the same layout.
// Shapes: complex AND real float32 streams (real: libdsp's native type), fused arithmetic, SP >= 2, UP = 1 ... 8; three or more outputs
// per m leave through the waves' LDS regions as contiguous kilobytes.  SP = 1, the pure interpolators: real streams up to x7 run the second
// kernel of this file, poly_int4_dma_kernel (the same fetch; four consecutive m per lane, their samples read as 16-byte groups into a register
// window), real x8 and complex x6 / x8 the first one.  Wire-format (u8) input: fetch_tile.  Exact mode and the other complex interpolators keep poly_rt_kernel /
// poly_rt1_kernel, and so do calls whose channels do not start on 16-byte boundaries.  VERDICT r4 item 6; profiles/r05/shapes_rt_dma.txt
// (complex: /7 0.54 -> 0.43 ms, 7/4 0.72 -> 0.59, /13 0.50 -> 0.40, /48 0.49 -> 0.37, 10/3 0.71 -> 0.50 ...), profiles/r05/shapes_real.txt,
// profiles/r05/shapes_interpolators.txt.
#include <stdint.h>
#ifdef SFE_DIAG
#include <stdlib.h>
#endif

#include "common.h"

namespace sfe {
namespace {

template <bool CPLX> struct El;
template <> struct El<true> {
    typedef v2f T;
    __device__ static __forceinline__ v2f mac(v2f acc, float t, v2f x) { return __builtin_elementwise_fma((v2f){t, t}, x, acc); }
};
template <> struct El<false> {
    typedef float T;
    __device__ static __forceinline__ float mac(float acc, float t, float x) { return __builtin_fmaf(t, x, acc); }
};
// W samples of type T as one aligned LDS read
template <bool CPLX, int W> struct Wide { typedef typename El<CPLX>::T V; };
template <> struct Wide<true, 2> {
    typedef v4f V;
    __device__ static __forceinline__ v2f get(const v4f &p, int w) { return w ? (v2f){p.z, p.w} : (v2f){p.x, p.y}; }
};
template <> struct Wide<false, 2> {
    typedef v2f V;
    __device__ static __forceinline__ float get(const v2f &p, int w) { return w ? p.y : p.x; }
};
template <> struct Wide<false, 4> {
    typedef v4f V;
    __device__ static __forceinline__ float get(const v4f &p, int w) { return w == 0 ? p.x : (w == 1 ? p.y : (w == 2 ? p.z : p.w)); }
};

// A tile of n_tile samples from stream index n_org on, into X from the 16-byte boundary at or below n_org (returns sh = the tile's first
// sample's place in X): interior tiles by LDS-DMA, tiles at a stream's ends by guarded loads into the same layout.  The caller synchronises.
// a.in_u8 (the receive wire format, gr-simplefe/lib/source_c_impl.cc / source_f_impl.cc: u8 offset binary, a complex sample = two bytes): the
// tile's RAW bytes land by the same DMA at a.raw_off -- 512 complex or 1024 real samples per kilobyte piece -- and are converted ONCE,
// four bytes = one 16-byte group of float32 per step, into the same X the float32 path fills: everything after the fetch is unchanged,
// and so are the bits ((b - 128) * (1 / 127): common.h u8_to_f32, what poly_rt_kernel's u8 path applies sample by sample).
template <bool CPLX>
__device__ __forceinline__ unsigned fetch_tile(const PolyTiledArgs &a, int ch, const typename El<CPLX>::T *hist,
                                               typename El<CPLX>::T *X, char *smem, long long n_org, unsigned n_tile, unsigned tid)
{
    typedef typename El<CPLX>::T T;
    constexpr int ESZ = CPLX ? 8 : 4, A16 = 16 / ESZ;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
    const unsigned wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane16 = (tid & 63u) * 16u;
    // the fetch starts on the 16-byte boundary at or below the tile's first sample (the channel's base is aligned: launcher)
    const unsigned sh = (unsigned)(((n_org % A16) + A16) % A16);
    auto dma = [&](const char *g, unsigned dst0, unsigned pieces) {          // pieces of 1 KiB from g (uniform) to LDS byte dst0 on
#pragma unroll 1
        for (unsigned p = wv; p < pieces; p += 4u) {
            const unsigned dst = dst0 + (p << 10);
            const char *gp = g + ((size_t)p << 10);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(lane16), "s"(gp), "s"(dst) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    if (a.in_u8) {
        constexpr int BSZ = CPLX ? 2 : 1, A16R = 16 / BSZ;                  // bytes per sample, samples per 16 raw bytes
        const unsigned char *in8 = static_cast<const unsigned char *>(a.in) + (size_t)ch * a.in_stride * BSZ;
        const unsigned shr = (unsigned)(((n_org % A16R) + A16R) % A16R);
        const long long g0 = n_org - (long long)shr;
        constexpr unsigned PIECE = 1024 / BSZ;
        const unsigned pieces = (n_tile + shr + PIECE - 1u) / PIECE;
        if (g0 >= 0 && g0 + (long long)pieces * PIECE <= a.n_in) {
            dma(reinterpret_cast<const char *>(in8 + g0 * BSZ), lds_base + a.raw_off, pieces);
            __syncthreads();                                                 // every wave's pieces have landed
            const unsigned d = shr - sh;                                     // raw sample of float sample s: s + d (d BSZ is a multiple of 4)
            const char *raw = smem + a.raw_off;
            const unsigned groups = (n_tile + sh + A16 - 1u) / A16;
#pragma unroll 1
            for (unsigned gi = tid; gi < groups; gi += 256u) {
                const unsigned w = *reinterpret_cast<const unsigned *>(raw + ((unsigned)A16 * gi + d) * BSZ);      // 2 complex or 4 real samples
                reinterpret_cast<v4f *>(smem)[gi] = (v4f){u8_to_f32(w & 255u), u8_to_f32((w >> 8) & 255u), u8_to_f32((w >> 16) & 255u), u8_to_f32(w >> 24)};
            }
        } else {
#pragma unroll 1
            for (unsigned s = tid; s < n_tile + sh; s += 256u) {
                const long long i = n_org - (long long)sh + s;
                T v = T{};
                if (i >= 0) {
                    if (i < a.n_in) {
                        if constexpr (CPLX) v = (v2f){u8_to_f32(in8[2 * i]), u8_to_f32(in8[2 * i + 1])};
                        else v = u8_to_f32(in8[i]);
                    }
                } else if (i >= -(long long)a.hl) v = hist[a.hl + i];
                X[s] = v;
            }
        }
        return sh;
    }
    const T *in = static_cast<const T *>(a.in) + (size_t)ch * a.in_stride;
    const long long g0 = n_org - (long long)sh;
    constexpr unsigned PIECE = 1024 / ESZ;                               // samples per wave instruction (1 KiB)
    const unsigned pieces = (n_tile + sh + PIECE - 1u) / PIECE;
    if (g0 >= 0 && g0 + (long long)pieces * PIECE <= a.n_in) {
        dma(reinterpret_cast<const char *>(in + g0), lds_base, pieces);
    } else {
#pragma unroll 1
        for (unsigned s = tid; s < n_tile + sh; s += 256u) {
            const long long i = g0 + s;                // the virtual stream: history, then this call's input, zero outside
            X[s] = i >= 0 ? (i < a.n_in ? in[i] : T{}) : (i >= -(long long)a.hl ? hist[a.hl + i] : T{});
        }
    }
    return sh;
}

// the history workgroup's copy: the last hl samples of the call's input, converted where the input is u8
template <bool CPLX>
__device__ __forceinline__ void carry_history(const PolyTiledArgs &a, int ch, unsigned tid)
{
    typedef typename El<CPLX>::T T;
    T *ho = static_cast<T *>(a.hist_out) + (size_t)ch * a.hl;
    if (a.in_u8) {
        constexpr int BSZ = CPLX ? 2 : 1;
        const unsigned char *in8 = static_cast<const unsigned char *>(a.in) + ((size_t)ch * a.in_stride + (a.n_in - a.hl)) * BSZ;
#pragma unroll 1
        for (unsigned i = tid; i < (unsigned)a.hl; i += 256u) {
            if constexpr (CPLX) ho[i] = (v2f){u8_to_f32(in8[2 * i]), u8_to_f32(in8[2 * i + 1])};
            else ho[i] = u8_to_f32(in8[i]);
        }
        return;
    }
    const T *in = static_cast<const T *>(a.in) + (size_t)ch * a.in_stride;
#pragma unroll 1
    for (unsigned i = tid; i < (unsigned)a.hl; i += 256u) ho[i] = in[a.n_in - a.hl + i];
}

// UPM = phase sums compiled in (= UP), MB = m per thread run together, W = samples per LDS read (1: odd SP)
template <bool CPLX, int UPM, int MB, int W>
__global__ __launch_bounds__(256) void poly_rt_dma_kernel(PolyTiledArgs a)
{
    typedef typename El<CPLX>::T T;
    constexpr int ESZ = CPLX ? 8 : 4, A16 = 16 / ESZ;       // samples per 16 bytes
    static_assert(W * ESZ <= 16, "a wide read is at most 16 bytes");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T *X = reinterpret_cast<T *>(smem);
    const unsigned tid = threadIdx.x;
    const int ch = blockIdx.y;
    const T *hist = static_cast<const T *>(a.hist) + (size_t)ch * a.hl;
    T *out = static_cast<T *>(a.out) + (size_t)ch * a.out_stride;

    if (a.hist_out && blockIdx.x == a.tiles) {        // the history workgroup (poly_tiled_kernel has the reasoning)
        carry_history<CPLX>(a, ch, tid);
        return;
    }
    const unsigned SP = (unsigned)a.SP;
    const int UP = a.UP, TMr = a.tm;
    const long long m0 = (long long)blockIdx.x * TMr;
    const long long n_org = (long long)SP * m0 + a.e_max - (a.Lp - 1);   // stream index of local sample 0
    const unsigned sh = fetch_tile<CPLX>(a, ch, hist, X, smem, n_org, SP * (unsigned)TMr + (unsigned)a.Lp, tid);
    __syncthreads();

    // taps: row qt of Gt holds the UP phases' taps at local time qt (padded to 8 floats): ONE scalar load per tap
    const __attribute__((address_space(4))) float *gt = (const __attribute__((address_space(4))) float *)a.Gt;
    const bool out_al = (reinterpret_cast<uintptr_t>(out) & 15u) == 0;
#pragma unroll 1
    for (int mi0 = (int)tid; mi0 < TMr; mi0 += 256 * MB) {
        T acc[MB][UPM];
        unsigned bj[MB];                                 // local sample of m_j's local time 0 (a clamped one beyond the tile: not stored)
#pragma unroll
        for (int j = 0; j < MB; j++) {
            bj[j] = sh + SP * (unsigned)(mi0 + 256 * j < TMr ? mi0 + 256 * j : mi0);
#pragma unroll
            for (int r = 0; r < UPM; r++) acc[j][r] = T{};
        }
        // local time qt descending = tap index ascending; sample (m, qt) sits at sh + SP m + qt
        auto single = [&](int qt) {
            float tp[UPM];
#pragma unroll
            for (int r = 0; r < UPM; r++) tp[r] = gt[8 * qt + r];       // consecutive scalar loads: merged into one s_load_dwordxN
#pragma unroll
            for (int j = 0; j < MB; j++) {
                const T x = X[bj[j] + (unsigned)qt];
#pragma unroll
                for (int r = 0; r < UPM; r++) acc[j][r] = El<CPLX>::mac(acc[j][r], tp[r], x);
            }
        };
        if constexpr (W == 1) {
#pragma unroll 2
            for (int qt = a.Lp - 1; qt >= 0; --qt) {
                T x[MB];
#pragma unroll
                for (int j = 0; j < MB; j++) x[j] = X[bj[j] + (unsigned)qt];
                float tp[UPM];
#pragma unroll
                for (int r = 0; r < UPM; r++) tp[r] = gt[8 * qt + r];
#pragma unroll
                for (int j = 0; j < MB; j++)
#pragma unroll
                    for (int r = 0; r < UPM; r++) acc[j][r] = El<CPLX>::mac(acc[j][r], tp[r], x[j]);
            }
        } else {
            typedef typename Wide<CPLX, W>::V V;
            const V *XW = reinterpret_cast<const V *>(smem);
            int qt = a.Lp - 1;
            // (sh + SP m + qt) mod W = (sh + qt) mod W in every lane: the taps in front of the first aligned group, one by one
#pragma unroll 1
            while (qt >= 0 && ((sh + (unsigned)qt) & (unsigned)(W - 1)) != (unsigned)(W - 1)) single(qt--);
#pragma unroll 2
            for (; qt >= W - 1; qt -= W) {               // one element = the samples of taps qt - W + 1 ... qt
                V p[MB];
#pragma unroll
                for (int j = 0; j < MB; j++) p[j] = XW[(bj[j] + (unsigned)qt) / (unsigned)W];
#pragma unroll
                for (int w = W - 1; w >= 0; w--) {       // highest local time first: tap index ascending
                    const int q = qt - (W - 1 - w);
                    float tp[UPM];
#pragma unroll
                    for (int r = 0; r < UPM; r++) tp[r] = gt[8 * q + r];
#pragma unroll
                    for (int j = 0; j < MB; j++) {
                        const T x = Wide<CPLX, W>::get(p[j], w);
#pragma unroll
                        for (int r = 0; r < UPM; r++) acc[j][r] = El<CPLX>::mac(acc[j][r], tp[r], x);
                    }
                }
            }
#pragma unroll 1
            while (qt >= 0) single(qt--);
        }
#pragma unroll
        for (int j = 0; j < MB; j++) {
            const int mi = mi0 + 256 * j;
            if constexpr (UPM >= 3) {
                // Three or more outputs per m: a lane's UP results are UP elements from its neighbour's -- small pieces at that stride, the
                // worst pattern there is for a launch that writes much.  A wave's 64 m are 64 UP CONSECUTIVE outputs: laid out in a region
                // of LDS of the wave's own (rows of UP + 1 cells: the writes spread over the banks) and read back 16 bytes at a time they
                // leave as whole contiguous kilobytes; nothing but the wave touches the region and a wave's LDS operations execute in
                // order: no barrier (poly_rt_kernel, DESIGN.md 4.2d, has the same for complex streams).
                const int mi_w = ((int)(tid & ~63u)) + (mi0 - (int)tid) + 256 * j;       // the wave's first m of this round
                const long long kw = (long long)UP * (m0 + mi_w);
                if (a.y_off && out_al && mi_w + 64 <= TMr && kw + 64LL * UP <= a.n_out) {      // uniform over the wave
                    T *Yw = reinterpret_cast<T *>(smem + a.y_off) + (tid >> 6) * (64u * (UPM + 1));
                    const unsigned lane = tid & 63u;
#pragma unroll
                    for (int r = 0; r < UPM; r++) Yw[lane * (UPM + 1) + r] = acc[j][r];
                    constexpr int NG = 64 * UPM / A16;                                    // 16-byte groups of the wave's outputs
#pragma unroll
                    for (int i = 0; i < (NG + 63) / 64; i++) {
                        const unsigned gi = lane + 64u * i;
                        if (NG % 64 == 0 || gi < (unsigned)NG) {
                            const unsigned o0 = (unsigned)A16 * gi;
                            T e[A16];
#pragma unroll
                            for (int c = 0; c < A16; c++) e[c] = Yw[((o0 + c) / UPM) * (UPM + 1) + (o0 + c) % UPM];
                            v4f q;
                            if constexpr (CPLX) q = (v4f){e[0].x, e[0].y, e[1].x, e[1].y};
                            else q = (v4f){e[0], e[1], e[2], e[3]};
                            __builtin_nontemporal_store(q, reinterpret_cast<v4f *>(out + kw + o0));
                        }
                    }
                    continue;
                }
            }
            if (mi >= TMr) continue;
            const long long k = (long long)UP * (m0 + mi);
            if constexpr ((UPM * ESZ) % 16 == 0) {
                if (out_al && k + UP <= a.n_out) {           // the lane's UP outputs as whole 16-byte pieces (k UP ESZ is a multiple of 16)
#pragma unroll
                    for (int r = 0; r < UPM; r += A16) {
                        v4f q;
                        if constexpr (CPLX) q = (v4f){acc[j][r].x, acc[j][r].y, acc[j][r + 1].x, acc[j][r + 1].y};
                        else q = (v4f){acc[j][r], acc[j][r + 1], acc[j][r + 2], acc[j][r + 3]};
                        *reinterpret_cast<v4f *>(out + k + r) = q;
                    }
                    continue;
                }
            }
            if constexpr (!CPLX && UPM == 2) {
                if ((reinterpret_cast<uintptr_t>(out) & 7u) == 0 && k + UP <= a.n_out) {
                    *reinterpret_cast<v2f *>(out + k) = (v2f){acc[j][0], acc[j][1]};
                    continue;
                }
            }
#pragma unroll
            for (int r = 0; r < UPM; r++)
                if (k + r < a.n_out) out[k + r] = acc[j][r];
        }
    }
}

// 9 ... 256 outputs per period (10/9, 16/15, 25/24, 147/160, x16, x32: near-unity rate matching and strong interpolation; until round 5 these ran the generic
// one-output-per-thread kernel at 0.05-0.3 of the roofline): the same tile, the same fetch; a thread takes one m and runs its UP phase sums EIGHT at
// a time -- one sample read from the LDS feeds eight multiply-adds, the eight taps of a (local time, group) come as one scalar load from the
// row-padded Gt (a.gt_pitch floats per local time) -- re-reading its Lp samples once per group.  Accumulation order per output: tap index
// ascending, fused -- the bits the other fused kernels would give.
template <bool CPLX>
__global__ __launch_bounds__(256) void poly_rt_dma_many_kernel(PolyTiledArgs a)
{
    typedef typename El<CPLX>::T T;
    constexpr int ESZ = CPLX ? 8 : 4, A16 = 16 / ESZ;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T *X = reinterpret_cast<T *>(smem);
    const unsigned tid = threadIdx.x;
    const int ch = blockIdx.y;
    const T *hist = static_cast<const T *>(a.hist) + (size_t)ch * a.hl;
    T *out = static_cast<T *>(a.out) + (size_t)ch * a.out_stride;
    if (a.hist_out && blockIdx.x == a.tiles) {
        carry_history<CPLX>(a, ch, tid);
        return;
    }
    const unsigned SP = (unsigned)a.SP;
    const int UP = a.UP, TMr = a.tm, pitch = a.gt_pitch;
    const long long m0 = (long long)blockIdx.x * TMr;
    const long long n_org = (long long)SP * m0 + a.e_max - (a.Lp - 1);
    const unsigned sh = fetch_tile<CPLX>(a, ch, hist, X, smem, n_org, SP * (unsigned)TMr + (unsigned)a.Lp, tid);
    __syncthreads();
    const __attribute__((address_space(4))) float *gt = (const __attribute__((address_space(4))) float *)a.Gt;
    // a tile of fewer than 256 m (a long input step): the threads beyond the m take the SAME m's later groups of
    // phases -- thread = (m, group selector), consecutive lanes consecutive m
    // (TMr: 64, 128 or 256 -- launcher -- so that a wave's lanes share their group selector: the taps stay SCALAR loads; per-lane groups made them
    // vector loads and cost this kernel half its speed)
    const int gsel = __builtin_amdgcn_readfirstlane((int)tid / TMr), gstep = 8 * (256 / TMr);
    {
        const int mi = (int)tid & (TMr - 1);
        const unsigned b = sh + SP * (unsigned)mi;
        const long long k = (long long)UP * (m0 + mi);
#pragma unroll 1
        for (int g0 = 8 * gsel; g0 < UP; g0 += gstep) {
            T acc[8];
#pragma unroll
            for (int r = 0; r < 8; r++) acc[r] = T{};
#pragma unroll 2
            for (int qt = a.Lp - 1; qt >= 0; --qt) {                 // local time descending = tap index ascending
                const T x = X[b + (unsigned)qt];
                float tp[8];
#pragma unroll
                for (int r = 0; r < 8; r++) tp[r] = gt[pitch * qt + g0 + r];
#pragma unroll
                for (int r = 0; r < 8; r++) acc[r] = El<CPLX>::mac(acc[r], tp[r], x);
            }
            T *o = out + k + g0;
            if (g0 + 8 <= UP && k + g0 + 8 <= a.n_out && (reinterpret_cast<uintptr_t>(o) & 15u) == 0) {      // the group as 16-byte pieces
#pragma unroll
                for (int r = 0; r < 8; r += A16) {
                    v4f q;
                    if constexpr (CPLX) q = (v4f){acc[r].x, acc[r].y, acc[r + 1].x, acc[r + 1].y};
                    else q = (v4f){acc[r], acc[r + 1], acc[r + 2], acc[r + 3]};
                    *reinterpret_cast<v4f *>(o + r) = q;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 8; r++)
                    if (g0 + r < UP && k + g0 + r < a.n_out) o[r] = acc[r];
            }
        }
    }
}

// REAL streams at SMALL input steps (SP = 1 ... 5; SP = 1: the pure interpolators): a register window.  Lane t takes FOUR consecutive m,
// 4 t ... 4 t + 3: their 4 Lp samples are Lp + 3 SP consecutive ones, read as whole aligned 16-byte groups (lane stride 16 SP bytes: no bank
// conflicts for an odd SP) from the highest local time down -- one new group per four taps, the (3 SP + 6) / 4 above it kept in registers --
// 40 bytes of LDS reads per m at SP = 1 and 32 taps where one sample per tap and m is 128.  That form (poly_rt_dma_kernel<false, UPM, MB, 1>)
// and the compile-time poly_tiled_kernel on 4-byte samples spend their time in the LDS pipe: 4 reads of 4 bytes a lane for 2 UP packed
// multiply-adds per four m and tap.  SH = the place of the tile's first sample in its 16-byte group (the same in every tile of a call:
// tiles are multiples of 4 m) and SPC = SP are compiled in, so that the window is indexed by constants.  Accumulation order per output: tap
// index ascending, fused -- the bits of poly_rt_kernel.  A lane's 4 UP outputs are consecutive; a wave's 256 UP leave through its LDS
// region (rows of 4 UP + 4 or + 8 floats) as whole kilobytes, or, UP = 1 and where the regions do not fit, as the lane's own 16-byte pieces.
template <int UPM, int SH, int SPC>
__global__ __launch_bounds__(256) void poly_int4_dma_kernel(PolyTiledArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *X = reinterpret_cast<float *>(smem);
    const unsigned tid = threadIdx.x;
    const int ch = blockIdx.y;
    const float *hist = static_cast<const float *>(a.hist) + (size_t)ch * a.hl;
    float *out = static_cast<float *>(a.out) + (size_t)ch * a.out_stride;

    if (a.hist_out && blockIdx.x == a.tiles) {        // the history workgroup
        carry_history<false>(a, ch, tid);
        return;
    }
    const int TMr = a.tm, Lp = a.Lp, Q = Lp >> 2;
    const long long m0 = (long long)blockIdx.x * TMr;
    const long long n_org = (long long)SPC * m0 + a.e_max - (Lp - 1);
    fetch_tile<false>(a, ch, hist, X, smem, n_org, (unsigned)(SPC * TMr) + (unsigned)Lp, tid);      // returns SH (launcher)
    __syncthreads();

    const __attribute__((address_space(4))) float *gt = (const __attribute__((address_space(4))) float *)a.Gt;
    const v4f *XW = reinterpret_cast<const v4f *>(smem);
    const bool out_al = (reinterpret_cast<uintptr_t>(out) & 15u) == 0;
    // taps 4 q ... 4 q + 3 of the lane's four m meet samples SH + 4 (SPC t + q) + (0 ... 3 SPC + 3): NG groups from SPC t + q on
    constexpr int NG = (SH + 3 * SPC + 3) / 4 + 1;
    constexpr int UNR = NG > 2 ? NG - 1 : 2;             // the period of the window's rotation: no register moves
#pragma unroll 1
    for (int t = (int)tid; 4 * t < TMr; t += 256) {
        float acc[4][UPM];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int r = 0; r < UPM; r++) acc[i][r] = 0.0f;
        // sample (m = 4 t + i, local time qt) sits at SH + SPC (4 t + i) + qt; local time descending = tap index ascending.
        // Lp % 4 taps in front of the whole groups of four, one sample per read
#pragma unroll 1
        for (int qt = Lp - 1; qt >= 4 * Q; --qt) {
            float tp[UPM];
#pragma unroll
            for (int r = 0; r < UPM; r++) tp[r] = gt[8 * qt + r];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float x = X[SH + SPC * (4 * t + i) + qt];
#pragma unroll
                for (int r = 0; r < UPM; r++) acc[i][r] = __builtin_fmaf(tp[r], x, acc[i][r]);
            }
        }
        v4f g[NG];
#pragma unroll
        for (int j = 1; j < NG; j++) g[j] = XW[SPC * t + Q - 1 + j];
#pragma unroll UNR
        for (int q = Q - 1; q >= 0; --q) {
            g[0] = XW[SPC * t + q];
            float e[4 * NG];
#pragma unroll
            for (int j = 0; j < NG; j++) {
                e[4 * j] = g[j].x;
                e[4 * j + 1] = g[j].y;
                e[4 * j + 2] = g[j].z;
                e[4 * j + 3] = g[j].w;
            }
#pragma unroll
            for (int w = 3; w >= 0; --w) {
                float tp[UPM];
#pragma unroll
                for (int r = 0; r < UPM; r++) tp[r] = gt[8 * (4 * q + w) + r];
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int r = 0; r < UPM; r++) acc[i][r] = __builtin_fmaf(tp[r], e[SH + SPC * i + w], acc[i][r]);
            }
#pragma unroll
            for (int j = NG - 1; j >= 1; j--) g[j] = g[j - 1];
        }
        const int t_w = t - (int)(tid & 63u);                                    // the wave's first lane's t
        const long long kw = (long long)UPM * (m0 + 4LL * t_w);
        if (a.y_off && out_al && 4 * (t_w + 64) <= TMr && kw + 256LL * UPM <= a.n_out) {      // uniform over the wave
            constexpr int ROW = 4 * UPM + (UPM % 2 ? 8 : 4);       // an ODD number of 16-byte cells per lane: the lanes' cells spread over the banks
            float *Yw = reinterpret_cast<float *>(smem + a.y_off) + (tid >> 6) * (64u * ROW);
            const unsigned lane = tid & 63u;
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int r = 0; r < UPM; r++) Yw[lane * ROW + i * UPM + r] = acc[i][r];
#pragma unroll
            for (int ii = 0; ii < UPM; ii++) {                                   // 64 UPM groups of 16 bytes, lanes side by side
                const unsigned gi = lane + 64u * ii;
                const v4f qv = *reinterpret_cast<const v4f *>(Yw + (gi / UPM) * ROW + 4u * (gi % UPM));
                __builtin_nontemporal_store(qv, reinterpret_cast<v4f *>(out + kw + 4u * gi));
            }
            continue;
        }
        const long long k0 = (long long)UPM * (m0 + 4LL * t);
        if (out_al && 4 * t + 4 <= TMr && k0 + 4 * UPM <= a.n_out) {             // the lane's 4 UP consecutive outputs as 16-byte pieces
            float f[4 * UPM];
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int r = 0; r < UPM; r++) f[i * UPM + r] = acc[i][r];
#pragma unroll
            for (int c = 0; c < UPM; c++) {
                const v4f qv = (v4f){f[4 * c], f[4 * c + 1], f[4 * c + 2], f[4 * c + 3]};
                if (UPM == 1) __builtin_nontemporal_store(qv, reinterpret_cast<v4f *>(out + k0));
                else *reinterpret_cast<v4f *>(out + k0 + 4 * c) = qv;
            }
            continue;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const long long k = (long long)UPM * (m0 + 4LL * t + i);
#pragma unroll
            for (int r = 0; r < UPM; r++)
                if (k + r < a.n_out) out[k + r] = acc[i][r];
        }
    }
}

// poly_rt_kernel's tile rule (polyphase.hip: rt_tile_m): the larger of the input and the output tile ~32 KiB -- 4096 complex samples, or
// 8192 real ones (a real stream's tile of 4096 samples is half the bytes behind the same per-tile costs: /7 0.55 against 0.42 ms)
int rt_dma_tile_m(int SP, int UP, int samples)
{
    const int w = SP > UP ? SP : UP;
    const int ideal = samples / w;
    if (ideal < 256) return ideal < 64 ? 64 : ideal / 64 * 64;
    int tm = 256;
    while (tm < 4096 && tm * 2 * 2 <= ideal * 3) tm *= 2;
    return tm;
}

template <bool CPLX, int UPM, int W>
void launch_w(const dim3 &grid, size_t lds, hipStream_t s, const PolyTiledArgs &a, int per_thread)
{
    const dim3 block(256);
    if (per_thread >= 4 && UPM <= 4) hipLaunchKernelGGL((poly_rt_dma_kernel<CPLX, UPM, 4, W>), grid, block, lds, s, a);
    else if (per_thread >= 2) hipLaunchKernelGGL((poly_rt_dma_kernel<CPLX, UPM, 2, W>), grid, block, lds, s, a);
    else hipLaunchKernelGGL((poly_rt_dma_kernel<CPLX, UPM, 1, W>), grid, block, lds, s, a);
}

template <bool CPLX, int UPM>
void launch_up(int W, const dim3 &grid, size_t lds, hipStream_t s, const PolyTiledArgs &a, int per_thread)
{
    if (W == 1) launch_w<CPLX, UPM, 1>(grid, lds, s, a, per_thread);
    else if (W == 2) launch_w<CPLX, UPM, 2>(grid, lds, s, a, per_thread);
    else if constexpr (!CPLX) launch_w<false, UPM, 4>(grid, lds, s, a, per_thread);
}

template <bool CPLX>
void launch_c(int UP, int W, const dim3 &grid, size_t lds, hipStream_t s, const PolyTiledArgs &a, int per_thread)
{
    switch (UP) {
    case 1: launch_up<CPLX, 1>(W, grid, lds, s, a, per_thread); break;
    case 2: launch_up<CPLX, 2>(W, grid, lds, s, a, per_thread); break;
    case 3: launch_up<CPLX, 3>(W, grid, lds, s, a, per_thread); break;
    case 4: launch_up<CPLX, 4>(W, grid, lds, s, a, per_thread); break;
    case 5: launch_up<CPLX, 5>(W, grid, lds, s, a, per_thread); break;
    case 6: launch_up<CPLX, 6>(W, grid, lds, s, a, per_thread); break;
    case 7: launch_up<CPLX, 7>(W, grid, lds, s, a, per_thread); break;
    default: launch_up<CPLX, 8>(W, grid, lds, s, a, per_thread); break;
    }
}

template <int UPM, int SPC>
void launch_int4_sh(int sh, const dim3 &grid, size_t lds, hipStream_t s, const PolyTiledArgs &a)
{
    const dim3 block(256);
    switch (sh) {
    case 0: hipLaunchKernelGGL((poly_int4_dma_kernel<UPM, 0, SPC>), grid, block, lds, s, a); break;
    case 1: hipLaunchKernelGGL((poly_int4_dma_kernel<UPM, 1, SPC>), grid, block, lds, s, a); break;
    case 2: hipLaunchKernelGGL((poly_int4_dma_kernel<UPM, 2, SPC>), grid, block, lds, s, a); break;
    default: hipLaunchKernelGGL((poly_int4_dma_kernel<UPM, 3, SPC>), grid, block, lds, s, a); break;
    }
}

// the compiled register-window shapes: SP = 1 with UP = 1 ... 7; SP = 2, 3 with UP = 1 ... 5; SP = 4, 5 with UP = 1; and 5/3.  (Measured against the
// kernels they replace, real streams, profiles/r05/shapes_real_window.txt: /2 +26 %, /3 +29 %, /4 +10 %, /5 +3 %, 3/2 +16 %, 2/3 +26 %, 3/4 +13 %;
// 5/3 +13 %; the other shapes at SP = 4, 5 with several outputs per m -- 4/3, 4/5, 5/2, 5/4 -- LOSE 2-8 % and keep theirs.)
bool int4_shape(int SP, int UP)
{
    if (UP < 1) return false;
    return SP == 1 ? UP <= 7 : (SP == 2 || SP == 3) ? UP <= 5 : SP == 4 ? UP == 1 : SP == 5 ? (UP == 1 || UP == 3) : false;
}

template <int SPC>
void launch_int4(int UP, int sh, const dim3 &grid, size_t lds, hipStream_t s, const PolyTiledArgs &a)
{
    if constexpr (SPC >= 4) {
        if (SPC == 5 && UP == 3) launch_int4_sh<3, 5>(sh, grid, lds, s, a);
        else launch_int4_sh<1, SPC>(sh, grid, lds, s, a);
    } else {
        switch (UP) {
        case 1: launch_int4_sh<1, SPC>(sh, grid, lds, s, a); break;
        case 2: launch_int4_sh<2, SPC>(sh, grid, lds, s, a); break;
        case 3: launch_int4_sh<3, SPC>(sh, grid, lds, s, a); break;
        case 4: launch_int4_sh<4, SPC>(sh, grid, lds, s, a); break;
        case 5: launch_int4_sh<5, SPC>(sh, grid, lds, s, a); break;
        default:
            if constexpr (SPC == 1) {
                if (UP == 6) launch_int4_sh<6, 1>(sh, grid, lds, s, a);
                else launch_int4_sh<7, 1>(sh, grid, lds, s, a);
            }
            break;
        }
    }
}

}  // namespace

// real streams: the (SP, UP) the register-window kernel is compiled for -- those calls come here even where a compile-time tiled kernel exists
bool poly_rt_dma_window_shape(int SP, int UP) { return int4_shape(SP, UP); }

// SFE_ESTATE: the shape or the buffers are outside what this kernel takes (the caller runs launch_poly_tiled)
int launch_poly_rt_dma(const PolyTiledPlan &plan, const PolyTiledArgs &a0, int data_complex, int in_u8, int n_channels, hipStream_t s)
{
    const int SP = plan.SP, UP = plan.UP;
    const int esz = data_complex ? 8 : 4, a16 = 16 / esz;
    // SP = 1, the pure interpolators: REAL streams run here too (poly_rt1_kernel / poly_tiled_kernel<1, UP> on 4-byte samples: x3 0.33, x4 0.27, x8
    // 0.13 of the roofline against 0.44, 0.53, 0.57 here); complex ones keep poly_rt1_kernel, whose pairs of m halve the LDS reads (x2 ... x5
    // 2-17 % ahead of this kernel, x7 level), except x6 and x8 (+6 % here: rows of 48 / 64 bytes leave as whole 16-byte stores):
    // profiles/r05/shapes_interpolators.txt
    int sp_min = data_complex ? 2 : 1;
    if (data_complex && SP == 1 && (UP == 6 || UP == 8)) sp_min = 1;
#ifdef SFE_DIAG
    if (const char *e = getenv("SFE_RT_DMA_SP1"))      // experiment: every pure interpolator here (scripts/collect_r05_p.sh)
        if (atoi(e)) sp_min = 1;
#endif
    if (UP > 8) {
        // 9 ... 256 outputs per period: poly_rt_dma_many_kernel -- one sample per LDS read at a lane stride of SP samples, so only where that stride
        // meets a bank at most twice (16/15, SP = 16: 16 lanes to a bank, 3.57 ms against the generic kernel's 2.14; profiles/r05/shapes_generic_kernel.txt)
        int g = SP, b = data_complex ? 32 : 64;
        while (b) { const int t = g % b; g = b; b = t; }
        if (g > 2) return SFE_ESTATE;
        sp_min = 1;
    }
    if (SP < sp_min || SP > (UP > 8 ? 256 : 64) || UP < 1 || UP > 256 || plan.Lp <= 0 || !plan.d_Gt) return SFE_ESTATE;
    // 16-byte lanes: every channel's first sample on a 16-byte boundary (u8 input: 8 complex or 16 real samples per lane)
    const int in_a16 = in_u8 ? (data_complex ? 8 : 16) : a16;
    if ((reinterpret_cast<uintptr_t>(a0.in) & 15u) || (n_channels > 1 && (a0.in_stride % in_a16))) return SFE_ESTATE;
#ifdef SFE_DIAG
    if (const char *e = getenv("SFE_RT_DMA"))        // A/B against poly_rt_kernel in one process (scripts/time_shapes.py)
        if (!atoi(e)) return SFE_ESTATE;
#endif
    // samples per LDS read: the power of two in SP, capped by what 16 bytes hold
    int W = 1;
    while (W < a16 && SP % (2 * W) == 0) W *= 2;
    PolyTiledArgs a = a0;
    a.SP = SP;
    a.UP = UP;
    a.tm = UP > 8 ? 256 : rt_dma_tile_m(SP, UP, data_complex ? 4096 : 8192);       // (many outputs per m: one m per thread)
    a.gt_pitch = plan.gt_pitch;
    while (UP > 8 && a.tm > 64 && ((size_t)SP * a.tm + plan.Lp + a16) * esz > 48 * 1024) a.tm /= 2;       // (64, 128 or 256: the kernel deals threads as (m, group), whole waves per group)
    size_t lds = ((((size_t)SP * a.tm + plan.Lp + a16) * esz + 1023) >> 10) << 10;        // whole 1 KiB pieces
    a.y_off = 0;
    if (lds > 60 * 1024) return SFE_ESTATE;
    if (UP >= 3 && UP <= 8 && lds + (size_t)4 * 64 * (UP + 1) * esz <= 60 * 1024) {      // + the four waves' output regions: 64 rows of UP + 1 cells each
        a.y_off = (unsigned)lds;                                          // (where they do not fit -- 11/8 -- the outputs leave lane by lane)
        lds += (size_t)4 * 64 * (UP + 1) * esz;
    }
    // real streams at small input steps: poly_int4_dma_kernel.  The interpolators up to x7 (x2 ... x5 13-24 % ahead of the one-sample-per-read
    // form, x6 / x7 5-7 %; x8 -- 256 multiply-adds per input sample: arithmetic, not the LDS -- 9 % behind it and stays:
    // profiles/r05/shapes_interpolators.txt) and SP = 2 ... 5 with up to five outputs per m (profiles/r05/shapes_real_window.txt)
    bool window = !data_complex && UP <= 8 && int4_shape(SP, UP);
#ifdef SFE_DIAG
    if (const char *e = getenv("SFE_RT_DMA_WINDOW")) window = window && SP <= atoi(e);      // the largest SP that takes this form (0: none)
#endif
    if (window) {
        a.y_off = 0;
        lds = ((((size_t)SP * a.tm + plan.Lp + 8) * 4 + 1023) >> 10) << 10;          // groups up to (SP tm + Lp) / 4 + 1 are read
        if (lds > 60 * 1024) return SFE_ESTATE;
        const size_t regions = (size_t)4 * 64 * (4 * UP + (UP % 2 ? 8 : 4)) * 4;
        if (UP >= 2 && lds + regions <= 60 * 1024) {                                 // (where they do not fit the outputs leave as the lanes' own pieces)
            a.y_off = (unsigned)lds;
            lds += regions;
        }
    }
    a.in_u8 = in_u8;
    a.raw_off = 0;
    if (in_u8) {                                     // + the tile's raw bytes: whole 1 KiB pieces from the 16-byte boundary below its first sample, 16 bytes of slack
        a.raw_off = (unsigned)lds;
        lds += (((size_t)SP * a.tm + plan.Lp + 16) * (data_complex ? 2 : 1) + 1023) / 1024 * 1024 + 16;
        if (lds > 60 * 1024) return SFE_ESTATE;
    }
    const long long mtot = (a.n_out + UP - 1) / UP;
    const long long tiles = (mtot + a.tm - 1) / a.tm;
    if (tiles > 0x7fffffffLL) return SFE_ESTATE;
    a.tiles = (unsigned)tiles;
    const dim3 grid((unsigned)tiles + (a.hist_out ? 1u : 0u), (unsigned)n_channels);
    const int per_thread = (a.tm + 255) / 256;
    if (UP > 8) {
        const dim3 block(256);
        if (data_complex) hipLaunchKernelGGL((poly_rt_dma_many_kernel<true>), grid, block, lds, s, a);
        else hipLaunchKernelGGL((poly_rt_dma_many_kernel<false>), grid, block, lds, s, a);
    } else if (window) {
        const int sh = (int)((((long long)a.e_max - (plan.Lp - 1)) % 4 + 4) % 4);
        switch (SP) {
        case 1: launch_int4<1>(UP, sh, grid, lds, s, a); break;
        case 2: launch_int4<2>(UP, sh, grid, lds, s, a); break;
        case 3: launch_int4<3>(UP, sh, grid, lds, s, a); break;
        case 4: launch_int4<4>(UP, sh, grid, lds, s, a); break;
        default: launch_int4<5>(UP, sh, grid, lds, s, a); break;
        }
    } else if (data_complex) launch_c<true>(UP, W, grid, lds, s, a, per_thread);
    else launch_c<false>(UP, W, grid, lds, s, a, per_thread);
    SFE_HIP(hipGetLastError());
    return SFE_OK;
}

}  // namespace sfe
