// poly_rt_dma.hip -- the runtime-shape tiled polyphase kernel with its tile fetched by LDS-DMA and read in place (round 5).
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
// each.  For an ODD SP none of it is needed: with the tile CONTIGUOUS in the LDS, lane m reads sample SP m + qt -- a stride of
// 2 SP dwords, which over a half-wave of 32 lanes visits every even bank once (gcd(SP, 32) = 1): conflict-free 8-byte reads.  A
// contiguous tile is exactly what LDS-DMA lands (global_load_lds_dwordx4: 64 lanes x 16 bytes = 1 KiB per instruction, no VGPR
// destination, no scatter): interior tiles are fetched that way, tiles at a stream's ends by guarded loads into the same layout.
// EVEN SP: see PAIR below.  Shapes: complex float32 streams, fused arithmetic, SP >= 2, UP = 1 ... 8 (three or more outputs per m leave
// through the waves' LDS regions as contiguous kilobytes); SP = 1, exact mode, real and u8 streams keep poly_rt_kernel / poly_rt1_kernel.  VERDICT r4 item 6; profiles/r05/shapes_rt_dma.txt: /7 0.54 -> 0.43 ms, 7/4 0.72 -> 0.59,
// /9 0.48 -> 0.38, /13 0.50 -> 0.40, /15 0.52 -> 0.39, 9/4 0.73 -> 0.58, 9/2 0.56 -> 0.46 (2^28 cf32, two processes each way).
#include <stdint.h>
#ifdef SFE_DIAG
#include <stdlib.h>
#endif

#include "common.h"

namespace sfe {
namespace {

// PAIR (even SP): two consecutive taps' samples are ONE aligned 16-byte read.  With SP = 2 o the sample of (m, qt) sits at
// sh + SP m + qt, whose parity is that of sh + qt -- the same in every lane -- so taps (qt, qt - 1) with sh + qt odd share the
// element o m + (sh + qt) / 2 of 16 bytes: half the LDS reads, and at an element stride of o per lane conflict-free for odd o
// (ds_read_b128 serves four groups of sixteen lanes whose lane numbers cover every residue mod 16; o a bijection on them),
// 2-way for SP = 4 (mod 8), 4-way for 8 (mod 16), ...: still cheaper than the scatter it replaces (profiles/r05/shapes_rt_dma.txt).
// The taps run in the same order, highest local time first: the same bits.
template <int UPM, int MB, bool PAIR>
__global__ __launch_bounds__(256) void poly_rt_dma_kernel(PolyTiledArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    v2f *X = reinterpret_cast<v2f *>(smem);
    const unsigned tid = threadIdx.x;
    const int ch = blockIdx.y;
    const v2f *in = static_cast<const v2f *>(a.in) + (size_t)ch * a.in_stride;
    const v2f *hist = static_cast<const v2f *>(a.hist) + (size_t)ch * a.hl;
    v2f *out = static_cast<v2f *>(a.out) + (size_t)ch * a.out_stride;

    if (a.hist_out && blockIdx.x == a.tiles) {        // the history workgroup (poly_tiled_kernel has the reasoning)
        v2f *ho = static_cast<v2f *>(a.hist_out) + (size_t)ch * a.hl;
#pragma unroll 1
        for (unsigned i = tid; i < (unsigned)a.hl; i += 256u) ho[i] = in[a.n_in - a.hl + i];
        return;
    }
    const unsigned SP = (unsigned)a.SP;
    const int UP = a.UP, TMr = a.tm;
    const long long m0 = (long long)blockIdx.x * TMr;
    const long long n_org = (long long)SP * m0 + a.e_max - (a.Lp - 1);   // stream index of local sample 0
    const unsigned n_tile = SP * (unsigned)TMr + (unsigned)a.Lp;
    // the fetch starts on a 16-byte boundary: one sample early when n_org is odd (the channel's base is aligned: launcher)
    const unsigned sh = (unsigned)(n_org & 1LL);
    const long long g0 = n_org - (long long)sh;
    const unsigned pieces = ((n_tile + sh) * 8u + 1023u) >> 10;          // 1 KiB = 128 samples per wave instruction
    if (g0 >= 0 && g0 + (long long)pieces * 128 <= a.n_in) {
        const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
        const unsigned wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane16 = (tid & 63u) * 16u;
        const char *g = reinterpret_cast<const char *>(in + g0);             // uniform
#pragma unroll 1
        for (unsigned p = wv; p < pieces; p += 4u) {
            const unsigned dst = lds_base + (p << 10);
            const char *gp = g + ((size_t)p << 10);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(lane16), "s"(gp), "s"(dst) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
#pragma unroll 1
        for (unsigned s = tid; s < n_tile + sh; s += 256u) {
            const long long i = g0 + s;                // the virtual stream: history, then this call's input, zero outside
            X[s] = i >= 0 ? (i < a.n_in ? in[i] : (v2f){0.0f, 0.0f}) : (i >= -(long long)a.hl ? hist[a.hl + i] : (v2f){0.0f, 0.0f});
        }
    }
    __syncthreads();

    // taps: row qt of Gt holds the UP phases' taps at local time qt (padded to 8 floats): ONE scalar load per tap
    const __attribute__((address_space(4))) float *gt = (const __attribute__((address_space(4))) float *)a.Gt;
    const bool out16 = (UPM % 2 == 0) && (reinterpret_cast<uintptr_t>(out) & 15u) == 0;
#pragma unroll 1
    for (int mi0 = (int)tid; mi0 < TMr; mi0 += 256 * MB) {
        v2f acc[MB][UPM];
        unsigned bj[MB];                                 // local sample of m_j's local time 0 (a clamped one beyond the tile: not stored)
#pragma unroll
        for (int j = 0; j < MB; j++) {
            bj[j] = sh + SP * (unsigned)(mi0 + 256 * j < TMr ? mi0 + 256 * j : mi0);
#pragma unroll
            for (int r = 0; r < UPM; r++) acc[j][r] = (v2f){0.0f, 0.0f};
        }
        // local time qt descending = tap index ascending; sample (m, qt) sits at sh + SP m + qt
        auto single = [&](int qt) {
            float tp[UPM];
#pragma unroll
            for (int r = 0; r < UPM; r++) tp[r] = gt[8 * qt + r];       // consecutive scalar loads: merged into one s_load_dwordxN
#pragma unroll
            for (int j = 0; j < MB; j++) {
                const v2f x = X[bj[j] + (unsigned)qt];
#pragma unroll
                for (int r = 0; r < UPM; r++) acc[j][r] = __builtin_elementwise_fma((v2f){tp[r], tp[r]}, x, acc[j][r]);
            }
        };
        if constexpr (!PAIR) {
#pragma unroll 2
            for (int qt = a.Lp - 1; qt >= 0; --qt) {
                v2f x[MB];
#pragma unroll
                for (int j = 0; j < MB; j++) x[j] = X[bj[j] + (unsigned)qt];
                float tp[UPM];
#pragma unroll
                for (int r = 0; r < UPM; r++) tp[r] = gt[8 * qt + r];
#pragma unroll
                for (int j = 0; j < MB; j++)
#pragma unroll
                    for (int r = 0; r < UPM; r++) acc[j][r] = __builtin_elementwise_fma((v2f){tp[r], tp[r]}, x[j], acc[j][r]);
            }
        } else {
            const v4f *X4 = reinterpret_cast<const v4f *>(smem);
            int qt = a.Lp - 1;
            if (sh) single(qt--);                        // sh + qt even: the LOW half of an element whose high half is no tap
#pragma unroll 2
            for (; qt >= 1; qt -= 2) {                   // sh + qt odd: element (sh + SP m + qt) / 2 = (taps qt - 1, qt)
                v4f p[MB];
#pragma unroll
                for (int j = 0; j < MB; j++) p[j] = X4[(bj[j] + (unsigned)qt) >> 1];
                float th[UPM], tl[UPM];
#pragma unroll
                for (int r = 0; r < UPM; r++) {
                    th[r] = gt[8 * qt + r];
                    tl[r] = gt[8 * (qt - 1) + r];
                }
#pragma unroll
                for (int j = 0; j < MB; j++) {
#pragma unroll
                    for (int r = 0; r < UPM; r++) acc[j][r] = __builtin_elementwise_fma((v2f){th[r], th[r]}, (v2f){p[j].z, p[j].w}, acc[j][r]);
#pragma unroll
                    for (int r = 0; r < UPM; r++) acc[j][r] = __builtin_elementwise_fma((v2f){tl[r], tl[r]}, (v2f){p[j].x, p[j].y}, acc[j][r]);
                }
            }
            if (qt == 0) single(0);
        }
#pragma unroll
        for (int j = 0; j < MB; j++) {
            const int mi = mi0 + 256 * j;
            if constexpr (UPM >= 3) {
                // Three or more outputs per m: a lane's UP results are 8 UP bytes from its neighbour's -- 8- or 16-byte pieces at that
                // stride, the worst pattern there is for a launch that writes much.  A wave's 64 m are 64 UP CONSECUTIVE outputs: laid out in
                // a region of LDS of the wave's own (rows of UP + 1 cells: the writes spread over the banks) and read back pair by pair
                // they leave as whole contiguous kilobytes; nothing but the wave touches the region and a wave's LDS operations
                // execute in order: no barrier (poly_rt_kernel, DESIGN.md 4.2d, has the same).
                const int mi_w = ((int)(tid & ~63u)) + (mi0 - (int)tid) + 256 * j;       // the wave's first m of this round
                const long long kw = (long long)UP * (m0 + mi_w);
                if (a.y_off && (reinterpret_cast<uintptr_t>(out) & 15u) == 0 && mi_w + 64 <= TMr && kw + 64LL * UP <= a.n_out) {      // uniform over the wave
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
            if constexpr (UPM >= 2) {
                if (out16 && k + UP <= a.n_out) {            // whole 16-byte pairs (UP even; k UP even, the channel's base aligned)
#pragma unroll
                    for (int r = 0; r + 1 < UPM; r += 2)
                        *reinterpret_cast<v4f *>(out + k + r) = (v4f){acc[j][r].x, acc[j][r].y, acc[j][r + 1].x, acc[j][r + 1].y};
                    continue;
                }
            }
#pragma unroll
            for (int r = 0; r < UPM; r++)
                if (k + r < a.n_out) out[k + r] = acc[j][r];
        }
    }
}

// poly_rt_kernel's tile rule (polyphase.hip: rt_tile_m): the larger of the input and the output tile ~4096 samples
int rt_dma_tile_m(int SP, int UP)
{
    const int w = SP > UP ? SP : UP;
    const int ideal = 4096 / w;
    if (ideal < 256) return ideal < 64 ? 64 : ideal / 64 * 64;
    int tm = 256;
    while (tm < 2048 && tm * 2 * 2 <= ideal * 3) tm *= 2;
    return tm;
}

}  // namespace

// SFE_ESTATE: the shape or the buffers are outside what this kernel takes (the caller runs launch_poly_tiled)
int launch_poly_rt_dma(const PolyTiledPlan &plan, const PolyTiledArgs &a0, int n_channels, hipStream_t s)
{
    const int SP = plan.SP, UP = plan.UP;
    // SP = 1 (the pure interpolators) stays with poly_rt1_kernel: there the LDS bandwidth binds and that kernel's pairs of m halve it
    if (SP < 2 || SP > 64 || UP < 1 || UP > 8 || plan.Lp <= 0 || !plan.d_Gt) return SFE_ESTATE;
    if (!(SP & 1) && (plan.Lp & 1)) return SFE_ESTATE;             // (the pairs want an even tap count: the planner's rows are multiples of SP)
    // 16-byte lanes: every channel's first sample on a 16-byte boundary
    if ((reinterpret_cast<uintptr_t>(a0.in) & 15u) || (n_channels > 1 && (a0.in_stride & 1))) return SFE_ESTATE;
#ifdef SFE_DIAG
    if (const char *e = getenv("SFE_RT_DMA"))        // A/B against poly_rt_kernel in one process (scripts/time_shapes.py)
        if (!atoi(e)) return SFE_ESTATE;
#endif
    PolyTiledArgs a = a0;
    a.SP = SP;
    a.UP = UP;
    a.tm = rt_dma_tile_m(SP, UP);
    size_t lds = ((((size_t)SP * a.tm + plan.Lp + 1) * 8 + 1023) >> 10) << 10;        // whole 1 KiB pieces
    a.y_off = 0;
    if (lds > 60 * 1024) return SFE_ESTATE;
    if (UP >= 3 && lds + (size_t)4 * 64 * (UP + 1) * 8 <= 60 * 1024) {      // + the four waves' output regions: 64 rows of UP + 1 cells each
        a.y_off = (unsigned)lds;                                          // (where they do not fit -- 11/8 -- the outputs leave lane by lane)
        lds += (size_t)4 * 64 * (UP + 1) * 8;
    }
    const long long mtot = (a.n_out + UP - 1) / UP;
    const long long tiles = (mtot + a.tm - 1) / a.tm;
    if (tiles > 0x7fffffffLL) return SFE_ESTATE;
    a.tiles = (unsigned)tiles;
    const dim3 grid((unsigned)tiles + (a.hist_out ? 1u : 0u), (unsigned)n_channels), block(256);
    const int per_thread = (a.tm + 255) / 256;
#define SFE_RD2(UPMv, PR)                                                                                          \
    do {                                                                                                      \
        if (per_thread >= 4 && UPMv <= 4) hipLaunchKernelGGL((poly_rt_dma_kernel<UPMv, 4, PR>), grid, block, lds, s, a);       \
        else if (per_thread >= 2) hipLaunchKernelGGL((poly_rt_dma_kernel<UPMv, 2, PR>), grid, block, lds, s, a);  \
        else hipLaunchKernelGGL((poly_rt_dma_kernel<UPMv, 1, PR>), grid, block, lds, s, a);                       \
    } while (0)
#define SFE_RD(UPMv) do { if (SP & 1) SFE_RD2(UPMv, false); else SFE_RD2(UPMv, true); } while (0)
    switch (UP) {
    case 1: SFE_RD(1); break;
    case 2: SFE_RD(2); break;
    case 3: SFE_RD(3); break;
    case 4: SFE_RD(4); break;
    case 5: SFE_RD(5); break;
    case 6: SFE_RD(6); break;
    case 7: SFE_RD(7); break;
    default: SFE_RD(8); break;
    }
#undef SFE_RD
#undef SFE_RD2
    SFE_HIP(hipGetLastError());
    return SFE_OK;
}

}  // namespace sfe
