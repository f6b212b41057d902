// poly_gen.hip -- the general (non-integer-step) rate in the transform domain.
//
// The reference's `resample` computes ALL U polyphase outputs of every input sample
// (libdsp/resample.cxx:100-114: m_out[j][n] = sum_i taps[i U + j] x[n - i]) and then, for every output
// instant t of the float32 time law, picks two neighbouring ones and blends them (:125-148):
//     y = s(p) (1 - mu) + mu s(p + 1),   p = floor(t), mu = t - p,   s(n U + j) = m_out[j][n].
// `mu` only enters that final blend; the two values it blends are phase samples -- U plain FIR filters
// of the SAME input -- and a transform yields those: ONE forward 4096-point transform of the input block,
// U spectrum products with the phases' spectra, U inverse transforms.  That is (1 + U) transforms per
// 4096 - ovl samples where the direct form (polyphase.hip: poly_seg_kernel) spends two dot products
// of plen taps per OUTPUT read at per-lane conflicting LDS addresses: 381 taps in 3 phases at rate 1.77 --
// ~570 flop per input sample direct, ~250 here.
//
// One workgroup = one block of the stream, the FIR kernel's transform (fir_fft.hip: N = 16 x 16 x 16, thread t
// owns 16 complex values, padded exchange buffer, the same twiddle tables and the same spectrum layout):
//   block b transforms x[b A - ovl .. b A - ovl + 4096), A <= 4096 - ovl (all of it for rates >= ~1), ovl >= plen a multiple of 16 (the outputs are
//   picked from LDS sample by sample, so the overlap need not be whole rows of 256 as in the FIR kernel), and
//   OWNS the outputs whose first phase sample is s(p) with floor(p / U) in [b A - 1, (b + 1) A - 1): both
//   s(p) and s(p + 1) then lie in what the block's inverse transforms produce validly.
//   0. the runs (timelaw.h: t_i = t0 + i d, exact in double) of the one or two reference calls the block
//      overlaps are copied to LDS; every thread finds the block's first and last output by the same
//      uniform search and expands ITS outputs k = k0 + tid + 256 q into (position, mu), kept in LDS [q][thread]
//   1. forward transform -> this thread's 16 bins X, kept in registers
//   2. for each phase j: X H_j -> inverse transform -> S_j[n] into the exchange buffer -> every thread
//      adds its outputs' share: (1 - mu) S_j[n] where the output's first sample has phase j, mu S_j[n'] where
//      its second one has (the next phase of the same input sample, or phase 0 of the next one)
//   3. the outputs are stored, lanes = consecutive outputs
// (pos, mu) are the reference's own sequence, bit for bit (the runs reproduce the float32 recurrence);
// the arithmetic is fused and transform-domain: rel-RMS ~3e-7 against the oracle, the exact mode stays on
// poly_seg_kernel.  Complex float32 streams, any rate the reference takes (>= 1 / U).
#include <stdint.h>
#ifdef SFE_DIAG
#include <stdlib.h>
#endif

#include "common.h"
#include "fft16.h"

namespace sfe {
namespace {

struct RunLds {            // TlSeg (timelaw.h) as three 8-byte words
    double t0;
    float d;
    int k0;
    int count, pad;
};
static_assert(sizeof(RunLds) == 24, "TlSeg layout");

constexpr int GEN_MAX_RUNS = 1024;       // runs of the (at most two) calls a block overlaps, in LDS: 24 KiB of the 34 KiB buffer

// KPT: outputs per thread (the block owns at most 256 KPT outputs; the launcher picks it from the rate)
// Registers: X (32) + the transform's working set (32) + the spectrum loads in flight + KPT accumulators is what fits three
// workgroups per CU (<= 168 VGPRs); so the twiddle bases are re-read from L2 where a stage needs them (24 VGPRs) and each
// thread's (position, mu) table sits in LDS behind the exchange buffer, [q][thread] (2 KPT VGPRs).
template <int KPT>
__global__ __launch_bounds__(256, KPT <= 9 ? 3 : 2) void poly_gen4096_kernel(PolyGenArgs a)
{
    __shared__ v2f lds[FFT_ROWS * LDS_K2_STRIDE];
    __shared__ unsigned tab_pos[KPT * 256];
    __shared__ float tab_mu[KPT * 256];
    const unsigned t = threadIdx.x, lo = t & 15u, hi = t >> 4;
    const int ch = blockIdx.y;
    const long long blk = blockIdx.x;
    const v2f *in = static_cast<const v2f *>(a.in) + (size_t)ch * a.in_stride;
    const v2f *hist = static_cast<const v2f *>(a.hist) + (size_t)ch * a.hl;
    v2f *out = static_cast<v2f *>(a.out) + (size_t)ch * a.out_stride;
    const int A = a.adv, U = a.U;                // input samples a block owns: 4096 - ovl, fewer where the outputs would not fit (launcher)

    // ---- the block's 16 rows (thread t: samples base + t + 256 r): requested first, they land under step 0
    v2f nx[16];
    const long long base = blk * A - a.ovl;
    if (base >= 0 && base + FFT_N <= a.n_in) {
#pragma unroll
        for (int r = 0; r < 16; r++) nx[r] = __builtin_nontemporal_load(in + base + 256 * r + t);
    } else {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const long long i = base + 256 * r + (long long)t;
            if (i >= 0) nx[r] = i < a.n_in ? in[i] : (v2f){0.0f, 0.0f};
            else nx[r] = i >= -(long long)a.hl ? hist[a.hl + i] : (v2f){0.0f, 0.0f};
        }
    }

    // ---- 0. which outputs are this block's, and where each of this thread's sits
    // positions on the upsampled grid, absolute (relative to the launch's first input sample):
    // P = in_off U + floor(t); the block owns Plo <= P < Phi.  Call c emitted the outputs with
    // c B U - 1 <= P < (c + 1) B U - 1 (its leftover output sits at relative position -1).
    const long long Plo = (long long)U * (blk * A - 1), Phi = Plo + (long long)U * A;
    const long long BU = (long long)a.blksize * U;
    // c0 = floor((Plo + 1) / BU), c1 = floor(Phi / BU): by a double-precision quotient and one correction either way (the
    // 64-bit integer divisions were ~300 scalar instructions per wave)
    auto fdiv = [&](long long x) -> long long {
        if (x < 0) return 0;                                      // block 0: Plo + 1 = 1 - U
        long long c = (long long)((double)x / (double)BU);
        if (c * BU > x) c--;
        if ((c + 1) * BU <= x) c++;
        return c;
    };
    long long c0 = fdiv(Plo + 1), c1 = fdiv(Phi);
    if (c1 >= a.n_chunks) c1 = a.n_chunks - 1;
    if (c0 > c1) c0 = c1;
    const SegChunk ca = a.chunks[c0], cb = a.chunks[c1];
    RunLds *runs = reinterpret_cast<RunLds *>(lds);
    const int na = ca.n_seg, nb = c1 > c0 ? cb.n_seg : 0;         // host guarantees na + nb <= GEN_MAX_RUNS
    {
        const unsigned long long *ga = reinterpret_cast<const unsigned long long *>(static_cast<const RunLds *>(a.segs) + ca.seg_first);
        const unsigned long long *gb = reinterpret_cast<const unsigned long long *>(static_cast<const RunLds *>(a.segs) + cb.seg_first);
        unsigned long long *ws = reinterpret_cast<unsigned long long *>(lds);
        for (int i = (int)t; i < 3 * na; i += 256) ws[i] = ga[i];
        for (int i = (int)t; i < 3 * nb; i += 256) ws[3 * na + i] = gb[i];
    }
    lds_barrier();
    // outputs of a call whose relative position is < bound (uniform: every thread runs the same search); *run: the run that
    // holds that output (or the one behind the last)
    auto count_below = [&](const RunLds *rs, int n_seg, int n_out, long long bound, int *run) -> int {
        *run = 0;
        if (n_seg == 0 || bound <= -1) return 0;
        const double bd = (double)bound;
        int l = 0, h = n_seg;                    // first run whose t0 >= bound
        while (l < h) {
            const int m = (l + h) >> 1;
            if (rs[m].t0 < bd) l = m + 1; else h = m;
        }
        if (l == 0) return 0;
        const RunLds g = rs[l - 1];
        long long i = g.count;
        if (g.count > 1 && g.d > 0.0f) {
            i = (long long)ceil((bd - g.t0) / (double)g.d);       // first i with t0 + i d >= bound, up to rounding:
            if (i < 0) i = 0;
            if (i > g.count) i = g.count;
            while (i < g.count && g.t0 + (double)i * (double)g.d < bd) i++;
            while (i > 0 && g.t0 + (double)(i - 1) * (double)g.d >= bd) i--;
        }
        *run = i < g.count ? l - 1 : l;
        const int k = g.k0 + (int)i;
        return k < n_out ? k : n_out;
    };
    const RunLds *ra = runs, *rb = runs + na;
    const long long offa = ca.in_off * U, offb = cb.in_off * U;
    // four bounds, four waves: wave w finds ONE of them (the search is uniform inside a wave: every lane of all four waves
    // running all four searches was a seventh of the kernel's vector instructions) and lane 0 publishes it
    __shared__ int s_bound[8];
    const unsigned w = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63u;
    {
        const bool in_b = w >= 2u;
        int run = 0;
        const int r = (in_b && c1 == c0) ? 0
                                         : count_below(in_b ? rb : ra, in_b ? nb : na, in_b ? cb.n_out : ca.n_out,
                                                       ((w & 1u) ? Phi : Plo) - (in_b ? offb : offa), &run);
        if (lane == 0) {
            s_bound[w] = r;
            s_bound[4 + w] = run;
        }
    }
    lds_barrier();
    const int ka_lo = __builtin_amdgcn_readfirstlane(s_bound[0]), ka_hi = __builtin_amdgcn_readfirstlane(s_bound[1]);
    const int kb_lo = __builtin_amdgcn_readfirstlane(s_bound[2]), kb_hi = __builtin_amdgcn_readfirstlane(s_bound[3]);
    const int ra_lo = __builtin_amdgcn_readfirstlane(s_bound[4]), rb_lo = __builtin_amdgcn_readfirstlane(s_bound[6]);
    const int Ta = ka_hi - ka_lo, T = Ta + (kb_hi - kb_lo);        // T <= 256 KPT (launcher)
    const long long k_first = Ta > 0 || c1 == c0 ? ca.k_first + ka_lo : cb.k_first + kb_lo;   // the block's outputs are consecutive

    // The table, [output of the block]: the byte address in the exchange buffer of the output's first phase sample << 16 |
    // (cells to its second one: 0 the next phase of the same input sample, 1 / 17 phase 0 of the next one) << 10 | the
    // second sample's phase << 5 | the first one's; and mu.  Filled 64 consecutive outputs of ONE call at a time (wave w:
    // every fourth such piece), so that what a piece needs to know about its call is uniform and scalar: the run of its first
    // output is found by walking on from the wave's previous piece, each lane then walks on to its own (pieces hold one to
    // a few runs, except where a call starts: there a dozen binades pass in as many outputs).  Round 4's first version
    // searched and walked per output, with 64-bit positions: 850 of the block's 2 900 vector instructions per wave.
    const unsigned Uu = (unsigned)U, Minv = Uu > 1u ? 0xFFFFFFFFu / Uu + 1u : 0u;      // floor(x / U) = mulhi(x, Minv), x < U A
    const unsigned e0 = (unsigned)a.ovl - 1u;    // transform element of the block's first owned input sample
#pragma unroll 1
    for (int part = 0; part < 2; part++) {
        const RunLds *rs = part ? rb : ra;
        const int ns = part ? nb : na, k_lo = part ? kb_lo : ka_lo, n_part = part ? T - Ta : Ta, idx_base = part ? Ta : 0;
        const unsigned off32 = (unsigned)((part ? offb : offa) - Plo);       // + floor(t) (|.| < 2^31, launcher): < U A, mod 2^32
        int lu = part ? rb_lo : ra_lo;
        if (lu > ns - 1) lu = ns - 1;
#pragma unroll 1
        for (int c = (int)w; 64 * c < n_part; c += 4) {
            const int kk0 = k_lo + 64 * c;
            while (lu + 1 < ns && __builtin_amdgcn_readfirstlane(rs[lu].k0 + rs[lu].count) <= kk0) lu++;
            const int i_part = 64 * c + (int)lane;
            if (i_part < n_part) {
                const int kk = kk0 + (int)lane;
                int l = lu;
                while (rs[l].k0 + rs[l].count <= kk) l++;
                const RunLds g = rs[l];
                const double tt = g.t0 + (double)(kk - g.k0) * (double)g.d;      // exact (timelaw.h)
                const double fl = floor(tt);
                const float mu = (float)(tt - fl);
                const unsigned pl = off32 + (unsigned)(int)fl;                  // position inside the block, < U A
                const unsigned n = Uu > 1u ? __umulhi(pl, Minv) : pl;
                const unsigned ph = pl - n * Uu, e = e0 + n;
                const bool wrap = ph + 1u == Uu;
                const unsigned cell = (e >> 8) * LDS_K2_STRIDE + (e & 255u);
                const unsigned dcell = wrap ? ((e & 255u) == 255u ? LDS_K2_STRIDE - 255u : 1u) : 0u;
                tab_pos[idx_base + i_part] = (cell << 19) | (dcell << 10) | ((wrap ? 0u : ph + 1u) << 5) | ph;
                tab_mu[idx_base + i_part] = mu;
            }
        }
    }
    lds_barrier();                               // the runs are dead: the buffer is the exchange buffer from here on

    // ---- twiddle bases (fir_fft.hip: W^(e (4a + b)) = q[a] p[b]), re-read where a stage needs them: the index is made
    // opaque each time so that the loads are not hoisted back out of the phase loop into 24 resident registers
    v2f p1[4], q1[4], p2[4], q2[4];
    auto load_tw1 = [&]() {
        unsigned tt = t;
        asm volatile("" : "+v"(tt));
#pragma unroll
        for (int k = 1; k < 4; k++) {
            p1[k] = a.tw1[k * 256 + tt];
            q1[k] = a.tw1[(k + 3) * 256 + tt];
        }
    };
    auto load_tw2 = [&]() {
        unsigned ll = lo;
        asm volatile("" : "+v"(ll));
#pragma unroll
        for (int k = 1; k < 4; k++) {
            p2[k] = a.tw2[k * 16 + ll];
            q2[k] = a.tw2[(k + 3) * 16 + ll];
        }
    };
    load_tw1();
    load_tw2();                                  // the sixteen-entry bases stay resident (12 VGPRs); the 256-entry ones are re-read
    const unsigned base_b = hi * LDS_K2_STRIDE + lo, base_c = hi * LDS_K2_STRIDE + lo * LDS_K1_STRIDE;

    // ---- 1. forward transform: F1 over n2, F2 over n1, F3 over n0 -> bin k of this thread in X[k]
    dft16<-1>(nx);
#pragma unroll
    for (int k = 0; k < 16; k++) {
        v2f x = nx[P16(k)];
        if ((k >> 2) && (k & 3)) x = cmul2(x, q1[k >> 2], p1[k & 3]);
        else if (k >> 2) x = cmul(x, q1[k >> 2]);
        else if (k & 3) x = cmul(x, p1[k & 3]);
        lds[t + (unsigned)k * LDS_K2_STRIDE] = x;
    }
    lds_barrier();
    v2f v[16];
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = lds[base_b + 16u * r];
    dft16<-1>(v);
    lds_barrier();
#pragma unroll
    for (int k = 0; k < 16; k++) {
        v2f x = v[P16(k)];
        if ((k >> 2) && (k & 3)) x = cmul2(x, q2[k >> 2], p2[k & 3]);
        else if (k >> 2) x = cmul(x, q2[k >> 2]);
        else if (k & 3) x = cmul(x, p2[k & 3]);
        lds[base_b + (unsigned)LDS_K1_STRIDE * k] = x;
    }
    lds_barrier();
    // the first phase's spectrum: requested here, it lands under the last forward stage; phase j + 1's is requested as soon
    // as phase j's has been multiplied in (16 loads from L2 per phase that nothing waits for)
    v2f hn[16];
    auto load_h = [&](int j) {
        unsigned tt = t;
        asm volatile("" : "+v"(tt));
        const v2f *hs = a.hs + (size_t)j * 16 * 256;
#pragma unroll
        for (int k = 0; k < 16; k++) hn[k] = hs[k * 256 + tt];
    };
    load_h(0);
    v2f X[16];
#pragma unroll
    for (int r = 0; r < 16; r++) X[r] = lds[base_c + r];
    dft16<-1>(X);                                // bin k sits in X[P16(k)]

    // ---- 2. phase by phase
    v2f acc[KPT];
#pragma unroll
    for (int q = 0; q < KPT; q++) acc[q] = (v2f){0.0f, 0.0f};
#pragma unroll 1
    for (int j = 0; j < U; j++) {
        lds_barrier();                           // (j > 0: every thread is done reading S_{j-1})
#pragma unroll
        for (int k = 0; k < 16; k++) v[P16(k)] = cmul(X[P16(k)], hn[k]);
        if (j + 1 < U) load_h(j + 1);
        load_tw1();                              // for this phase's last stage
        dft16_rev<+1>(v);
#pragma unroll
        for (int k = 0; k < 16; k++) lds[base_c + k] = v[k];
        lds_barrier();
        // I2: over k1
#pragma unroll
        for (int r = 0; r < 16; r++) {
            v2f x = lds[base_b + (unsigned)LDS_K1_STRIDE * r];
            if ((r >> 2) && (r & 3)) x = cmul2_conj(x, q2[r >> 2], p2[r & 3]);
            else if (r >> 2) x = cmul_conj(x, q2[r >> 2]);
            else if (r & 3) x = cmul_conj(x, p2[r & 3]);
            v[r] = x;
        }
        dft16<+1>(v);
        lds_barrier();
#pragma unroll
        for (int k = 0; k < 16; k++) lds[base_b + 16u * k] = v[P16(k)];
        lds_barrier();
        // I3: over k2 -> S_j of transform elements t + 256 r, written back to the cells this thread read
#pragma unroll
        for (int r = 0; r < 16; r++) {
            v2f x = lds[t + (unsigned)r * LDS_K2_STRIDE];
            if ((r >> 2) && (r & 3)) x = cmul2_conj(x, q1[r >> 2], p1[r & 3]);
            else if (r >> 2) x = cmul_conj(x, q1[r >> 2]);
            else if (r & 3) x = cmul_conj(x, p1[r & 3]);
            v[r] = x;
        }
        dft16<+1>(v);
#pragma unroll
        for (int r = 0; r < 16; r++) lds[t + (unsigned)r * LDS_K2_STRIDE] = v[P16(r)];
        lds_barrier();
        // the outputs' shares of S_j
        const unsigned ju = (unsigned)j;
        const char *lb = reinterpret_cast<const char *>(lds);
#pragma unroll
        for (int q = 0; q < KPT; q++) {
            if ((int)t + 256 * q >= T) continue;
            const unsigned pw = tab_pos[256 * q + t];
            const float mu_q = tab_mu[256 * q + t];
            const unsigned a0 = pw >> 16;                                      // byte address of the first sample
            if ((pw & 31u) == ju) {
                const v2f s0 = *reinterpret_cast<const v2f *>(lb + a0);
                const float om = 1.0f - mu_q;                      // resample.cxx:147
                acc[q] = __builtin_elementwise_fma((v2f){om, om}, s0, acc[q]);
            }
            if (((pw >> 5) & 31u) == ju) {
                const v2f s1 = *reinterpret_cast<const v2f *>(lb + a0 + (((pw >> 10) & 31u) << 3));
                acc[q] = __builtin_elementwise_fma((v2f){mu_q, mu_q}, s1, acc[q]);
            }
        }
    }
    // ---- 3. lanes = consecutive outputs
#pragma unroll
    for (int q = 0; q < KPT; q++)
        if ((int)t + 256 * q < T) __builtin_nontemporal_store(acc[q], out + k_first + (long long)t + 256 * q);
}

}  // namespace

#ifdef SFE_DIAG
int launch_poly_gen_persistent(const PolyGenArgs &a, int max_runs_two_calls, float step, int n_channels, hipStream_t s, int tickets);
#endif

int poly_gen_outputs_per_block(int U, int adv, float step)
{
    // consecutive outputs are >= step (1 - 2^-22) apart on the upsampled grid
    const double span = (double)U * adv;
    return (int)(span / ((double)step * (1.0 - 1.0 / 4194304.0))) + 2;
}

// SFE_ESTATE: the shape is outside what this kernel takes (the caller uses poly_seg_kernel)
int launch_poly_gen(const PolyGenArgs &a0, int max_runs_two_calls, float step, int n_channels, hipStream_t s)
{
    if (a0.n_chunks <= 0) return SFE_OK;
    PolyGenArgs a = a0;
    if (a.ovl < a.plen || a.ovl >= FFT_N / 2 || (a.ovl & 15) || max_runs_two_calls > GEN_MAX_RUNS) return SFE_ESTATE;
    // the table's fields: 5 bits per phase; a call's positions as 32-bit integers
    if (a.U > 32 || (long long)a.blksize * a.U >= 0x7fffffffLL) return SFE_ESTATE;
    // A block owns `adv` input samples: all 4096 - ovl its transform yields validly while their outputs fit the table
    // (256 x 16), fewer below that -- rates under ~1, MORE outputs than inputs: what `resample` takes and `decimate` refuses,
    // libdsp/resample.cxx:91 -- down to the reference's own limit rate = 1 / U (step 1: every upsampled position an output).
    // The transform count per input sample grows as adv shrinks; the direct form's cost grows with the OUTPUT count, faster.
    a.adv = FFT_N - a.ovl;
    if (poly_gen_outputs_per_block(a.U, a.adv, step) > 256 * 16) {
        a.adv = (int)((256.0 * 16 - 2) * (double)step * (1.0 - 1.0 / 4194304.0) / a.U) & ~15;
        if (a.adv < 512 || poly_gen_outputs_per_block(a.U, a.adv, step) > 256 * 16) return SFE_ESTATE;
    }
    if (a.blksize < a.adv) return SFE_ESTATE;                 // a block overlaps at most two reference calls
#ifdef SFE_DIAG
    // the persistent form with the next block fetched ahead (measured and not kept: diag/poly_gen_persistent.hip)
    if (const char *e = getenv("SFE_GEN_PERSISTENT"))
        if (atoi(e) > 0 && a.adv == FFT_N - a.ovl) return launch_poly_gen_persistent(a, max_runs_two_calls, step, n_channels, s, atoi(e) > 1);
#endif
    const int per_block = poly_gen_outputs_per_block(a.U, a.adv, step);
    const long long A = a.adv;
    const long long nblk = (a.n_in + A - 1) / A;
    if (nblk > 0x7fffffffLL) return SFE_ESTATE;
    dim3 grid((unsigned)nblk, (unsigned)n_channels), block(256);
    const int kpt = (per_block + 255) / 256;
    if (kpt <= 6) hipLaunchKernelGGL((poly_gen4096_kernel<6>), grid, block, 0, s, a);
    else if (kpt <= 9) hipLaunchKernelGGL((poly_gen4096_kernel<9>), grid, block, 0, s, a);
    else if (kpt <= 12) hipLaunchKernelGGL((poly_gen4096_kernel<12>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((poly_gen4096_kernel<16>), grid, block, 0, s, a);
    SFE_HIP(hipGetLastError());
    return SFE_OK;
}

}  // namespace sfe
