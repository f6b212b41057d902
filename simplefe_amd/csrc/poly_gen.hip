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
// poly_seg_kernel.  Complex and real streams (REAL below), float32 or the u8 wire format (IN_U8), any rate the reference
// takes (>= 1 / U).
#include <stdint.h>

#include <type_traits>

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
// REAL: a real float32 stream -- libdsp's native type (the reference's classes take float*).  The taps are real, so ONE
// complex transform carries TWO consecutive blocks, z = x_A + j x_B: Re(IFFT(Z H_j)) is block A's phase j, Im block B's.  A
// workgroup then owns the 2 A input samples of the pair -- one contiguous range of positions, up to three reference calls --
// and each table entry says which half its two samples are read from; twice the outputs per transform, KPT up to 22
// (two workgroups per CU).  The direct kernel ran such a stream at 0.05 of the roofline (2^29 samples, rate 1.77, 381 taps
// in 3 phases: 8.65 ms).
// IN_U8: the stream is the device's receive wire format, u8 offset binary ((I, Q) byte pairs; REAL: one byte per sample),
// converted as the rows are loaded -- (b - 128) / 127, the reference's converter (gr-simplefe/lib/source_c_impl.cc:121-132):
// the same bits as the float32 path fed the converted samples.  The carried history is float32 either way.
// NOPRO (diagnostic library only, round 5): the whole of step 0 -- call records, runs, bound searches, the table of (position, mu) -- replaced by a
// synthetic table of a.diag_T outputs at a fixed step (WRONG results on purpose): what a table that arrived ready-made could save at most.
template <int KPT, bool REAL, bool IN_U8, bool NOPRO = false>
__global__ __launch_bounds__(256, KPT <= 9 ? 3 : 2) void poly_gen4096_kernel(PolyGenArgs a)
{
    constexpr int NC = REAL ? 3 : 2;             // reference calls a block (pair) can overlap: its span <= (REAL ? 2 : 1) x blksize
    typedef typename std::conditional<REAL, float, v2f>::type E;      // a sample
    __shared__ v2f lds[FFT_ROWS * LDS_K2_STRIDE];
    __shared__ unsigned tab_pos[KPT * 256];
    __shared__ float tab_mu[KPT * 256];
    __shared__ int s_bound[2 * NC], s_run[NC];
    const unsigned t = threadIdx.x, lo = t & 15u, hi = t >> 4;
    const int ch = blockIdx.y;
    const long long blk = (REAL ? 2 : 1) * (long long)blockIdx.x;      // REAL: the pair's first block
    const E *in = static_cast<const E *>(a.in) + (size_t)ch * a.in_stride;
    const E *hist = static_cast<const E *>(a.hist) + (size_t)ch * a.hl;
    E *out = static_cast<E *>(a.out) + (size_t)ch * a.out_stride;
    const int A = a.adv, U = a.U;                // input samples a block owns: 4096 - ovl, fewer where the outputs would not fit (launcher)

    // ---- the block's 16 rows (thread t: samples base + t + 256 r): requested first, they land under step 0
    v2f nx[16];
    const long long base = blk * A - a.ovl;
    if constexpr (IN_U8) {
        const unsigned char *in8 = static_cast<const unsigned char *>(a.in) + (size_t)ch * a.in_stride * (REAL ? 1 : 2);
        auto at = [&](long long i) -> E {
            if (i >= 0) {
                if (i >= a.n_in) return E{};
                if constexpr (REAL) return u8_to_f32(in8[i]);
                else return (v2f){u8_to_f32(in8[2 * i]), u8_to_f32(in8[2 * i + 1])};
            }
            return i >= -(long long)a.hl ? hist[a.hl + i] : E{};
        };
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const long long i = base + 256 * r + (long long)t;
            if constexpr (REAL) nx[r] = (v2f){at(i), at(i + A)};
            else nx[r] = at(i);
        }
    } else if constexpr (!REAL) {
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
    } else {
        if (base >= 0 && base + A + FFT_N <= a.n_in) {
#pragma unroll
            for (int r = 0; r < 16; r++)
                nx[r] = (v2f){__builtin_nontemporal_load(in + base + 256 * r + t), __builtin_nontemporal_load(in + base + A + 256 * r + t)};
        } else {
            auto at = [&](long long i) -> float {
                if (i >= 0) return i < a.n_in ? in[i] : 0.0f;
                return i >= -(long long)a.hl ? hist[a.hl + i] : 0.0f;
            };
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const long long i = base + 256 * r + (long long)t;
                nx[r] = (v2f){at(i), at(i + A)};
            }
        }
    }

    int T = 0;
    long long k_first = 0;
    if constexpr (NOPRO) {
        T = a.diag_T < 256 * KPT ? a.diag_T : 256 * KPT;
        k_first = (long long)blockIdx.x * T;
        const unsigned Uu = (unsigned)U, e0 = (unsigned)a.ovl - 1u;
#pragma unroll
        for (int q = 0; q < KPT; q++) {
            const unsigned idx = t + 256u * q;
            const unsigned pl = (unsigned)(((unsigned long long)idx * (unsigned)A * Uu) / (unsigned)(T > 0 ? T : 1));      // spread over the block's positions
            const unsigned n = pl / Uu, ph = pl - n * Uu, e = e0 + (n < (unsigned)A ? n : (unsigned)A - 1u);
            const bool wrap = ph + 1u == Uu;
            const unsigned cell = (e >> 8) * LDS_K2_STRIDE + (e & 255u);
            const unsigned dcell = wrap ? ((e & 255u) == 255u ? LDS_K2_STRIDE - 255u : 1u) : 0u;
            tab_pos[idx] = (int)idx < T ? ((cell << 19) | (dcell << 10) | ((wrap ? 0u : ph + 1u) << 5) | ph) : (31u | (31u << 5));
            tab_mu[idx] = 0.25f;
        }
    } else {
    // ---- 0. which outputs are this block's, and where each of them sits
    // positions on the upsampled grid, absolute (relative to the launch's first input sample):
    // P = in_off U + floor(t); the block (pair) owns Plo <= P < Phi.  Call c emitted the outputs with
    // c B U - 1 <= P < (c + 1) B U - 1 (its leftover output sits at relative position -1).
    const long long Plo = (long long)U * (blk * A - 1), Phi = Plo + (long long)(REAL ? 2 : 1) * U * A;
    const long long BU = (long long)a.blksize * U;
    // floor((Plo + 1) / BU), floor(Phi / BU): by a double-precision quotient and one correction either way (the
    // 64-bit integer divisions were ~300 scalar instructions per wave)
    auto fdiv = [&](long long x) -> long long {
        if (x < 0) return 0;                                      // block 0: Plo + 1 = 1 - U
        long long c = (long long)((double)x / (double)BU);
        if (c * BU > x) c--;
        if ((c + 1) * BU <= x) c++;
        return c;
    };
    long long c_first = fdiv(Plo + 1), c_last = fdiv(Phi);
    if (c_last >= a.n_chunks) c_last = a.n_chunks - 1;
    if (c_first > c_last) c_first = c_last;
    if (c_last > c_first + NC - 1) c_last = c_first + NC - 1;      // (cannot happen: blksize >= adv, launcher)
    const int ncalls = (int)(c_last - c_first) + 1;
    SegChunk cc[NC];
    int nsg[NC], rbase[NC + 1];
    rbase[0] = 0;
#pragma unroll
    for (int i = 0; i < NC; i++) {
        cc[i] = a.chunks[c_first + (i < ncalls ? i : 0)];
        nsg[i] = i < ncalls ? cc[i].n_seg : 0;                    // host guarantees their sum <= GEN_MAX_RUNS
        rbase[i + 1] = rbase[i] + nsg[i];
    }
    RunLds *runs = reinterpret_cast<RunLds *>(lds);
    {
        unsigned long long *ws = reinterpret_cast<unsigned long long *>(lds);
#pragma unroll
        for (int i = 0; i < NC; i++) {
            const unsigned long long *g = reinterpret_cast<const unsigned long long *>(static_cast<const RunLds *>(a.segs) + cc[i].seg_first);
            for (int j = (int)t; j < 3 * nsg[i]; j += 256) ws[3 * rbase[i] + j] = g[j];
        }
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
    // 2 NC bounds (each call's first and last owned output), four waves: wave w finds bounds w, w + 4 (the search is uniform
    // inside a wave: every lane of all four waves running all the searches was a seventh of the kernel's vector
    // instructions) and lane 0 publishes them
    const unsigned w = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63u;
#pragma unroll
    for (int sx = 0; sx < 2 * NC; sx += 4) {
        const int sb = sx + (int)w;              // bound sb: call sb / 2, its lower (even) or upper (odd) end
        if (sb < 2 * NC) {
            int run = 0, r = 0;
#pragma unroll
            for (int i = 0; i < NC; i++)
                if (i == (sb >> 1) && i < ncalls)
                    r = count_below(runs + rbase[i], nsg[i], cc[i].n_out, ((sb & 1) ? Phi : Plo) - cc[i].in_off * U, &run);
            if (lane == 0) {
                s_bound[sb] = r;
                if (!(sb & 1)) s_run[sb >> 1] = run;
            }
        }
    }
    lds_barrier();
    int k_lo[NC], n_part[NC], idx_base[NC + 1], run_lo[NC];
    idx_base[0] = 0;
    k_first = cc[0].k_first + __builtin_amdgcn_readfirstlane(s_bound[0]);
    bool have_first = false;
#pragma unroll
    for (int i = 0; i < NC; i++) {
        k_lo[i] = __builtin_amdgcn_readfirstlane(s_bound[2 * i]);
        n_part[i] = __builtin_amdgcn_readfirstlane(s_bound[2 * i + 1]) - k_lo[i];
        run_lo[i] = __builtin_amdgcn_readfirstlane(s_run[i]);
        idx_base[i + 1] = idx_base[i] + n_part[i];
        if (!have_first && n_part[i] > 0) {      // the block's outputs are consecutive: they start in the first call that has any
            k_first = cc[i].k_first + k_lo[i];
            have_first = true;
        }
    }
    T = idx_base[NC];                            // <= 256 KPT (launcher)
    if (T == 0) return;                          // uniform: nothing lands here (the block behind a stream that ends on a block seam)

    // The table, [output of the block]: the byte address in the exchange buffer of the output's first phase sample << 19 |
    // (cells to its second one: 0 the next phase of the same input sample, 1 / 17 phase 0 of the next one) << 10 | the
    // second sample's phase << 5 | the first one's (REAL: | the half, bit 15); and mu.  Filled 64 consecutive outputs of ONE
    // call at a time (wave w: every fourth such piece), so that what a piece needs to know about its call is uniform and
    // scalar: the run of its first output is found by walking on from the wave's previous piece, each lane then walks on to
    // its own (pieces hold one to a few runs, except where a call starts: there a dozen binades pass in as many outputs).
    // Round 4's first version searched and walked per output, with 64-bit positions: 850 of the block's 2 900 vector
    // instructions per wave.
    // (round 5) Slots behind the block's last output hold an entry that matches no phase -- cell 0, both phase fields 31 (the
    // launcher takes U <= 31, so a real phase is at most 30) -- so that the shares loop of step 2 runs without a guard and
    // without a branch per output.
#pragma unroll
    for (int q = 0; q < KPT; q++)
        if ((int)t + 256 * q >= T) {
            tab_pos[256 * q + t] = 31u | (31u << 5);
            tab_mu[256 * q + t] = 0.0f;
        }
    const unsigned Uu = (unsigned)U, Minv = Uu > 1u ? 0xFFFFFFFFu / Uu + 1u : 0u;      // floor(x / U) = mulhi(x, Minv), x < 2 U A
    const unsigned e0 = (unsigned)a.ovl - 1u;    // transform element of a block's first owned input sample
#pragma unroll
    for (int part = 0; part < NC; part++) {
        const RunLds *rs = runs + rbase[part];
        const int ns = nsg[part];
        const unsigned off32 = (unsigned)(cc[part].in_off * U - Plo);       // + floor(t) (|.| < 2^31, launcher): < 2 U A, mod 2^32
        int lu = run_lo[part];
        if (lu > ns - 1) lu = ns - 1;
#pragma unroll 1
        for (int c = (int)w; 64 * c < n_part[part]; c += 4) {
            const int kk0 = k_lo[part] + 64 * c;
            while (lu + 1 < ns && __builtin_amdgcn_readfirstlane(rs[lu].k0 + rs[lu].count) <= kk0) lu++;
            const int i_part = 64 * c + (int)lane;
            if (i_part < n_part[part]) {
                const int kk = kk0 + (int)lane;
                int l = lu;
                while (rs[l].k0 + rs[l].count <= kk) l++;
                const RunLds g = rs[l];
                const double tt = g.t0 + (double)(kk - g.k0) * (double)g.d;      // exact (timelaw.h)
                const double fl = floor(tt);
                const float mu = (float)(tt - fl);
                const unsigned pl = off32 + (unsigned)(int)fl;                  // position inside the block (pair)
                unsigned n = Uu > 1u ? __umulhi(pl, Minv) : pl;
                const unsigned ph = pl - n * Uu;
                unsigned half = 0u;
                if (REAL && n >= (unsigned)A) {                                 // the pair's second block: the transform's imaginary part
                    n -= (unsigned)A;
                    half = 1u;
                }
                const unsigned e = e0 + n;
                const bool wrap = ph + 1u == Uu;
                const unsigned cell = (e >> 8) * LDS_K2_STRIDE + (e & 255u);
                const unsigned dcell = wrap ? ((e & 255u) == 255u ? LDS_K2_STRIDE - 255u : 1u) : 0u;
                tab_pos[idx_base[part] + i_part] = (cell << 19) | (half << 15) | (dcell << 10) | ((wrap ? 0u : ph + 1u) << 5) | ph;
                tab_mu[idx_base[part] + i_part] = mu;
            }
        }
    }
    }                                            // (!NOPRO)
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
    E acc[KPT];
#pragma unroll
    for (int q = 0; q < KPT; q++) acc[q] = E{};
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
        // the outputs' shares of S_j.  Branch-free (round 5): every slot reads its two cells in every phase and weighs them with
        // (1 - mu) / mu where the sample's phase is j and with 0 where it is not -- the guarded form (a test, an exec mask and a
        // branch per sample and phase) cost ~600 scalar instructions per wave and block on a kernel bound by instruction issue at
        // three waves per SIMD (profiles/r05/general_rate_counters_1p77.txt).  A share weighed 0 leaves the accumulator as it was.
        const unsigned ju = (unsigned)j;
        const char *lb = reinterpret_cast<const char *>(lds);
#pragma unroll
        for (int q = 0; q < KPT; q++) {
            const unsigned pw = tab_pos[256 * q + t];
            const float mu_q = tab_mu[256 * q + t];
            const float w0 = (pw & 31u) == ju ? 1.0f - mu_q : 0.0f;                // resample.cxx:147
            const float w1 = ((pw >> 5) & 31u) == ju ? mu_q : 0.0f;
            if constexpr (!REAL) {
                const unsigned a0 = pw >> 16;                                  // byte address of the first sample
                const v2f s0 = *reinterpret_cast<const v2f *>(lb + a0);
                const v2f s1 = *reinterpret_cast<const v2f *>(lb + a0 + (((pw >> 10) & 31u) << 3));
                acc[q] = __builtin_elementwise_fma((v2f){w0, w0}, s0, acc[q]);
                acc[q] = __builtin_elementwise_fma((v2f){w1, w1}, s1, acc[q]);
            } else {
                const unsigned a0 = (pw >> 16) + ((pw >> 13) & 4u);            // ... of the half the output belongs to: + 4 bytes for the imaginary part
                const float s0 = *reinterpret_cast<const float *>(lb + a0);
                const float s1 = *reinterpret_cast<const float *>(lb + a0 + (((pw >> 10) & 31u) << 3));
                acc[q] = __builtin_fmaf(w0, s0, acc[q]);
                acc[q] = __builtin_fmaf(w1, s1, acc[q]);
            }
        }
    }
    // ---- 3. lanes = consecutive outputs
#pragma unroll
    for (int q = 0; q < KPT; q++)
        if ((int)t + 256 * q < T) __builtin_nontemporal_store(acc[q], out + k_first + (long long)t + 256 * q);
}

}  // namespace

int poly_gen_outputs_per_block(int U, int adv, float step)
{
    // consecutive outputs are >= step (1 - 2^-22) apart on the upsampled grid
    const double span = (double)U * adv;
    return (int)(span / ((double)step * (1.0 - 1.0 / 4194304.0))) + 2;
}

// SFE_ESTATE: the shape is outside what this kernel takes (the caller uses poly_seg_kernel)
// max_runs: the most runs any two (complex) / three (real) consecutive reference calls of the launch have between them
int launch_poly_gen(const PolyGenArgs &a0, int max_runs, float step, int n_channels, hipStream_t s)
{
    if (a0.n_chunks <= 0) return SFE_OK;
    PolyGenArgs a = a0;
    if (a.ovl < a.plen || a.ovl >= FFT_N / 2 || (a.ovl & 15) || max_runs > GEN_MAX_RUNS) return SFE_ESTATE;
    // the table's fields: 5 bits per phase; a call's positions as 32-bit integers
    if (a.U > 31 || (long long)a.blksize * a.U >= 0x7fffffffLL) return SFE_ESTATE;      // (31: the phase value no slot has)
    // A block owns `adv` input samples: all 4096 - ovl its transform yields validly while their outputs fit the table
    // (256 x 16; a real stream's pair of blocks: 256 x 22), fewer below that -- rates under ~1, MORE outputs than inputs: what
    // `resample` takes and `decimate` refuses, libdsp/resample.cxx:91 -- down to the reference's own limit rate = 1 / U (step 1:
    // every upsampled position an output).  The transform count per input sample grows as adv shrinks; the direct form's
    // cost grows with the OUTPUT count, faster.
    const int per_wg = a.real ? 2 : 1, kpt_max = a.real ? 22 : 16;
    a.adv = FFT_N - a.ovl;
    if (poly_gen_outputs_per_block(a.U, per_wg * a.adv, step) > 256 * kpt_max) {
        a.adv = (int)((256.0 * kpt_max - 2) * (double)step * (1.0 - 1.0 / 4194304.0) / (a.U * per_wg)) & ~15;
        if (a.adv < 512 || poly_gen_outputs_per_block(a.U, per_wg * a.adv, step) > 256 * kpt_max) return SFE_ESTATE;
    }
    if (a.blksize < a.adv) return SFE_ESTATE;                 // a block overlaps at most two reference calls, a pair three
    const int per_block = poly_gen_outputs_per_block(a.U, per_wg * a.adv, step);
    const long long A = a.adv;
    // block b owns the outputs whose first sample n = floor(P / U) lies in [b A - 1, (b + 1) A - 1): sample n_in - 1, which the
    // reference still emits from while its phase leaves a successor (resample.cxx:137-146), belongs to block floor(n_in / A) --
    // one MORE than ceil(n_in / A) blocks when n_in is a multiple of A (that output was never stored)
    const long long nblk = (a.n_in / A + 1 + per_wg - 1) / per_wg;
    if (nblk > 0x7fffffffLL) return SFE_ESTATE;
    dim3 grid((unsigned)nblk, (unsigned)n_channels), block(256);
    const int kpt = (per_block + 255) / 256;
#ifdef SFE_DIAG
    // SFE_GEN_NOPRO=1 (scripts/ab_general.py): the prologue ablation, complex float32 streams at up to nine outputs per thread only
    if (const char *e = getenv("SFE_GEN_NOPRO"))
        if (atoi(e) && !a.real && !a.in_u8 && kpt <= 9) {
            a.diag_T = per_block - 3;                     // (never more outputs than the law emits: the synthetic blocks write T each)
            hipLaunchKernelGGL((poly_gen4096_kernel<9, false, false, true>), grid, block, 0, s, a);
            SFE_HIP(hipGetLastError());
            return SFE_OK;
        }
#endif
#define SFE_GEN(KPTv, REALv)                                                                           \
    do {                                                                                              \
        if (a.in_u8) hipLaunchKernelGGL((poly_gen4096_kernel<KPTv, REALv, true>), grid, block, 0, s, a);      \
        else hipLaunchKernelGGL((poly_gen4096_kernel<KPTv, REALv, false>), grid, block, 0, s, a);            \
    } while (0)
    if (a.real) {
        if (kpt <= 12) SFE_GEN(12, true);
        else if (kpt <= 18) SFE_GEN(18, true);
        else SFE_GEN(22, true);
    } else {
        if (kpt <= 6) SFE_GEN(6, false);
        else if (kpt <= 9) SFE_GEN(9, false);
        else if (kpt <= 12) SFE_GEN(12, false);
        else SFE_GEN(16, false);
    }
#undef SFE_GEN
    SFE_HIP(hipGetLastError());
    return SFE_OK;
}

}  // namespace sfe
