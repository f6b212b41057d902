// diag/poly_gen_persistent.hip -- DIAGNOSTIC flavour only (build.py --diag): the general-rate transform kernel
// (../poly_gen.hip: poly_gen4096p_kernel) as a PERSISTENT grid that fetches the next block's rows, call records and runs
// while the current block's last inverse transform runs.  Measured and not kept (DESIGN.md 4.3b,
// profiles/r04/general_persistent_ab.txt): with its blocks dealt statically it is 12-16 % SLOWER than one workgroup per
// block (the hardware's own dispatch onto three slots per CU balances what a static deal does not), with tickets from a
// device counter it is level with it -- the two co-resident workgroups already cover a block's start-up latency.
// SFE_GEN_PERSISTENT=1 (static) / 2 (tickets; one channel) in the environment of a process that loaded the diagnostic
// library selects it (poly_gen.hip: launch_poly_gen).
// What it took to make the prefetch cost nothing (kept here because each item is a trap of its own):
//   - the no-next path must assign the prefetch registers too, or the current block's values stay live through the loop;
//   - the DFT16's constants in SGPR pairs (SFE_W16_SCALAR): hoisted out of the block loop as VGPR pairs they spill;
//   - thread-index-derived address pieces made opaque per block for the same reason;
//   - every load behind the phase's twiddle loads, in straight-line code: vector-memory results return in order, and a
//     branch around a load makes every later wait conservative (s_waitcnt vmcnt(0)) -- hence the separate EDGE launch
//     and the clamped, unconditional forms of the guarded loads.
#include <stdint.h>

#include "../common.h"
#define SFE_W16_SCALAR        // fft16.h: the DFT16's constants in SGPR pairs (this kernel loops over blocks at 168 VGPRs)
#include "../fft16.h"

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
// EDGE: the launch of the blocks whose rows are not all inside the stream -- a channel's first one (history in front) and
// its last ones (zeros behind): list entry 0 is block 0, entry e > 0 block blk_first + e - 1.  The main launch takes the
// interior blocks blk_first + e, whose rows are sixteen plain loads: with the guarded form on a branch next to them every
// wait behind the request was conservative (s_waitcnt vmcnt(0): the waits for the twiddles waited for the rows from HBM).
__device__ unsigned g_ticket, g_done;
template <int KPT, bool EDGE>
__global__ __launch_bounds__(256, KPT <= 9 ? 3 : 2) void poly_gen4096p_kernel(PolyGenArgs a)
{
    __shared__ v2f lds[FFT_ROWS * LDS_K2_STRIDE];
    __shared__ unsigned tab_pos[KPT * 256];
    __shared__ float tab_mu[KPT * 256];
    __shared__ int s_bound[8];
    const unsigned t = threadIdx.x, lo = t & 15u, hi = t >> 4;
    const unsigned w = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63u;
    const int A = FFT_N - a.ovl, U = a.U;
    const long long BU = (long long)a.blksize * U;
    const long long nblk = a.nblk;

    // ---- the grid is persistent: workgroup g takes the blocks g, g + gridDim.x, ... of the launch's (channel, block) list.
    // What a block needs from memory before it can start -- its 16 rows, the records of the (at most two) reference calls
    // it overlaps and their runs -- is requested while the PREVIOUS block's last inverse transform runs (the registers of
    // the forward spectrum and of the phase's spectrum are free by then): one workgroup per block had ~5 us of a block's
    // ~26 us in that chain of dependent loads with the workgroup's slot idle.
    auto real = [&](long long e) -> long long { return EDGE ? (e == 0 ? 0 : a.blk_first + e - 1) : a.blk_first + e; };
    long long blk = blockIdx.x;                  // entry of the launch's list
    int ch = 0;
    while (blk >= nblk) {
        blk -= nblk;
        ch++;
    }
    if (ch >= a.n_channels) return;

    // c0 = floor((Plo + 1) / BU), c1 = floor(Phi / BU): by a double-precision quotient and one correction either way (the
    // 64-bit integer divisions were ~300 scalar instructions per wave)
    auto fdiv = [&](long long x) -> long long {
        if (x < 0) return 0;                                      // block 0: Plo + 1 = 1 - U
        long long c = (long long)((float)x * __builtin_amdgcn_rcpf((float)BU));
        if (c * BU > x) c--;
        if ((c + 1) * BU <= x) c++;
        if (c * BU > x) c--;
        if ((c + 1) * BU <= x) c++;
        return c;
    };
    // positions on the upsampled grid, absolute (relative to the launch's first input sample):
    // P = in_off U + floor(t); block b owns Plo <= P < Phi.  Call c emitted the outputs with
    // c B U - 1 <= P < (c + 1) B U - 1 (its leftover output sits at relative position -1).
    auto calls_of = [&](long long b, long long *c0, long long *c1) {
        const long long Plo = (long long)U * (b * A - 1), Phi = Plo + (long long)U * A;
        long long x0 = fdiv(Plo + 1), x1 = fdiv(Phi);
        if (x1 >= a.n_chunks) x1 = a.n_chunks - 1;
        if (x0 > x1) x0 = x1;
        *c0 = x0;
        *c1 = x1;
    };
    // Everything fetched ahead comes by VECTOR loads: a scalar load counts on lgkmcnt like the LDS operations, which return in
    // order and scalar loads do not -- every LDS wait behind one would wait for it too.  The two calls' records (SegChunk, 8
    // dwords each) come one dword per lane, lanes 0-7 and 8-15 of every wave, and are read back lane by lane.
    v2f nx[16];                                  // the block's rows (thread t: samples base + t + 256 r)
    unsigned long long rw[2];                    // the runs of its calls, words t and t + 256 of (call a's ++ call b's)
    int rec;                                     // lane l < 16: dword l & 7 of call (l >> 3)'s record
    auto load_rows = [&](long long b, int c) {
        const v2f *in = static_cast<const v2f *>(a.in) + (size_t)c * a.in_stride;
        const v2f *hist = static_cast<const v2f *>(a.hist) + (size_t)c * a.hl;
        const long long base = b * A - a.ovl;
        unsigned tt = t;                         // opaque: what the addresses share is computed here, not kept from block to block
        asm volatile("" : "+v"(tt));
        if constexpr (!EDGE) {
#pragma unroll
            for (int r = 0; r < 16; r++) nx[r] = __builtin_nontemporal_load(in + base + 256 * r + tt);
        } else {
            // sixteen UNCONDITIONAL loads here too, from a clamped address: no branch behind the request
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const long long i = base + 256 * r + (long long)tt;
                const bool ok_in = i >= 0 && i < a.n_in, ok_h = i < 0 && i >= -(long long)a.hl;
                const v2f *p = ok_in ? in + i : (ok_h ? hist + (a.hl + i) : in);
                const v2f x = *p;
                nx[r] = ok_in || ok_h ? x : (v2f){0.0f, 0.0f};
            }
        }
    };
    auto load_records = [&](long long c0, long long c1) {          // (every lane loads: no branch around the load)
        rec = *(reinterpret_cast<const int *>(a.chunks + ((lane & 8u) ? c1 : c0)) + (lane & 7u));
    };
    auto load_runs = [&](int seg_a, int n_a, int seg_b, int n_b) {      // n_b = 0: the block lies inside one call
        const unsigned long long *ga = reinterpret_cast<const unsigned long long *>(static_cast<const RunLds *>(a.segs) + seg_a);
        const unsigned long long *gb = reinterpret_cast<const unsigned long long *>(static_cast<const RunLds *>(a.segs) + seg_b);
        unsigned tt = t;
        asm volatile("" : "+v"(tt));
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int i = (int)tt + 256 * k;
            rw[k] = *(i < 3 * n_a ? ga + i : (i < 3 * (n_a + n_b) ? gb + (i - 3 * n_a) : ga));      // (words behind the runs: not used)
        }
    };
    int meta;                                    // lane l < 4: seg_first (l & 1 = 0) / n_seg (1) of call (l >> 1): where its runs are
    auto load_meta = [&](long long c0, long long c1) {
        meta = *(reinterpret_cast<const int *>(a.chunks + ((lane & 2u) ? c1 : c0)) + 6 + (lane & 1u));
    };

    // ---- twiddle bases (fir_fft.hip: W^(e (4a + b)) = q[a] p[b]), re-read where a stage needs them: the
    // index is made opaque each time so that the loads are not hoisted back out of the loops into 24 resident registers
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
    const unsigned base_b = hi * LDS_K2_STRIDE + lo, base_c = hi * LDS_K2_STRIDE + lo * LDS_K1_STRIDE;
    const unsigned Uu = (unsigned)U, Minv = Uu > 1u ? 0xFFFFFFFFu / Uu + 1u : 0u;      // floor(x / U) = mulhi(x, Minv), x < U A
    const unsigned e0 = (unsigned)a.ovl - 1u;    // transform element of a block's first owned input sample

    // ---- the first block's loads
    {
        long long c0, c1;
        calls_of(real(blk), &c0, &c1);
        load_rows(real(blk), ch);
        const SegChunk ca = a.chunks[c0], cb = a.chunks[c1];
        load_records(c0, c1);
        load_runs(ca.seg_first, ca.n_seg, cb.seg_first, c1 > c0 ? cb.n_seg : 0);
    }

#pragma unroll 1
    for (;;) {
        // ---- the block after this one
        // tickets (experiment; one channel): the first gridDim.x entries are the static ones, the rest drawn
        __shared__ unsigned s_next;
        unsigned tk = 0;
        const bool tickets = !EDGE && a.tickets;
        if (tickets && t == 0) tk = atomicAdd(&g_ticket, 1u) + gridDim.x;
        long long nblk_b = blk + gridDim.x;      // EDGE: static
        int nch = ch;
        if (!tickets) {
            while (nblk_b >= nblk) {
                nblk_b -= nblk;
                nch++;
            }
        }
        bool has_next = nch < a.n_channels;

        // ---- 0. which outputs are this block's, and where each of them sits
        long long c0, c1;
        const long long rblk = real(blk);
        calls_of(rblk, &c0, &c1);
        const long long Plo = (long long)U * (rblk * A - 1), Phi = Plo + (long long)U * A;
        SegChunk ca, cb;
        {
            int r[16];
#pragma unroll
            for (int k = 0; k < 16; k++) r[k] = __builtin_amdgcn_readlane(rec, k);
            auto rd = [&](int o, SegChunk *c) {
                c->in_off = (long long)(((unsigned long long)(unsigned)r[o + 1] << 32) | (unsigned)r[o]);
                c->k_first = (long long)(((unsigned long long)(unsigned)r[o + 3] << 32) | (unsigned)r[o + 2]);
                c->m = r[o + 4];
                c->n_out = r[o + 5];
                c->seg_first = r[o + 6];
                c->n_seg = r[o + 7];
            };
            rd(0, &ca);
            rd(8, &cb);
        }
        RunLds *runs = reinterpret_cast<RunLds *>(lds);
        const int na = ca.n_seg, nb = c1 > c0 ? cb.n_seg : 0;         // host guarantees na + nb <= GEN_MAX_RUNS
        lds_barrier();                           // every wave is done with the previous block's S and table
        {
            unsigned long long *ws = reinterpret_cast<unsigned long long *>(lds);
            ws[t] = rw[0];
            ws[t + 256] = rw[1];
            if (3 * (na + nb) > 512) {           // more runs than were fetched ahead (rates with rounding ties)
                const unsigned long long *ga = reinterpret_cast<const unsigned long long *>(static_cast<const RunLds *>(a.segs) + ca.seg_first);
                const unsigned long long *gb = reinterpret_cast<const unsigned long long *>(static_cast<const RunLds *>(a.segs) + cb.seg_first);
                for (int i = (int)t + 512; i < 3 * (na + nb); i += 256) ws[i] = i < 3 * na ? ga[i] : gb[i - 3 * na];
            }
        }
        lds_barrier();
        // outputs of a call whose relative position is < bound (uniform: every thread runs the same search); *run: the run
        // that holds that output (or the one behind the last)
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
        // four bounds, four waves: wave w finds ONE of them (the search is uniform inside a wave: every lane of all four
        // waves running all four searches was a seventh of the kernel's vector instructions) and lane 0 publishes it
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

        // The table, [output of the block]: the byte address in the exchange buffer of the output's first phase sample << 16
        // | (cells to its second one: 0 the next phase of the same input sample, 1 / 17 phase 0 of the next one) << 10 | the
        // second sample's phase << 5 | the first one's; and mu.  Filled 64 consecutive outputs of ONE call at a time (wave w:
        // every fourth such piece), so that what a piece needs to know about its call is uniform and scalar: the run of its
        // first output is found by walking on from the wave's previous piece, each lane then walks on to its own (pieces
        // hold one to a few runs, except where a call starts: there a dozen binades pass in as many outputs).
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

        // ---- 1. forward transform: F1 over n2, F2 over n1, F3 over n0 -> bin k of this thread in X[k]
        load_tw1();
        load_tw2();
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
        // the first phase's spectrum: requested here, it lands under the last forward stage; phase j + 1's is requested as
        // soon as phase j's has been multiplied in (16 loads from L2 per phase that nothing waits for)
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
        auto phase = [&](const int j, auto &&after_product, auto &&after_first_stage, auto &&before_last_stage) {
            lds_barrier();                           // (j > 0: every thread is done reading S_{j-1})
#pragma unroll
            for (int k = 0; k < 16; k++) v[P16(k)] = cmul(X[P16(k)], hn[k]);
            after_product();
            load_tw1();                              // for this phase's last stage
            dft16_rev<+1>(v);
#pragma unroll
            for (int k = 0; k < 16; k++) lds[base_c + k] = v[k];
            load_tw2();                              // for the next stage (v is in LDS: the registers are there)
            after_first_stage();
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
            before_last_stage();
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
        };
#pragma unroll 1
        for (int j = 0; j + 1 < U; j++) phase(j, [&]() { load_h(j + 1); }, []() {}, []() {});
        // the last phase: X and its spectrum are dead once multiplied -- the next block's rows and call records go there, and
        // a stage on, when the records' (seg_first, n_seg) have arrived, the calls' runs.  They are requested BEHIND this
        // phase's twiddle loads: vector-memory results return in order, a wait for the twiddles issued behind sixteen rows
        // from HBM would wait for the rows (the first version had it that way: 16 % slower than one workgroup per block).
        // Straight-line: a workgroup's last block requests itself again rather than branch around the loads (a branch would
        // make every later wait conservative: s_waitcnt vmcnt(0)).
        long long nc0 = 0, nc1 = 0;
        if (tickets) {
            if (t == 0) s_next = tk;
            lds_barrier();
            nblk_b = __builtin_amdgcn_readfirstlane(s_next);
            has_next = nblk_b < nblk;
        }
        const long long pblk = real(has_next ? nblk_b : blk);
        const int pch = has_next ? nch : ch;
        phase(U - 1, []() {},
              [&]() {
                  calls_of(pblk, &nc0, &nc1);
                  load_meta(nc0, nc1);           // (first: the wait for it, a stage on, then leaves the rows in flight)
                  load_records(nc0, nc1);
                  load_rows(pblk, pch);
              },
              [&]() {
                  asm volatile("" : "+v"(meta));      // (here, not earlier: the scheduler had moved the read -- and its wait -- up a stage)
                  const int sa = __builtin_amdgcn_readlane(meta, 0), n_a = __builtin_amdgcn_readlane(meta, 1);
                  const int sb = __builtin_amdgcn_readlane(meta, 2), n_b = __builtin_amdgcn_readlane(meta, 3);
                  load_runs(sa, n_a, sb, nc1 > nc0 ? n_b : 0);
              });
        // ---- 3. lanes = consecutive outputs
        v2f *out = static_cast<v2f *>(a.out) + (size_t)ch * a.out_stride;
#pragma unroll
        for (int q = 0; q < KPT; q++)
            if ((int)t + 256 * q < T) __builtin_nontemporal_store(acc[q], out + k_first + (long long)t + 256 * q);
        if (!has_next) break;
        blk = nblk_b;
        ch = nch;
    }
    if (!EDGE && a.tickets && t == 0 && atomicAdd(&g_done, 1u) == gridDim.x - 1) {
        g_ticket = 0;
        g_done = 0;
        __threadfence();
    }
}

}  // namespace

// SFE_ESTATE: the shape is outside what this kernel takes (the caller uses poly_seg_kernel)
int poly_gen_outputs_per_block(int U, int adv, float step);

int launch_poly_gen_persistent(const PolyGenArgs &a, int max_runs_two_calls, float step, int n_channels, hipStream_t s, int tickets)
{
    if (a.n_chunks <= 0) return SFE_OK;
    if (a.ovl < a.plen || a.ovl >= FFT_N / 2 || (a.ovl & 15) || a.blksize < FFT_N - a.ovl || max_runs_two_calls > GEN_MAX_RUNS)
        return SFE_ESTATE;
    // the table's fields: 5 bits per phase; a call's positions as 32-bit integers
    if (a.U > 32 || (long long)a.blksize * a.U >= 0x7fffffffLL) return SFE_ESTATE;
    const int per_block = poly_gen_outputs_per_block(a.U, FFT_N - a.ovl, step);
    if (per_block > 256 * 16) return SFE_ESTATE;
    const long long A = FFT_N - a.ovl;
    const long long nblk = (a.n_in + A - 1) / A;
    if (nblk * n_channels > 0x7fffffffLL) return SFE_ESTATE;
    const int kpt = (per_block + 255) / 256;
    // interior blocks 1 .. n_int (every row inside the stream: b A - ovl >= 0, b A - ovl + 4096 <= n_in), edge blocks 0 and
    // n_int + 1 .. nblk - 1; both launches persistent: as many workgroups as are resident at once (3 per CU up to 9 outputs
    // per thread, else 2), workgroup g taking the list entries g, g + grid, ...
    long long n_int = (a.n_in + a.ovl - FFT_N) / A;
    if (a.n_in + a.ovl < FFT_N || n_int < 0) n_int = 0;
    if (n_int > nblk - 1) n_int = nblk - 1;
    const long long resident = (long long)device_cu_count() * (kpt <= 9 ? 3 : 2);
    auto launch = [&](bool edge) {
        PolyGenArgs b = a;
        b.n_channels = n_channels;
        b.tickets = tickets && n_channels == 1;
        b.blk_first = edge ? n_int + 1 : 1;
        b.nblk = edge ? nblk - n_int : n_int;
        const long long total = b.nblk * n_channels;
        if (total <= 0) return;
        dim3 grid((unsigned)(total < resident ? total : resident)), block(256);
        if (edge) {
            if (kpt <= 6) hipLaunchKernelGGL((poly_gen4096p_kernel<6, true>), grid, block, 0, s, b);
            else if (kpt <= 9) hipLaunchKernelGGL((poly_gen4096p_kernel<9, true>), grid, block, 0, s, b);
            else if (kpt <= 12) hipLaunchKernelGGL((poly_gen4096p_kernel<12, true>), grid, block, 0, s, b);
            else hipLaunchKernelGGL((poly_gen4096p_kernel<16, true>), grid, block, 0, s, b);
        } else {
            if (kpt <= 6) hipLaunchKernelGGL((poly_gen4096p_kernel<6, false>), grid, block, 0, s, b);
            else if (kpt <= 9) hipLaunchKernelGGL((poly_gen4096p_kernel<9, false>), grid, block, 0, s, b);
            else if (kpt <= 12) hipLaunchKernelGGL((poly_gen4096p_kernel<12, false>), grid, block, 0, s, b);
            else hipLaunchKernelGGL((poly_gen4096p_kernel<16, false>), grid, block, 0, s, b);
        }
    };
    launch(false);
    launch(true);
    SFE_HIP(hipGetLastError());
    return SFE_OK;
}

}  // namespace sfe
