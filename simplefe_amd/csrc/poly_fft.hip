// poly_fft.hip -- rational resampling / decimation by an integer step, in the transform domain.
//
// The law (libdsp/resample.cxx:119-150, integer-valued step, mu == 0) in the tiled form of
// polyphase.hip: output UP*m + r = sum_d g_r[d] x[SP*m + e_max - d], d < Lp.  Split d = SP*i + c:
//     y_r[m] = sum_c (h_rc * u_c)[m],   u_c[m] = x[SP*m + e_max - c],   h_rc[i] = g_r[SP*i + c]
// -- UP*SP short filters (Li = Lp/SP taps) running at the LOW rate.  Overlap-save with 256-point
// transforms: per segment of V = 257 - Li low-rate points, SP forward FFTs (one per input polyphase
// component), a UP x SP matrix multiply per bin, UP inverse FFTs.  For the 5/3, 381-tap shape that
// is ~100 flop per input sample against ~300 for the direct dot products: the kernel is bound by
// the sample stream, not by the vector ALU.
//
// Mapping.  A 256-point FFT is 16 x 16: sixteen lanes with 16 complex registers each and ONE
// exchange through LDS that stays inside the wave (no workgroup barrier).  A workgroup is 16 such
// lane groups and processes a PASS of R segments: R*SP groups run forward transforms of this
// pass's segments while R*UP groups run the inverse transforms of the previous pass's segments
// (software pipeline), all executing the same forward-FFT instruction stream -- the inverse is
// the forward transform read out at index (256 - n) mod 256, so it costs no arithmetic.  Between
// the two, thread b of the workgroup owns bin b: it holds the UP*SP spectra H_rc[b]/256 in
// registers for the life of the (persistent) workgroup and turns the R*SP forward results into
// R*UP inverse inputs.
//   S0  all threads: the pass's R*SP*256 input samples (loaded coalesced during the last pass's
//       S3) are scattered into the forward jobs' LDS areas by polyphase component -- the
//       transpose happens on the LDS write
//   S1  every group: 16 reads of its job's area, DFT16, twiddle, exchange, DFT16
//   S2  forward groups: spectrum -> the job's area;  inverse groups: store y_r[m] (a segment's
//       UP phases sit in one wave, so a store instruction covers whole lines of out[UP*m + r])
//   S3  bin owners: Y_r = sum_c H_rc X_c for the pass's segments -> the inverse jobs' areas
// Three workgroup barriers per pass; 35.6 KiB LDS (16 job areas + the per-lane twiddle bases),
// <= 128 VGPRs -> 4 workgroups per CU.  DESIGN.md section 4.2c has the measurements.
#include <stdint.h>
#include <stdlib.h>

#include <atomic>

#include "common.h"
#include "fft16.h"

namespace sfe {
namespace {

constexpr int PF_M = 256;
constexpr int PF_AREA = 272;      // cells per lane group: 16 rows of 17 (exchange), 256 used otherwise

// staged-sample swizzle: sample k of component c sits in its group's area at cell k ^ pf_swz(c).
// ds_write_b64 is banked per 16 lanes over 32 banks and area bases are multiples of 32 banks, so
// without it the scatter of S0 (16 consecutive lanes = 5 components x 3-4 consecutive k) lands
// 3-4 deep on the same banks.  Only the low four bits move, i.e. a sample stays in its row of 16:
// the group's row reads (ds_read_b64, 32 lanes = two groups, 64 banks) stay conflict-free.
__host__ __device__ constexpr unsigned pf_swz(unsigned c) { return c < 4 ? 4u * c : 2u + 4u * (c - 4); }
static_assert(pf_swz(7) < 16, "the swizzle must stay inside a row of 16");

// Round 2: for the coalesced scatter (thread t stages sample t + 256 i) an ADDITIVE rotation inside
// the row of 16 makes it conflict-free where the XOR cannot: sample j = SP k + c goes to cell
// (k + rot(c)) mod 16 of its row, and rot is chosen so that this equals a j mod 16 with a odd -- 16
// consecutive lanes then hit 16 different cells.  For odd SP a = SP^-1 mod 16 (rot(c) = a c); the even
// ones were found by search.  The XOR form left every 16-lane ds_write_b64 group of the 5/3 shape
// 2-way conflicted: SQ_LDS_BANK_CONFLICT = 160 cycles per pass = 9.8 % of the kernel's LDS cycles
// (profiles/r02/resample_sq_counters.txt: 18.6 M = 116 207 passes x 160, to the count).  The u8
// wide-lane scatter (8 consecutive samples per lane) keeps the XOR form: under the rotation its
// lanes would fall 8 deep on two cells.
template <int SP> struct PfRot;
template <> struct PfRot<1> { [[maybe_unused]] static constexpr unsigned v[1] = {0}; };
template <> struct PfRot<2> { static constexpr unsigned v[2] = {0, 8}; };
template <> struct PfRot<3> { static constexpr unsigned v[3] = {0, 11, 6}; };
template <> struct PfRot<4> { static constexpr unsigned v[4] = {0, 4, 8, 12}; };
template <> struct PfRot<5> { static constexpr unsigned v[5] = {0, 13, 10, 7, 4}; };
template <> struct PfRot<6> { static constexpr unsigned v[6] = {0, 8, 3, 11, 6, 14}; };
template <> struct PfRot<7> { static constexpr unsigned v[7] = {0, 7, 14, 5, 12, 3, 10}; };
template <> struct PfRot<8> { static constexpr unsigned v[8] = {0, 2, 4, 6, 8, 10, 12, 14}; };
// cell (inside the group's area) of staged sample k of component c
template <int SP, bool ROT>
__device__ __forceinline__ unsigned pf_cell(unsigned c, unsigned k)
{
    if constexpr (ROT) {
        unsigned r = 0;
#pragma unroll
        for (int q = 0; q < SP; q++) r = c == (unsigned)q ? PfRot<SP>::v[q] : r;
        return (k & ~15u) | ((k + r) & 15u);
    } else {
        return k ^ pf_swz(c);
    }
}

// PAIR (real data): the real and imaginary parts of a transform carry two CONSECUTIVE real
// segments of the stream (the sub-filters are real, so they stay apart): a real stream costs
// what a complex one does per sample pair.
// DIAG (instantiated under -DSFE_DIAG only, scripts/ablate.py): bit 0 = input loads replaced by
// constants, bit 1 = output stores folded into one never-taken store, bit 2 = no spectrum stage.
// TICKET: passes are drawn from per-XCD work counters instead of walked at a fixed stride
// (fir_fft.hip has the reasoning and the measurements), channel-major over all channels.
// LATE (diagnostic): 1 = the next pass's samples are requested after S3 instead of before it,
// 2 = the first half of its segments before S3 and the second half after, 3 = a whole pass AHEAD: requested right
// after S0 has staged the current pass (the draw runs one pass further ahead), held in registers through S1..S3.
template <int SP, int UP, int R, bool IN_U8, bool PAIR, int DIAG = 0, bool TICKET = false, int LATE = 0, int WPS = 4>
__global__ __launch_bounds__(256, WPS) void poly_fft256_kernel(PolyFftArgs a)
{
    constexpr int F = R * SP, I = R * UP;
    static_assert(F + I <= 16, "one lane group per transform");
    static_assert(SP <= 8, "pf_swz covers components 0..7");
    __shared__ v2f lds[16 * PF_AREA + 96];     // 16 group areas + the six twiddle bases per lane
    __shared__ unsigned s_next;
    const unsigned t = threadIdx.x, l = t & 15u, g = t >> 4;
    // TICKET: one grid dimension, passes of ALL channels drawn channel-major (ticket k = channel k / n_pass,
    // pass k % n_pass); otherwise blockIdx.y is the workgroup's channel for good
    int ch = TICKET ? 0 : blockIdx.y;          // channel of the pass being staged / transformed forward
    // Which transform a lane group runs.  Areas are indexed by JOB (forward job f = seg*SP + c' ->
    // area f, inverse job v = seg*UP + r -> area F + v).  With UP > 1 the UP inverse jobs of a
    // segment sit in ONE wave (the top UP groups of wave 3 - seg), so that a store instruction of
    // that wave covers out[UP*m + r] for all r: contiguous 128-byte lines instead of every UP-th
    // 8 bytes; forward jobs fill the remaining groups in order.
    bool is_fwd, is_inv;
    unsigned job;
    if constexpr (UP > 1) {
        static_assert(R <= 4 && UP <= 4, "one wave hosts one segment's inverse transforms");
        const unsigned w = g >> 2, pos = g & 3u, sw = 3u - w;
        is_inv = sw < (unsigned)R && pos >= (unsigned)(4 - UP);
        const unsigned f = g - UP * (w > (unsigned)(4 - R) ? w - (4 - R) : 0u);
        is_fwd = !is_inv && f < (unsigned)F;
        job = is_inv ? sw * UP + (pos - (4 - UP)) : f;
    } else {
        is_fwd = g < (unsigned)F;
        is_inv = !is_fwd && g < (unsigned)(F + I);
        job = is_fwd ? g : g - F;
    }
    const unsigned seg = is_fwd ? job / SP : job / UP;       // segment of the pass this group works on
    const unsigned comp = is_fwd ? job % SP : job % UP;      // input component c' / output phase r
    const unsigned area = is_fwd ? job : (is_inv ? F + job : 15u);   // idle groups: a harmless area of their own
    unsigned cell0 = area * PF_AREA + l;                     // this lane's column of its job's area
    constexpr bool ROT = !(IN_U8 && !PAIR && 32 * SP + 1 <= 256) && !(DIAG & 8);      // == !WIDE (defined below); DIAG bit 3: the round-1 XOR layout
    unsigned cell_in = area * PF_AREA + (is_fwd ? pf_cell<SP, ROT>(comp, l) : l);

    constexpr int ISZ = (IN_U8 ? 2 : 8) / (PAIR ? 2 : 1);    // bytes per input sample
    constexpr int ESZ = PAIR ? 4 : 8;                         // bytes per float32 sample (history, output)
    auto in_of = [&](int c) -> const char * { return static_cast<const char *>(a.in) + (size_t)c * a.in_stride * ISZ; };
    auto out_of = [&](int c) -> char * { return static_cast<char *>(a.out) + (size_t)c * a.out_stride * ESZ; };
    auto hist_of = [&](int c) -> const char * { return static_cast<const char *>(a.hist) + (size_t)c * a.hl * ESZ; };

    // DIAG bits 7 / 8 (diagnostic library only): static priority for the waves that host the inverse transforms and their
    // global stores (waves 2, 3 of the 5/3 shape) / for the forward-only waves (MI355X_MICROARCH.md, two waves per SIMD, item 4)
    if constexpr (DIAG & 128) { if (t >= 128u) __builtin_amdgcn_s_setprio(1); }
    if constexpr (DIAG & 256) { if (t < 128u) __builtin_amdgcn_s_setprio(1); }
    const v2f *tw = reinterpret_cast<const v2f *>(a.tw);
    // the six twiddle bases per lane W_256^(l k), W_256^(4 l k) sit in LDS and are re-read every
    // pass (6 ds_read_b64): 12 VGPRs the prefetched samples need more (measured +4 % over
    // registers, which spill at <= 128 VGPRs)
    if (t < 96) lds[16 * PF_AREA + t] = tw[t];
    v2f G[UP][SP];     // this thread's bin of every sub-filter spectrum
#pragma unroll
    for (int r = 0; r < UP; r++)
#pragma unroll
        for (int c = 0; c < SP; c++) {
            G[r][c] = reinterpret_cast<const v2f *>(a.H)[(r * SP + c) * PF_M + t];
            asm volatile("" : "+v"(G[r][c]));
        }

    // one complex sample (or, PAIR, one real sample in .x) at p[lane]
    auto load_in = [&](const char *p, unsigned lane) -> v2f {
        if constexpr (PAIR) {
            if constexpr (IN_U8) return (v2f){u8_to_f32(__builtin_nontemporal_load(reinterpret_cast<const unsigned char *>(p) + lane)), 0.0f};
            else return (v2f){__builtin_nontemporal_load(reinterpret_cast<const float *>(p) + lane), 0.0f};
        } else if constexpr (IN_U8) {   // wire format: (b - 128) / 127 on load (gr-simplefe/lib/source_c_impl.cc:121-132)
            const unsigned w = __builtin_nontemporal_load(reinterpret_cast<const unsigned short *>(p) + lane);
            return (v2f){u8_to_f32(w & 0xFFu), u8_to_f32(w >> 8)};
        } else {
            return __builtin_nontemporal_load(reinterpret_cast<const v2f *>(p) + lane);
        }
    };
    // sample idx of the virtual stream history ++ input ++ zeros (edge segments only)
    auto load_guarded = [&](const char *in_c, const char *hist_c, long long idx) -> v2f {
        if (idx >= 0) return idx < a.n_in ? load_in(in_c + idx * ISZ, 0u) : (v2f){0.0f, 0.0f};
        if (idx + a.hl < 0) return (v2f){0.0f, 0.0f};
        if constexpr (PAIR) return (v2f){reinterpret_cast<const float *>(hist_c)[idx + a.hl], 0.0f};
        else return reinterpret_cast<const v2f *>(hist_c)[idx + a.hl];
    };
    // the R*SP*256 staged samples of a pass, thread t: sample t + 256 i of each segment.  PAIR:
    // transform (pass*R + sg) carries real segments 2*(pass*R + sg) in .x and the next one in .y.
    auto load_pass = [&](v2f (&s)[R * SP], const char *in_c, long long pass, int sg_lo = 0, int sg_hi = R) {
        if constexpr (DIAG & 1) {       // ablation: no input loads
            unsigned u = t + ((unsigned)pass << 12);
            asm volatile("" : "+v"(u));
#pragma unroll
            for (int i = 0; i < R * SP; i++) s[i] = (v2f){__builtin_bit_cast(float, 0x3f000000u | ((u + 256u * i) & 0x7fffffu)), 0.25f};
            return;
        }
#pragma unroll
        for (int sg = 0; sg < R; sg++) {
            if (sg < sg_lo || sg >= sg_hi) continue;
            // stream index of the segment's staged sample 0 (uniform)
            const long long sidx = (pass * R + sg) * (PAIR ? 2 : 1);
            const long long start = (sidx * a.V - a.ovl) * SP + a.e_max - (SP - 1);
            // interior passes only (pass_interior): edge passes are staged straight into LDS in S0,
            // so that their guarded 64-bit addressing is never live beside the prefetched samples and
            // the spectra registers (it cost 14 spilled VGPRs in EVERY pass, round 1)
            const char *p = in_c + start * ISZ;
#pragma unroll
            for (int i = 0; i < SP; i++) {
                // the pass's first and last rows are what its neighbours read too: they may stay in the L2
                if (!PAIR && !IN_U8 && ((sg == 0 && i == 0) || (sg == R - 1 && i == SP - 1)) && a.halo_keep)
                    s[sg * SP + i] = reinterpret_cast<const v2f *>(p)[t + 256u * i];
                else
                    s[sg * SP + i] = load_in(p, t + 256u * i);
                if constexpr (PAIR) s[sg * SP + i].y = load_in(p + (long long)a.V * SP * ISZ, t + 256u * i).x;
            }
        }
    };
    auto pass_interior = [&](long long pass) -> bool {
        if constexpr (DIAG & 1) return true;
        const long long s0 = ((pass * R) * (PAIR ? 2 : 1) * a.V - a.ovl) * SP + a.e_max - (SP - 1);
        const long long s1 = ((pass * R + R - 1) * (PAIR ? 2 : 1) * a.V - a.ovl) * SP + a.e_max - (SP - 1);
        const long long span = (long long)PF_M * SP + (PAIR ? (long long)a.V * SP : 0);
        return s0 >= 0 && s1 + span <= a.n_in;
    };
    // an edge pass (history in front, ragged end): every staged sample guarded, one at a time, into its cell
    auto stage_edge_pass = [&](int c, long long pass, auto cell_of) {
        const char *in_c = in_of(c), *hist_c = hist_of(c);
#pragma unroll 1
        for (int sg = 0; sg < R; sg++) {
            const long long sidx = (pass * R + sg) * (PAIR ? 2 : 1);
            const long long start = (sidx * a.V - a.ovl) * SP + a.e_max - (SP - 1);
#pragma unroll 1
            for (int i = 0; i < SP; i++) {
                const unsigned j = t + 256u * i;
                const long long idx = start + (long long)j;
                v2f v = load_guarded(in_c, hist_c, idx);
                if constexpr (PAIR) v.y = load_guarded(in_c, hist_c, idx + (long long)a.V * SP).x;
                lds[(sg * SP + j % SP) * PF_AREA + cell_of(j % SP, j / SP)] = v;
            }
        }
    };

    // WIDE (u8 complex input): a pass's samples are requested as 16-byte lanes (8 samples each,
    // from the 16-byte boundary at or below the segment's start) instead of 2-byte lanes -- a
    // tenth of the load instructions and whole-line requests -- kept raw in 4 VGPRs per segment
    // and converted when they are scattered.  Edge passes (history, stream end, unaligned base)
    // take the per-sample path, synchronously.
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    constexpr int NCHUNK = 32 * SP + 1;                       // 16-byte chunks covering 256*SP samples at any phase
    constexpr bool WIDE = IN_U8 && !PAIR && NCHUNK <= 256;
    auto seg_start = [&](long long pass, int sg) -> long long {
        return ((pass * R + sg) * a.V - a.ovl) * SP + a.e_max - (SP - 1);
    };
    auto pass_is_wide = [&](int c, long long pass) -> bool {
        if (!WIDE || (reinterpret_cast<uintptr_t>(in_of(c)) & 15u) != 0) return false;
        const long long a0 = seg_start(pass, 0) & ~7LL, a1 = seg_start(pass, R - 1) & ~7LL;
        return a0 >= 0 && a1 + 8LL * NCHUNK <= a.n_in;
    };
    auto load_pass_wide = [&](v4u (&raw)[R], const char *in_c, long long pass) {
#pragma unroll
        for (int sg = 0; sg < R; sg++) {
            const long long a0 = seg_start(pass, sg) & ~7LL;
            if (t < (unsigned)NCHUNK) raw[sg] = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(in_c + a0 * 2) + t);
        }
    };

    // work counters: group g = blockIdx.x % tgroups draws the passes g, g + tgroups, ... from its own
    // counter (128 bytes apart); the launch's last draw of a group zeroes it for the next launch
    const unsigned tg = a.tgroups, grp = TICKET ? blockIdx.x % tg : 0u;
    const unsigned npc32 = (unsigned)a.n_pass;                            // passes per channel
    const unsigned np32 = TICKET ? a.total : npc32;                       // tickets in the launch: all channels' passes
    // (runs of Q = 2^tqs consecutive passes per counter, as fir_fft.hip: neighbours' overlaps meet in one XCD's L2)
    const unsigned Q = 1u << a.tqs, row = tg << a.tqs, rem = TICKET ? np32 % row : 0u;
    const unsigned mine = TICKET ? (np32 / row << a.tqs) + (rem > grp * Q ? (rem - grp * Q < Q ? rem - grp * Q : Q) : 0u) : 0u;
    const unsigned last_draw = TICKET ? mine + (gridDim.x - grp + tg - 1u) / tg - 1u : 0u;
    // issued early, looked at late (fir_fft.hip: draw_issue / draw_finish has the reasoning)
    auto draw_issue = [&]() -> unsigned {
        return __hip_atomic_fetch_add(a.ticket + 32u * grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto draw_finish = [&](unsigned c) -> unsigned {
        unsigned *const ctr = a.ticket + 32u * grp;
        if (c == last_draw) __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long k = (((unsigned long long)(c >> a.tqs) * tg + grp) << a.tqs) + (c & (Q - 1u));
        return k < np32 ? (unsigned)k : 0xFFFFFFFFu;
    };
    auto draw = [&]() -> unsigned { return draw_finish(draw_issue()); };
    // ticket -> (channel, pass of that channel); "no ticket" -> pass = n_pass
    auto decode = [&](unsigned k, int &c, long long &p) {
        if (k == 0xFFFFFFFFu) {
            p = a.n_pass;
        } else {
            const unsigned cc = k / npc32;
            c = (int)cc;
            p = (long long)(k - cc * npc32);
        }
    };
    long long first = blockIdx.x;
    if constexpr (TICKET) {
        if (t == 0) s_next = draw();
        lds_barrier();
        decode(__builtin_amdgcn_readfirstlane(s_next), ch, first);
        lds_barrier();
    }
    // state carry-over fused in: the workgroup that takes a channel's pass 0 also writes the
    // NEXT call's history, in[n_in - hl .. n_in) as float32 -- one launch per call instead of two
    auto carry_history = [&](int c) {
        const char *src = in_of(c) + (a.n_in - a.hl) * ISZ;
        char *ho = static_cast<char *>(a.hist_out) + (size_t)c * a.hl * ESZ;
#pragma unroll 1
        for (unsigned i = t; i < (unsigned)a.hl; i += 256u) {
            const v2f smp = load_in(src, i);
            if constexpr (PAIR) reinterpret_cast<float *>(ho)[i] = smp.x;
            else reinterpret_cast<v2f *>(ho)[i] = smp;
        }
    };
    // LATE == 3: the pass after `first` is known before the loop starts (and every later one a pass ahead)
    [[maybe_unused]] long long nx = LATE == 3 ? first + gridDim.x : 0;
    [[maybe_unused]] int nxch = ch;
    if constexpr (TICKET && LATE == 3) {
        if (t == 0) s_next = first < a.n_pass ? draw() : 0xFFFFFFFFu;
        lds_barrier();
        decode(__builtin_amdgcn_readfirstlane(s_next), nxch, nx);
        lds_barrier();
    }
    long long prev = -1;       // pass whose inverse transforms run in this iteration
    int pch = ch;              // ... and its channel
    v2f s[R * SP];
    v4u raw[R];
    // passes are dealt so that at any moment the resident workgroups read one compact window of the
    // stream.  Giving each WORKGROUP a contiguous run of passes (so the overlap re-read hits L2)
    // measured 8 % SLOWER: a thousand separate read/write streams cost HBM more than the 5 % of
    // re-read bytes they save.  Runs of eight per COUNTER (round 2) keep the window compact and
    // still put neighbours on one XCD.
    bool cur_wide = false;      // this pass's samples were requested ahead: as wide raw lanes (WIDE) ...
    bool cur_fast = false;      // ... or as per-sample registers s[] (interior pass of a non-WIDE kernel)
    if (first < a.n_pass) {
        cur_wide = pass_is_wide(ch, first);
        if (cur_wide) load_pass_wide(raw, in_of(ch), first);
        else if (!WIDE) {
            cur_fast = pass_interior(first);
            if (cur_fast) load_pass(s, in_of(ch), first);
        }
    }
    for (long long pass = first;;) {
        const bool cur = pass < a.n_pass;
        if (!cur && prev < 0) break;
        // ---- S0: stage this pass's input (requested during the last iteration's S3), transposed by component
        if (cur && WIDE && cur_wide) {
            unsigned tt = t;
            asm volatile("" : "+v"(tt));
            if (tt < (unsigned)NCHUNK) {
#pragma unroll
                for (int sg = 0; sg < R; sg++) {
                    const int delta = (int)(seg_start(pass, sg) & 7LL);       // uniform
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        const unsigned w = raw[sg][e >> 1] >> (16 * (e & 1));
                        const unsigned j = 8u * tt + e - (unsigned)delta;      // wraps for the samples before the start
                        if (j < 256u * SP) {
                            const unsigned c = j % SP, k = j / SP;
                            lds[(sg * SP + c) * PF_AREA + pf_cell<SP, ROT>(c, k)] = (v2f){u8_to_f32(w & 0xFFu), u8_to_f32((w >> 8) & 0xFFu)};
                        }
                    }
                }
            }
        } else if (cur && !cur_fast) {
            // an edge pass: nothing was requested ahead
            stage_edge_pass(ch, pass, [](unsigned c, unsigned k) { return pf_cell<SP, ROT>(c, k); });
        } else if (cur) {
            unsigned tt = t;
            asm volatile("" : "+v"(tt));      // recompute the scatter cells here instead of keeping R*SP of them live
            // low four bits of the cell: with the rotation they do not depend on i -- (k + rot(c)) mod 16
            // = a j mod 16 = a t mod 16 for odd SP (256 i = 0 mod 16), and for SP = 2, 4, 8 both c and
            // k mod 16 of sample t + 256 i are those of sample t
            constexpr unsigned INV = SP == 3 ? 11u : SP == 5 ? 13u : SP == 7 ? 7u : 1u;      // SP^-1 mod 16
            constexpr bool LOW_FIXED = ROT && (SP == 1 || SP == 2 || SP == 3 || SP == 4 || SP == 5 || SP == 7 || SP == 8);
            const unsigned low = (SP & 1) ? (INV * tt) & 15u : pf_cell<SP, ROT>(tt % SP, tt / SP) & 15u;
            if constexpr (DIAG & 32) {      // ablation: the staged samples never reach LDS (bounds what taking S0 off the ds_write path can give)
                v2f sink = s[0];
#pragma unroll
                for (int i = 1; i < R * SP; i++) sink += s[i];
                if (sink.x == 1.2345e38f) lds[tt] = sink;
            } else {
#pragma unroll
            for (int sg = 0; sg < R; sg++)
#pragma unroll
                for (int i = 0; i < SP; i++) {
                    const unsigned j = tt + 256u * i, c = j % SP, k = j / SP;
                    const unsigned cell = LOW_FIXED ? ((k & ~15u) | low) : pf_cell<SP, ROT>(c, k);
                    lds[(sg * SP + c) * PF_AREA + cell] = s[sg * SP + i];
                }
            }
        }
        lds_barrier();
        if constexpr (LATE == 3) {      // s[] was consumed by S0: the next pass's samples have S1..S3 to land
            if (nx < a.n_pass) {
                cur_wide = pass_is_wide(nxch, nx);
                if (cur_wide) {
                    load_pass_wide(raw, in_of(nxch), nx);
                } else if (!WIDE) {
                    cur_fast = pass_interior(nx);
                    if (cur_fast) load_pass(s, in_of(nxch), nx);
                }
            } else {
                cur_wide = cur_fast = false;
            }
        }
        unsigned drawn = 0u;
        bool drawing = TICKET && cur;
        if constexpr (LATE == 3) drawing = drawing && nx < a.n_pass;      // one failing draw per workgroup either way
        if (drawing && t == 0) drawn = draw_issue();       // finished and published before the barrier that ends S2
        // ---- S1: the transform (forward groups: staged samples; inverse groups: Y of the last pass)
        v2f v[16];
        asm volatile("" : "+v"(cell_in), "+v"(cell0));
#pragma unroll
        for (int r = 0; r < 16; r++) v[r] = lds[cell_in + 16u * r];
        dft16<-1>(v);
        v2f p2[4], q2[4];
        {
            unsigned tl = 16 * PF_AREA + l;
            asm volatile("" : "+v"(tl));
#pragma unroll
            for (int k = 1; k < 4; k++) {
                p2[k] = lds[tl + (k - 1) * 16];
                q2[k] = lds[tl + (k + 2) * 16];
            }
        }
#pragma unroll
        for (int k = 0; k < 16; k++) {
            v2f x = v[P16(k)];
            if ((k >> 2) && (k & 3)) x = cmul2(x, q2[k >> 2], p2[k & 3]);
            else if (k >> 2) x = cmul(x, q2[k >> 2]);
            else if (k & 3) x = cmul(x, p2[k & 3]);
            if constexpr (DIAG & 16) v[P16(k)] = x;        // ablation: no exchange at all (results are not a transform)
            else lds[cell0 + 17u * k] = x;
        }
        // the exchange stays inside this wave: program order + the compiler's waitcnt suffice
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if constexpr (!(DIAG & 16)) {
#pragma unroll
        for (int r = 0; r < 16; r++) v[r] = lds[cell0 + 16u * l + r];
        }
        dft16<-1>(v);                       // bin l + 16 k0 is in v[P16(k0)]
        // the draw is finished and published HERE: nothing of this pass is in flight yet (its samples were
        // consumed in S0), so the wait for the atomic waits for nothing else -- after S2 it would wait for
        // S2's output stores.  s_next is read after the barrier that ends S2.
        if (TICKET) asm volatile("" : "+v"(drawn));      // the wait sits here on every wave's path
        if (TICKET && t == 0) s_next = drawing ? draw_finish(drawn) : 0xFFFFFFFFu;
        // ---- S2
        if (is_fwd) {
#pragma unroll
            for (int k0 = 0; k0 < 16; k0++) lds[cell0 + 16u * k0] = v[P16(k0)];
        } else if (is_inv && prev >= 0) {
            // forward transform of Y, read backwards: y[n] = FFT(Y)[(256 - n) mod 256]
            // output index of this group's n = 0: uniform part + per-thread part (32-bit)
            const long long ku = ((prev * R) * (PAIR ? 2 : 1) * a.V - a.ovl) * UP;
            const int koff = (int)(seg * (PAIR ? 2u : 1u) * (unsigned)a.V * UP + comp);
            // outputs of this group exist for ovl <= n and ku + koff + UP n < n_out
            const long long remu = a.n_out - ku;
            const int rem = (remu > (1 << 30) ? (1 << 30) : (int)remu) - koff;
            const long long ko0 = ku + koff;
            const int lim = rem > 256 * UP ? 256 * UP : rem;           // UP n < lim; n < 256
            int nu = (256 - (int)l) * UP;                              // UP n for k0 = 0 (n = 256 - l; lane 0: n = 256 is bin 0 = sample 0, below)
            asm volatile("" : "+v"(nu));                               // keep the 16 offsets immediates off this, not 16 registers
            const int ovu = a.ovl * UP;
            // ovu <= x < lim as ONE unsigned compare per store: (x - ovu) < (lim - ovu)
            const unsigned span = lim > ovu ? (unsigned)(lim - ovu) : 0u;
            const unsigned xb = (unsigned)(nu - ovu);
            char *const out_c = out_of(pch);
            char *op = out_c + (ko0 + nu) * ESZ;
            if constexpr (PAIR) {
                // .x belongs to real segment 2*sigma, .y to segment 2*sigma + 1, V*UP outputs further on
                const int limB = rem - a.V * UP > 256 * UP ? 256 * UP : rem - a.V * UP;
                const unsigned spanB = limB > ovu ? (unsigned)(limB - ovu) : 0u;
                char *opB = op + (long long)a.V * UP * ESZ;
#pragma unroll
                for (int k0 = 0; k0 < 16; k0++) {
                    const unsigned xo = xb - (unsigned)(16 * UP * k0);
                    if (xo < span) __builtin_nontemporal_store(v[P16(k0)].x, reinterpret_cast<float *>(op - 16 * UP * ESZ * k0));
                    if (xo < spanB) __builtin_nontemporal_store(v[P16(k0)].y, reinterpret_cast<float *>(opB - 16 * UP * ESZ * k0));
                }
                if (l == 0 && a.ovl == 0) {
                    if (lim > 0) __builtin_nontemporal_store(v[P16(0)].x, reinterpret_cast<float *>(out_c + ko0 * ESZ));
                    if (limB > 0) __builtin_nontemporal_store(v[P16(0)].y, reinterpret_cast<float *>(out_c + (ko0 + (long long)a.V * UP) * ESZ));
                }
            } else {
                if constexpr (DIAG & 2) {       // ablation: no output stores
                    v2f acc = v[0];
#pragma unroll
                    for (int k0 = 1; k0 < 16; k0++) acc += v[k0];
                    if (acc.x == 1.2345e38f) *reinterpret_cast<v2f *>(op) = acc;
                } else if constexpr ((DIAG & 64) || UP * SP > 15) {       // the round-2 form, one exec-masked store and one branch per row: A/B (DIAG bit 6), and
                                                                          // the shapes whose 16+ spectra registers leave no room for the form below (5/4: 13 spilled VGPRs)
#pragma unroll
                for (int k0 = 0; k0 < 16; k0++)
                    if (xb - (unsigned)(16 * UP * k0) < span) __builtin_nontemporal_store(v[P16(k0)], reinterpret_cast<v2f *>(op - 16 * UP * 8 * k0));
                } else {
                    // Sixteen straight-line buffer stores through ONE descriptor over this pass's window of the output:
                    // its record count ends the window at n_out (the hardware drops what lies beyond), and a lane whose
                    // element is overlap (n < ovl) gets an offset no record count reaches -- a compare and a select per
                    // row instead of an exec-mask save, a branch and a restore (16 branches per pass in the round-2 ISA).
                    typedef int v2i __attribute__((ext_vector_type(2)));
                    char *const wb = out_c + ku * 8;                                     // uniform; offsets below are >= ovu*8
                    const long long recs = remu < 0 ? 0 : (remu > (1 << 26) ? (1 << 26) : remu);
                    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<uintptr_t>(wb));
                    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(reinterpret_cast<uintptr_t>(wb) >> 32));
                    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                        reinterpret_cast<void *>(((uintptr_t)hi << 32) | lo), 0, (int)(recs * 8), 0x00020000);
                    const unsigned span2 = 256u * UP > (unsigned)ovu ? 256u * UP - (unsigned)ovu : 0u;      // x < 256 UP: n = 256 is bin 0, below
                    // row k0 sits 16 UP elements below row k0 - 1: ONE per-lane offset (row 15's) selected against the
                    // unreachable one, the row's distance from it as a constant the instruction's offset field / a scalar holds
                    int vlow = (koff + nu) * 8 - 16 * UP * 8 * 15;
                    asm volatile("" : "+v"(vlow));
#pragma unroll
                    for (int k0 = 0; k0 < 16; k0++) {
                        const int sel = xb - (unsigned)(16 * UP * k0) < span2 ? vlow : (int)0x7FFF0000;
                        constexpr int STEP = 16 * UP * 8;
                        const int up = STEP * (15 - k0);                                  // 0 .. 15 STEP
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, v[P16(k0)]), rs, sel + (up & 2047), up & ~2047, 2 /* nt */);
                    }
                }
                if (l == 0 && a.ovl == 0 && lim > 0) __builtin_nontemporal_store(v[P16(0)], reinterpret_cast<v2f *>(out_c + ko0 * 8));
            }
        }
        lds_barrier();
        long long next = pass + gridDim.x;
        int nch = ch;
        if constexpr (TICKET) decode(__builtin_amdgcn_readfirstlane(s_next), nch, next);
        // v[] is dead and the next pass's samples are not requested yet: the point of least register pressure
        if (a.hist_out && cur && pass == 0) carry_history(ch);
        // request the next pass's samples now: v[] is dead, they land while S3 and the barrier run
        // (requesting them a whole pass ahead -- LATE == 3, variants A / B of the diagnostic library -- needs three
        // workgroups per CU or spills, and is no faster than this point at the same residency:
        // profiles/r03/resample_ahead.txt.  Latency is not what binds.)
        auto request_next = [&](int sg_lo, int sg_hi) {
            if (next < a.n_pass) {
                cur_wide = pass_is_wide(nch, next);
                if (cur_wide) {
                    if (sg_lo == 0) load_pass_wide(raw, in_of(nch), next);
                } else if (!WIDE) {
                    cur_fast = pass_interior(next);
                    if (cur_fast) load_pass(s, in_of(nch), next, sg_lo, sg_hi);
                }
            } else {
                cur_wide = cur_fast = false;
            }
        };
        if (LATE == 0) request_next(0, R);
        if (LATE == 2) request_next(0, (R + 1) / 2);
        // ---- S3: bin t of every segment of this pass
        if (cur && !(DIAG & 4)) {
#pragma unroll
            for (int sg = 0; sg < R; sg++) {
                v2f acc[UP];
#pragma unroll
                for (int c = 0; c < SP; c++) {
                    const v2f x = lds[(sg * SP + c) * PF_AREA + t];
#pragma unroll
                    for (int r = 0; r < UP; r++) acc[r] = c ? cmac(acc[r], x, G[r][c]) : cmul(x, G[r][c]);
                }
#pragma unroll
                for (int r = 0; r < UP; r++) lds[(F + sg * UP + r) * PF_AREA + t] = acc[r];
            }
        }
        lds_barrier();
        if (LATE == 1) request_next(0, R);
        if (LATE == 2) request_next((R + 1) / 2, R);
        prev = cur ? pass : -1;
        pch = ch;
        if constexpr (LATE == 3) {      // `next` / `nch` decoded above are the pass AFTER nx
            pass = nx;
            ch = nxch;
            nx = TICKET ? next : nx + gridDim.x;
            nxch = nch;
        } else {
            pass = next;
            ch = nch;
        }
    }
}

template <int SP, int UP, int R, bool IN_U8, bool PAIR, int DIAG = 0, bool TICKET = false, int LATE = 0, int WPS = 4>
int launch_one(const PolyFftArgs &a0, int n_channels, hipStream_t s)
{
    PolyFftArgs a = a0;
    // workgroups of this instantiation a CU holds (a property of the code object, the same on every
    // gfx950 device) x the compute units of the launch's device (cached per device ordinal, common.h)
    static std::atomic<int> per_cu_cache{0};
    int per_cu = per_cu_cache.load(std::memory_order_relaxed);
    if (!per_cu) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, poly_fft256_kernel<SP, UP, R, IN_U8, PAIR, DIAG, TICKET, LATE, WPS>, 256, 0) != hipSuccess || per_cu < 1)
            return hip_fail(hipGetLastError(), "poly_fft occupancy");
        per_cu_cache.store(per_cu, std::memory_order_relaxed);
    }
    const long long resident = (long long)device_cu_count() * per_cu;
    // persistent workgroups, shared over the channels
    // fixed-stride walk: two workgroups per resident slot (1, 2, 3, 4, 8 measured within noise of each
    // other); work counters: exactly the resident count -- any further workgroup would start when
    // the counters are already exhausted and only pay its prologue (grid x1 0.701, x2 0.714, x4 0.723 ms)
    long long factor = TICKET ? 1 : 2;
#ifdef SFE_DIAG
    if (const char *e = getenv("SFE_RS_WG_FACTOR")) factor = atoi(e) > 0 ? atoi(e) : factor;
#endif
    long long cap = TICKET ? factor * resident : (factor * resident + n_channels - 1) / n_channels;
    if (cap < 1) cap = 1;
    const long long total = a.n_pass * (TICKET ? n_channels : 1);
    dim3 grid((unsigned)(total < cap ? total : cap), TICKET ? 1u : (unsigned)n_channels);
    if (TICKET) {
        if (!a.ticket || total + grid.x >= 0xFFFFFFFFLL) return SFE_ESTATE;
        a.total = (unsigned)total;
        a.tgroups = POLY_TICKET_GROUPS < grid.x ? POLY_TICKET_GROUPS : grid.x;
        a.tqs = 3;            // `t` against `t^3!1` (scripts/ab_rs.py): -0.6 % time, the overlap re-read leaves HBM
        a.halo_keep = 1;
#ifdef SFE_DIAG
        if (const char *e = getenv("SFE_RS_TQS")) a.tqs = atoi(e) >= 0 && atoi(e) <= 8 ? (unsigned)atoi(e) : 0u;
        if (const char *e = getenv("SFE_RS_HALO_KEEP")) a.halo_keep = atoi(e) ? 1u : 0u;
#endif
    }
    hipLaunchKernelGGL((poly_fft256_kernel<SP, UP, R, IN_U8, PAIR, DIAG, TICKET, LATE, WPS>), grid, dim3(256), 0, s, a);
    hipError_t err = hipGetLastError();
    return err == hipSuccess ? SFE_OK : hip_fail(err, "poly_fft launch");
}

}  // namespace

// instantiated shapes: (SP, UP, R) with R*(SP + UP) <= 16 lane groups
#define SFE_PF_SHAPES(X) \
    X(5, 3, 2) X(3, 2, 3) X(5, 2, 2) X(4, 3, 2) X(2, 1, 5) X(3, 1, 4) X(4, 1, 3) X(5, 1, 2) \
    X(6, 1, 2) X(7, 1, 2) X(8, 1, 1) X(7, 2, 1) X(5, 4, 1)

int poly_fft_segments(int SP, int UP)
{
#define SFE_PF(sp, up, r) if (SP == sp && UP == up) return r;
    SFE_PF_SHAPES(SFE_PF)
#undef SFE_PF
    return 0;
}

int launch_poly_fft(const PolyFftPlan &plan, const PolyFftArgs &a0, int data_complex, int in_u8, int n_channels, hipStream_t s)
{
    PolyFftArgs a = a0;
#ifdef SFE_DIAG
    // ablations of the headline shape (scripts/ablate.py): SFE_RS_DIAG bit 0 no loads, 1 no stores, 2 no spectrum stage
    if (const char *e = getenv("SFE_RS_DIAG")) {
        const int d = atoi(e) & 7;
        if (d && plan.SP == 5 && plan.UP == 3 && data_complex && !in_u8) {
            const long long m_count = (a.n_out + plan.UP - 1) / plan.UP, n_seg = (m_count + a.V - 1) / a.V;
            a.n_pass = (n_seg + 1) / 2;
            switch (d) {
            case 1: return launch_one<5, 3, 2, false, false, 1>(a, n_channels, s);
            case 2: return launch_one<5, 3, 2, false, false, 2>(a, n_channels, s);
            case 3: return launch_one<5, 3, 2, false, false, 3>(a, n_channels, s);
            case 4: return launch_one<5, 3, 2, false, false, 4>(a, n_channels, s);
            case 7: return launch_one<5, 3, 2, false, false, 7>(a, n_channels, s);
            default: break;
            }
        }
    }
#endif
    const int R = poly_fft_segments(plan.SP, plan.UP);
    if (!R || a.n_out <= 0) return SFE_ESTATE;
    const long long m_count = (a.n_out + plan.UP - 1) / plan.UP;
    const long long n_seg = (m_count + a.V - 1) / a.V;
    const long long n_xf = data_complex ? n_seg : (n_seg + 1) / 2;      // real data: two segments per transform
    a.n_pass = (n_xf + R - 1) / R;
#ifdef SFE_DIAG
    // SFE_RS_VARIANT: s = fixed-stride walk (round 1), t = tickets (the product's single-channel kernel), l = tickets + late request
    if (const char *e = getenv("SFE_RS_VARIANT")) {
        if (plan.SP == 5 && plan.UP == 3 && data_complex && !in_u8 && n_channels == 1) {
            if (e[0] == 's') return launch_one<5, 3, 2, false, false, 0, false, 0>(a, n_channels, s);
            if (e[0] == 't') return launch_one<5, 3, 2, false, false, 0, true, 0>(a, n_channels, s);
            if (e[0] == 'l') return launch_one<5, 3, 2, false, false, 0, true, 1>(a, n_channels, s);
            if (e[0] == 'L') return launch_one<5, 3, 2, false, false, 0, false, 1>(a, n_channels, s);
            if (e[0] == 'x') return launch_one<5, 3, 2, false, false, 8, true, 0>(a, n_channels, s);         // t with the round-1 XOR scatter layout
            if (e[0] == 'h') return launch_one<5, 3, 2, false, false, 0, true, 2>(a, n_channels, s);         // half before S3, half after
            if (e[0] == 'e') return launch_one<5, 3, 2, false, false, 16, true, 0>(a, n_channels, s);        // t without the in-wave exchange (upper bound for doing it off the LDS)
            if (e[0] == 'w') return launch_one<5, 3, 2, false, false, 0, true, 0, 3>(a, n_channels, s);     // 3 workgroups per CU, 168 VGPRs: no spill
            if (e[0] == 'p') return launch_one<5, 3, 2, false, false, 128, true, 0>(a, n_channels, s);       // t, waves 2-3 (inverse transforms + stores) at priority 1
            if (e[0] == 'q') return launch_one<5, 3, 2, false, false, 256, true, 0>(a, n_channels, s);       // t, waves 0-1 (forward only) at priority 1
            if (e[0] == 'b') return launch_one<5, 3, 2, false, false, 64, true, 0>(a, n_channels, s);        // t with the round-2 exec-masked stores
            if (e[0] == 'z') return launch_one<5, 3, 2, false, false, 32, true, 0>(a, n_channels, s);        // t without the S0 scatter writes (bound for LDS-DMA staging)
            if (e[0] == 'Z') return launch_one<5, 3, 2, false, false, 48, true, 0>(a, n_channels, s);        // neither the scatter nor the exchange
            if (e[0] == 'y') return launch_one<5, 3, 2, false, false, 32, true, 0, 3>(a, n_channels, s);     // z at 3 workgroups per CU
            if (e[0] == 'A') return launch_one<5, 3, 2, false, false, 0, true, 3, 3>(a, n_channels, s);      // samples requested a whole pass ahead, 3 workgroups per CU (168 VGPRs)
            if (e[0] == 'B') return launch_one<5, 3, 2, false, false, 0, true, 3, 4>(a, n_channels, s);      // ... at 4 per CU (spills)
        }
    }
#endif
    // work counters whenever the passes of all channels fit the 32-bit ticket (else the fixed-stride walk per channel)
    const bool tk = a.ticket != nullptr && a.n_pass * (long long)n_channels < 0xFFF00000LL;
#define SFE_PF1(sp, up, r, U8, PR) (tk ? launch_one<sp, up, r, U8, PR, 0, true>(a, n_channels, s) : launch_one<sp, up, r, U8, PR>(a, n_channels, s))
#define SFE_PF(sp, up, r)                                                                         \
    if (plan.SP == sp && plan.UP == up)                                                           \
        return data_complex ? (in_u8 ? SFE_PF1(sp, up, r, true, false) : SFE_PF1(sp, up, r, false, false)) \
                            : (in_u8 ? SFE_PF1(sp, up, r, true, true) : SFE_PF1(sp, up, r, false, true));
    SFE_PF_SHAPES(SFE_PF)
#undef SFE_PF
#undef SFE_PF1
    return SFE_ESTATE;
}

}  // namespace sfe
