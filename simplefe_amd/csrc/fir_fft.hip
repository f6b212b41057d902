// fir_fft.hip -- the headline kernel: streaming FIR by overlap-save with a 4096-point complex
// FFT held entirely in one workgroup's registers + LDS (gfx950).
//
// What it computes is the net effect of blkconv::process() over a stream
// (libdsp/blkconv.cxx:77-110): y[n] = sum_k h[k] x[n-k] with carried state.  The reference
// does it by overlap-ADD on real blocks with FFTW; the block scheme does not show in the
// result, so the GPU uses overlap-SAVE (no read-modify-write of the output, no inter-block
// dependency) and transforms I + jQ as ONE complex sequence: complex data and complex taps
// cost the same as real ones.
//
// Shape: N = 4096 = 16 x 16 x 16.  256 threads, thread t owns 16 complex values in VGPRs.
//   index n = n0 + 16 n1 + 256 n2,  bin k = k2 + 16 k1 + 256 k0
//   F1  t = n0+16n1 : DFT16 over n2, times W_4096^(t k2)        -> LDS [k2][n1][n0]
//   F2  t = n0+16k2 : DFT16 over n1, times W_256^(n0 k1)        -> LDS [k2][k1][n0]
//   F3  t = k1+16k2 : DFT16 over n0, times H[k]/N, IDFT16 over k0 -> LDS (same cells)
//   I2  t = n0+16k2 : times conj W_256, IDFT16 over k1          -> LDS [k2][n1][n0]
//   I3  t = n0+16n1 : times conj W_4096, IDFT16 over k2 -> y[t + 256 n2]
// Global traffic is perfectly coalesced in both directions (lane == consecutive sample,
// register == row of 256); the first hl/256 rows of the result are the overlap-save discard.
// LDS rows are padded (272 / 17 complex) so every ds_read_b64/ds_write_b64 is conflict-free.
// Twiddle bases AND this thread's 16 bins of the taps' spectrum H/N (hreg, 32 VGPRs) live in
// registers for the life of the (persistent) workgroup, which walks transforms blockIdx.x,
// +gridDim.x, ...  Budget: <= 128 VGPRs and 34 KiB LDS -> 4 workgroups per CU.
//
// Build flavours: the product library (libsfe_dsp.so) contains only the kernels the C ABI can
// reach and reads no environment variable.  -DSFE_DIAG (libsfe_dsp_diag.so, simplefe_amd/build.py
// build_lib(diag=True); used by scripts/ only) adds the A/B variants, the bare access-pattern
// kernels and the load/store suppression switches behind SFE_FIR_VARIANT / SFE_FIR_DIAG.
#include <stdint.h>
#include <stdlib.h>

#include "common.h"
#include "fft16.h"

namespace sfe {
namespace {

// `p` is wave-uniform (an SGPR pair), `lane` the per-lane element offset: keeps the address
// math scalar so each row costs one global_load with an SGPR base and a shared VGPR offset.
template <bool IN_C, bool IN_U8 = false>
__device__ __forceinline__ v2f load_sample(const void *p, unsigned lane)
{
    if constexpr (IN_U8) {   // wire format: (b - 128) / 127 on load
        if constexpr (IN_C) {
            const unsigned w = __builtin_nontemporal_load(reinterpret_cast<const unsigned short *>(p) + lane);
            return (v2f){u8_to_f32(w & 0xFFu), u8_to_f32(w >> 8)};
        } else {
            return (v2f){u8_to_f32(__builtin_nontemporal_load(reinterpret_cast<const unsigned char *>(p) + lane)), 0.0f};
        }
    }
    // nontemporal: the sample stream is read once (measured -4% on the access pattern alone)
    if constexpr (IN_C) return __builtin_nontemporal_load(reinterpret_cast<const v2f *>(p) + lane);
    else return (v2f){__builtin_nontemporal_load(reinterpret_cast<const float *>(p) + lane), 0.0f};
}

// value held by lane K of this lane's quad / by the odd lane of this lane's pair: DPP quad_perm
// moves (plain VALU), not ds_bpermute -- the packing runs 15 rows x 2 segments per transform
template <int K>
__device__ __forceinline__ unsigned quad_lane(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, K | (K << 2) | (K << 4) | (K << 6), 0xF, 0xF, true);
}
__device__ __forceinline__ unsigned quad_odd(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 1 | (1 << 2) | (3 << 4) | (3 << 6), 0xF, 0xF, true);
}

// four 10-bit codes -> the 5-byte group of the transmit wire format (sink_f_impl.cc:133-140): the
// high-bits byte and the first three low bytes go out as ONE (unaligned) dword store, the last low
// byte as a byte store -- two stores per group instead of five
__device__ __forceinline__ void store_group(unsigned char *d, unsigned u0, unsigned u1, unsigned u2, unsigned u3)
{
    const unsigned w = ((u0 >> 8) | ((u1 >> 8) << 2) | ((u2 >> 8) << 4) | ((u3 >> 8) << 6)) | ((u0 & 0xFFu) << 8) |
                       ((u1 & 0xFFu) << 16) | ((u2 & 0xFFu) << 24);
    __builtin_memcpy(d, &w, 4);
    d[4] = (unsigned char)u3;
}

#ifdef SFE_DIAG
// In-kernel clock of a launch (MI355X_MICROARCH.md, DVFS give-back item 6): workgroup 0 stamps s_memtime (shader clocks) and
// s_memrealtime (100 MHz) when it starts and when it leaves; the ratio of the two differences x 100 MHz is the clock the chip
// held over the launch.  Diagnostic library only (sfe_dsp_diag_fir_clock, scripts/ab_fir.py); the values go nowhere else.
__device__ unsigned long long g_fir_clk[4];
#define SFE_FIR_STAMP(i)                                                                   \
    do {                                                                                   \
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {                      \
            g_fir_clk[i] = __builtin_amdgcn_s_memtime();                                   \
            g_fir_clk[(i) + 1] = __builtin_amdgcn_s_memrealtime();                         \
        }                                                                                  \
    } while (0)
#else
#define SFE_FIR_STAMP(i) do { } while (0)
#endif

// Every instantiation: 256 threads, four workgroups per CU (<= 128 VGPRs), this thread's 16 bins of H/N in registers (32 VGPRs:
// re-reading 32 KiB from L2 per transform shares the CU's vector-memory path with the samples and cost 10 %), transforms drawn from
// device counters.  (Rounds 1-3 carried five more template switches -- waves per SIMD, a register prefetch, an XOR-swizzled
// exchange buffer, the spectrum streamed from L2, a fixed-stride walk instead of counters -- each measured and lost; round 5
// removed them and their diagnostic instantiations: profiles/r02 and r03 hold the tables, the history the code.)
// PAIR (real data, real taps only): the real and imaginary parts of one transform carry two
// CONSECUTIVE real segments of the stream (z = x_A + j x_B; real taps keep them apart), so a real
// stream costs what a complex one does per sample instead of twice as much.
// OUT_TX10 (real streams with PAIR, or complex in and out): the output is written in the device's
// transmit wire format, 10-bit offset binary, 4 floats (4 real or 2 complex samples) in 5 bytes -- ((short)(x*511)+512)&0x3FF, packed as
// gr-simplefe/lib/sink_f_impl.cc:117-143 / examples/bpsk/bpsk.cxx:76-101 do on the host.
// DMA (complex float32 input, 16-byte aligned channels): an interior transform's 32 KiB of input are
// requested by LDS-DMA (global_load_lds_dwordx4: 16-byte lanes, no VGPR destination) straight into
// the exchange buffer's padded row layout, and they are requested EARLY -- as soon as the previous
// transform's last LDS reads are done, i.e. before its final DFT16 and its 15 rows of stores -- so
// part of the HBM latency runs under that work without costing a register.  F1 then reads its column from LDS.
// Edge transforms (history in front, ragged end) keep the guarded register loads.
template <bool IN_C, bool OUT_C, bool IN_U8 = false, bool PAIR = false, bool OUT_TX10 = false, bool DMA = false, int DIAG = 0,
          bool ACC = false, bool WP = false, bool HCH = false>
__global__ __launch_bounds__(256, 4) void fir_fft4096_kernel(FirFftArgs a)
{
    // HCH: per-channel taps -- the spectrum registers are reloaded when the workgroup's next transform
    // belongs to another channel (channel-major tickets: every few transforms at 64 x 2^24, not every one)
    // WP (with DMA): wave-private [n2|k2][column] layout.  Thread t touches column t of the first /
    // last exchange layout in F1 (read + write), I3 (read) and nothing else does between the I2->I3
    // barrier and the F1->F2 barrier; if every wave's 64 columns sit in a region of their own AND the
    // wave's LDS-DMA pieces cover exactly that region, the path "I3 reads -> request the next
    // transform -> (this transform's last DFT16 and stores) -> wait -> F1" involves no other wave:
    // two of the eight workgroup barriers go, and the four waves issue their memory bursts when
    // each is ready instead of together.  Cell of (row, column c): region c/64 (1152 cells), row
    // r at 144 (r mod 8) + 64 (r / 8) -- one DMA piece = rows (i, i+8) of the wave's 64 columns
    // (lanes 0-31 / 32-63), 128 contiguous cells, pieces 144 apart: 144 = 16 mod 32 keeps the F2
    // reads / I2 writes (two adjacent k2 per 32 lanes) conflict-free exactly as 272 does.
    static_assert(!WP || DMA, "the wave-private layout exists for the LDS-DMA path");
    // ACC / a.shift (filters longer than one transform can overlap: partitioned convolution, api_fir.hip
    // fir_run): this launch applies ONE partition h_p = h[p*hl .. (p+1)*hl) of the taps to the stream
    // delayed by a.shift = p*hl samples and, with ACC, adds its result to what the earlier partitions
    // left in the output.  a.hist_len >= a.hl + a.shift samples of history precede the input.
    static_assert(!ACC || (!OUT_TX10 && !DMA), "accumulating launches write float32 through the plain store path");
    // Tickets: the persistent workgroups do not walk a fixed stride (blockIdx.x, +gridDim.x, ...) but
    // draw the next transform from one device-wide counter, channel-major.  Whatever the speed of
    // individual workgroups, the transforms in flight are then always the ~1000 NEXT ones of the
    // stream: one compact, advancing window of reads and one of writes, as a one-workgroup-per-
    // transform grid gives (HBM moves that pattern 10 % faster than 1000 drifting fixed-stride
    // walks: the kernel's bare access pattern 0.716 against 0.792 ms, profiles/r02) -- without
    // re-loading the 44 twiddle / spectrum registers per transform.  The draw for transform i+1 is
    // issued early in transform i (one lane; its latency runs under two exchange stages).
    // DIAG (instantiated under -DSFE_DIAG only, scripts/ablate.py): bit 0 = input loads replaced by
    // constants, bit 1 = output stores folded into one never-taken store -- compile-time, so the
    // product kernel's instruction stream and register allocation are untouched.
    static_assert(!DMA || (IN_C && !IN_U8 && !PAIR), "LDS-DMA input: complex float32, padded layout");
    // DIAG bit 3 (diagnostic library only, variants y / u): a row is stored as soon as the butterfly of the last DFT16
    // that completes it is done, instead of behind the whole DFT16 (DESIGN.md 9 lead (ii); it spills: see the store section)
    constexpr bool INTERLEAVE = (DIAG & 8) != 0;
    // DIAG bits 4 and 5 (diagnostic library only, round 5's energy pass): ablations that leave WRONG results and a
    // true timing / power picture.  Bit 4 (NOMID): the two MIDDLE exchanges (F2 -> F3 and I1 -> I2) neither write nor read the
    // LDS and their three barriers go -- each thread carries on with its own sixteen values; the arithmetic is untouched.  What a
    // 64 x 64 decomposition with one exchange per direction could save at most (it would pay cross-lane moves for it).
    // Bit 5 (NORECON): every twiddle is ONE complex multiply (by q alone) instead of q[a] p[b]: what a resident table of all 15
    // factors per stage could save at most (it would pay 36 more VGPRs for it).
    constexpr bool NOMID = (DIAG & 16) != 0, NORECON = (DIAG & 32) != 0;
    auto tw2f = [](v2f x, v2f q, v2f p) -> v2f { return NORECON ? cmul(x, q) : cmul2(x, q, p); };
    auto tw2i = [](v2f x, v2f q, v2f p) -> v2f { return NORECON ? cmul_conj(x, q) : cmul2_conj(x, q, p); };
    __shared__ v2f lds[WP ? 4 * WP_REGION : FFT_ROWS * LDS_K2_STRIDE];
    __shared__ unsigned s_next;       // the next transform drawn by lane 0
    const unsigned t = threadIdx.x;   // unsigned: lets loads/stores use SGPR base + 32-bit VGPR offset
    SFE_FIR_STAMP(0);
    int ch = 0;
    const unsigned lo = t & 15, hi = t >> 4;
    // LDS cell of element (k2, a, n0), a = n1 or k1 (padded rows: every access conflict-free):
    //   [k2][n1][n0] -> 272 k2 + 16 n1 + n0,  [k2][k1][n0] -> 272 k2 + 17 k1 + n0
    const unsigned base_a = WP ? (t >> 6) * WP_REGION + (t & 63u) : t;   // + row_a(k2)
    const unsigned base_b = hi * LDS_K2_STRIDE + lo;
    const unsigned base_c = hi * LDS_K2_STRIDE + lo * LDS_K1_STRIDE;
    const unsigned base_w = (hi & 7u) * WP_PITCH + (hi >> 3) * 64u + lo;      // WP: row hi of column lo (+ 16 n1)
    // offset of row r in the first / last layout, relative to the thread's column
    auto row_a = [](int r) -> unsigned { return WP ? (unsigned)((r & 7) * WP_PITCH + (r >> 3) * 64) : (unsigned)(r * LDS_K2_STRIDE); };
    auto cell_b1 = [&](int a1) -> unsigned {
        if (WP) return base_w + (unsigned)((a1 >> 2) * WP_REGION + 16 * (a1 & 3));     // column lo + 16 a1 of row hi
        return base_b + 16u * a1;
    };
    auto cell_b2 = [&](int a1) -> unsigned { return base_b + (unsigned)LDS_K1_STRIDE * a1; };
    auto cell_c = [&](int n0) -> unsigned { return base_c + n0; };

    constexpr int ISZ = IN_U8 ? (IN_C ? 2 : 1) : (IN_C ? 8 : 4);   // bytes per input sample
    const char *in_c, *hist_c;
    char *out_c;
    auto in_of = [&](int c) -> const char * { return static_cast<const char *>(a.in) + (size_t)c * a.in_stride * ISZ; };
    v2f hreg[16];
    auto load_spectrum = [&](int c) {
        if constexpr (HCH) {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                hreg[k] = (a.hs + (size_t)c * a.hs_stride + k * 256)[threadIdx.x];
                asm volatile("" : "+v"(hreg[k]));
            }
        }
    };
    auto set_channel = [&](int c) {
        in_c = in_of(c);
        // 10-bit output: the channel's floats (2 per complex sample) in whole groups of 4 -> 5 bytes
        out_c = static_cast<char *>(a.out) + (OUT_TX10 ? (size_t)c * (a.out_stride * (OUT_C ? 2 : 1) / 4) * 5 : (size_t)c * a.out_stride * (OUT_C ? 8 : 4));
        hist_c = static_cast<const char *>(a.hist) + (size_t)c * a.hist_len * (IN_C ? 8 : 4);
    };
    load_spectrum(ch);
    set_channel(ch);

    // Per-thread twiddle bases, resident for the whole launch.  A twiddle with exponent
    // e*(4a+b) is applied as q[a]*p[b], q[a] = W^(4 e a), p[b] = W^(e b): 12 complex registers
    // for the two twiddle stages instead of 30+30.
    v2f p1[4], q1[4], p2[4], q2[4];
#pragma unroll
    for (int k = 1; k < 4; k++) {
        p1[k] = a.tw1[k * 256 + t];          // W_4096^(t k)
        q1[k] = a.tw1[(k + 3) * 256 + t];    // W_4096^(4 t k)
        p2[k] = a.tw2[k * 16 + lo];          // W_256^(lo k)
        q2[k] = a.tw2[(k + 3) * 16 + lo];    // W_256^(4 lo k)
    }

    const int row0 = a.hl >> 8;   // rows discarded by overlap-save
    // state carry-over fused in (the workgroup that takes a channel's transform 0): next call's
    // history = the last hl input samples, converted to float32 if the stream is u8
    auto carry_history = [&]() {
        char *ho = static_cast<char *>(a.hist_out) + (size_t)ch * a.hist_len * (IN_C ? 8 : 4);
        for (int r = 0; r < (a.hist_len >> 8); r++) {
            const v2f s = load_sample<IN_C, IN_U8>(in_c + (a.n - a.hist_len + 256 * r) * ISZ, t);
            if constexpr (IN_C) reinterpret_cast<v2f *>(ho)[256 * r + t] = s;
            else reinterpret_cast<float *>(ho)[256 * r + t] = s.x;
        }
    };
    // one draw from the launch's counter; the launch's LAST draw (every workgroup draws exactly one
    // ticket beyond the end) puts the counter back to zero for the next launch on this handle
    // (one address serves an atomic every ~13 ns -- 70 000 draws from ONE counter take as long as the
    // whole launch -- so the workgroups are dealt into a.tgroups groups, group g drawing the
    // transforms g, g + tgroups, ... from its own counter, 128 bytes apart)
    const unsigned tg = a.tgroups, grp = blockIdx.x % tg;
    unsigned *const my_ticket = a.ticket + 32u * grp;
    // group g owns the runs g, g + tg, ... of Q = 2^tqs consecutive transforms; ticket c of the group is
    // transform ((c / Q) tg + g) Q + c % Q.  The group's share: Q per complete row of tg runs + its part of the last row.
    const unsigned Q = 1u << a.tqs, row = tg << a.tqs, rem = a.total % row;
    // (per-channel spectra, FirFftArgs::ch_groups: the group owns whole channels, grp, grp + tg, ...)
    const bool by_channel = HCH && a.ch_groups;
    const unsigned mine = by_channel ? (a.total / (unsigned)a.nblk / tg) * (unsigned)a.nblk
                                     : (a.total / row << a.tqs) + (rem > grp * Q ? (rem - grp * Q < Q ? rem - grp * Q : Q) : 0u);
    const unsigned last_draw = mine + (gridDim.x - grp + tg - 1u) / tg - 1u;
    // a launch with no more transforms than workgroups (the per-block host calls) deals them by
    // blockIdx and leaves the counters alone: two atomic round trips less on a 10 us kernel
    const bool few = a.total <= gridDim.x;
    // A draw is ISSUED (the atomic goes out, nothing looks at what it returns) and FINISHED two stages later
    // (reset on the last value, ticket -> transform): looked at where it is issued, the returning atomic is
    // waited for with s_waitcnt vmcnt(0) -- behind the previous transform's 15 stores, which nothing else ever
    // waits for -- by the wave every barrier of the transform then waits for.  (The file is compiled with the
    // atomic optimizer off: its wave-aggregated form reads the result back at once.)
    auto draw_issue = [&]() -> unsigned {
        if (few) return 0u;
        return __hip_atomic_fetch_add(my_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto draw_finish = [&](unsigned c) -> unsigned {
        if (few) return 0xFFFFFFFFu;
        if (c == last_draw) __hip_atomic_store(my_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (by_channel) {
            if (c >= mine) return 0xFFFFFFFFu;
            const unsigned i = c / (unsigned)a.nblk;                     // the group's i-th channel
            return (grp + tg * i) * (unsigned)a.nblk + (c - i * (unsigned)a.nblk);
        }
        const unsigned long long k = (((unsigned long long)(c >> a.tqs) * tg + grp) << a.tqs) + (c & (Q - 1u));
        return k < a.total ? (unsigned)k : 0xFFFFFFFFu;
    };
    auto draw = [&]() -> unsigned { return draw_finish(draw_issue()); };
    const unsigned nblk32 = (unsigned)a.nblk;
    if (!HCH) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            hreg[k] = (a.hs + k * 256)[t];
            asm volatile("" : "+v"(hreg[k]));   // pin: otherwise the load is sunk back into the loop
        }
    }

    // Loads one transform's 16 rows (thread t: samples base + t + 256 r) into registers.
    auto load_rows = [&](v2f (&x)[16], long long blk) {
        constexpr int ESZ = IN_C ? 8 : 4;     // history is always float32
        if constexpr (PAIR) {
            // segment A = transform 2*blk of the real stream, segment B = transform 2*blk + 1
            const long long baseA = 2 * blk * a.advance - a.hl - a.shift, baseB = baseA + a.advance;
            if constexpr (IN_U8) {
                // real u8 stream: each segment's 4 KiB as one 16-byte lane per thread, parked raw in
                // LDS and picked up bytewise (the complex form is below)
                if (baseA >= 0 && baseB + FFT_N <= a.n && (reinterpret_cast<uintptr_t>(in_c) & 15u) == 0) {
                    typedef unsigned v4u __attribute__((ext_vector_type(4)));
                    const v4u rA = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(in_c + baseA) + t);
                    const v4u rB = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(in_c + baseB) + t);
                    v4u *raw = reinterpret_cast<v4u *>(lds);
                    raw[t] = rA;
                    raw[256u + t] = rB;
                    lds_barrier();
                    const unsigned char *rs = reinterpret_cast<const unsigned char *>(lds);
#pragma unroll
                    for (int r = 0; r < 16; r++) x[r] = (v2f){u8_to_f32(rs[256 * r + t]), u8_to_f32(rs[4096 + 256 * r + t])};
                    lds_barrier();      // F1 rewrites these cells
                    return;
                }
            }
            if (baseA >= 0 && baseB + FFT_N <= a.n) {
#pragma unroll
                for (int r = 0; r < 16; r++)
                    x[r] = (v2f){load_sample<false, IN_U8>(in_c + (baseA + 256 * r) * ISZ, t).x,
                                 load_sample<false, IN_U8>(in_c + (baseB + 256 * r) * ISZ, t).x};
            } else {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    float p[2];
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        const long long row = (q ? baseB : baseA) + 256 * r;   // uniform
                        if (row + (long long)t < 0) p[q] = load_sample<false>(hist_c + (a.hist_len + row) * ESZ, t).x;
                        else if (row + (long long)t < a.n) p[q] = load_sample<false, IN_U8>(in_c + row * ISZ, t).x;
                        else p[q] = 0.0f;
                    }
                    x[r] = (v2f){p[0], p[1]};
                }
            }
            return;
        }
        const long long base = blk * a.advance - a.hl - a.shift;   // stream index of transform element 0
        if constexpr (DIAG & 1) {       // ablation: no input loads (values that keep the arithmetic alive)
            unsigned u = t + ((unsigned)blk << 12);
            asm volatile("" : "+v"(u));        // opaque per iteration: nothing of this is hoisted out of the loop
#pragma unroll
            for (int r = 0; r < 16; r++) x[r] = (v2f){__builtin_bit_cast(float, 0x3f000000u | ((u + 256u * r) & 0x7fffffu)), 0.25f};
            return;
        }
        if constexpr (IN_U8 && IN_C) {
            // u8 wire format, interior transform, 16-byte-aligned stream: the transform's 8 KiB are
            // requested as 16-byte lanes (two per thread instead of sixteen 2-byte ones), parked raw
            // in the first 8 KiB of the exchange buffer and picked up as this thread's 16 samples.
            // (base*2 is a multiple of 512: advance and hl are multiples of 256.)
            if (base >= 0 && base + FFT_N <= a.n && (reinterpret_cast<uintptr_t>(in_c) & 15u) == 0) {
                typedef unsigned v4u __attribute__((ext_vector_type(4)));
                const v4u *src = reinterpret_cast<const v4u *>(in_c + base * 2);
                const v4u r0 = __builtin_nontemporal_load(src + t), r1 = __builtin_nontemporal_load(src + 256u + t);
                v4u *raw = reinterpret_cast<v4u *>(lds);
                raw[t] = r0;
                raw[256u + t] = r1;
                lds_barrier();
                const unsigned short *rs = reinterpret_cast<const unsigned short *>(lds);
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const unsigned w = rs[256 * r + t];
                    x[r] = (v2f){u8_to_f32(w & 0xFFu), u8_to_f32(w >> 8)};
                }
                lds_barrier();      // F1 rewrites these cells
                return;
            }
        }
        if (base >= 0 && base + FFT_N <= a.n) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                // rows 0 and 15 are the ones a neighbouring transform reads too: without the nontemporal hint they
                // may stay in the XCD's L2 for the neighbour drawn microseconds later (launcher: a.halo_keep)
                if (IN_C && !IN_U8 && (r == 0 || r == 15) && ((a.halo_keep >> r) & 1u))
                    x[r] = reinterpret_cast<const v2f *>(in_c + (base + 256 * r) * ISZ)[t];
                else
                    x[r] = load_sample<IN_C, IN_U8>(in_c + (base + 256 * r) * ISZ, t);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const long long row = base + 256 * r;          // uniform
                if (row + (long long)t < 0) x[r] = load_sample<IN_C>(hist_c + (a.hist_len + row) * ESZ, t);
                else if (row + (long long)t < a.n) x[r] = load_sample<IN_C, IN_U8>(in_c + row * ISZ, t);
                else x[r] = (v2f){0.0f, 0.0f};
            }
        }
    };

    // LDS-DMA of one interior transform: wave w requests rows 4w..4w+3 as eight 1-KiB pieces (half a
    // row of 128 samples each: 64 lanes x 16 bytes), piece (row, half) landing at the padded cell
    // 272*row + 128*half -- the [n2][t] cells F1 reads.  M0 = the piece's LDS byte address.
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) v2f *)lds;
    const unsigned wv = __builtin_amdgcn_readfirstlane(t >> 6), lane16 = (t & 63u) * 16u;
    const unsigned lane_wp = ((t >> 5) & 1u) * (8u * 256u * 8u) + (t & 31u) * 16u;    // WP: row +8 for the upper half-wave
    auto interior = [&](long long blk) -> bool {
        if constexpr (DIAG & 1) return false;
        const long long base = blk * a.advance - a.hl - a.shift;
        return base >= 0 && base + FFT_N <= a.n;
    };
    auto dma_rows = [&](const char *chan, long long blk) {
        const char *g = chan + (blk * a.advance - a.hl - a.shift) * 8;      // uniform
#pragma unroll
        for (int p = 0; p < 8; p++) {
            // the piece's place in the transform goes into the SCALAR base, the lane's into the one vector
            // offset all eight pieces share (eight per-piece vector offsets cost the wave-private form 4 spilled
            // VGPRs, each reloaded behind an s_waitcnt vmcnt(0) between two requests)
            unsigned dst, off;
            const char *gp;
            if constexpr (WP) {
                // piece p of wave wv: rows p (lanes 0-31) and p + 8 (lanes 32-63) of columns 64 wv .. 64 wv + 63
                dst = lds_base + (wv * WP_REGION + (unsigned)p * WP_PITCH) * 8u;
                gp = g + ((unsigned)p * 256u + 64u * wv) * 8u;
                off = lane_wp;
            } else {
                const unsigned row = 4u * wv + (p >> 1), half = p & 1;
                dst = lds_base + (row * LDS_K2_STRIDE + half * 128u) * 8u;
                gp = g + (row * 256u + half * 128u) * 8u;
                off = lane16;
            }
            unsigned keep;
            // rows 0 and 15 (with hl = 256: the halo this transform re-reads and the one its successor will)
            // may stay in the L2: no nontemporal hint on their pieces when the launcher says so
            // (WP: piece p carries rows p and p + 8 of the wave's columns -- kept when either row is in the mask)
            const bool shared_row = WP ? (((a.halo_keep >> (unsigned)p) | (a.halo_keep >> (unsigned)(p + 8))) & 1u) != 0
                                       : ((a.halo_keep >> (4u * wv + (unsigned)(p >> 1))) & 1u) != 0;      // halo_keep: a mask over the 16 rows
            if (shared_row)
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(off), "s"(gp), "s"(dst) : "memory");
            else
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(off), "s"(gp), "s"(dst) : "memory");
        }
    };

    v2f nx[16];
    long long blk;
    unsigned kt_next = 0;
    {
        if (t == 0) s_next = few ? blockIdx.x : draw();
        lds_barrier();
        const unsigned kt = __builtin_amdgcn_readfirstlane(s_next);
        if (kt >= a.total) return;            // uniform: more workgroups than transforms
        ch = (int)(kt / nblk32);
        blk = kt - (unsigned)ch * nblk32;
        load_spectrum(ch);
        set_channel(ch);
    }
    bool landed = false;         // DMA: this transform's rows were requested by the previous iteration
    bool counted = false;        // ... and exactly 15 stores were issued after them (vmcnt(15) suffices)
    if (DMA && interior(blk)) {
        dma_rows(in_c, blk);
        landed = true;
    }
    for (;;) {
        v2f v[16];
        if (a.hist_out && blk == 0) carry_history();
        if (DMA && landed) {
            // the DMA pieces are older than the previous transform's stores: waiting for all but the
            // 15 youngest vector-memory operations retires them and leaves the stores in flight
            if (counted) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!WP) lds_barrier();       // every wave's pieces have landed (WP: this wave's own are all it reads)
#pragma unroll
            for (int r = 0; r < 16; r++) nx[r] = lds[base_a + row_a(r)];
        } else load_rows(nx, blk);
        // ---- F1: over n2, twiddle W_4096^(t k2), scatter to [k2][t]
        dft16<-1>(nx);
#pragma unroll
        for (int k = 0; k < 16; k++) {
            v2f x = nx[P16(k)];
            if ((k >> 2) && (k & 3)) x = tw2f(x, q1[k >> 2], p1[k & 3]);
            else if (k >> 2) x = cmul(x, q1[k >> 2]);
            else if (k & 3) x = cmul(x, p1[k & 3]);
            lds[base_a + row_a(k)] = x;
        }
        lds_barrier();
        unsigned drawn = 0;
        if (t == 0) drawn = draw_issue();      // finished two stages further down
        // ---- F2: gather n1 for (k2=hi, n0=lo)
#pragma unroll
        for (int r = 0; r < 16; r++) v[r] = lds[cell_b1(r)];
        dft16<-1>(v);
        if (!NOMID) lds_barrier();
        v2f wmid[16];                             // NOMID only: what would have gone through the LDS
#pragma unroll
        for (int k = 0; k < 16; k++) {
            v2f x = v[P16(k)];
            if ((k >> 2) && (k & 3)) x = tw2f(x, q2[k >> 2], p2[k & 3]);
            else if (k >> 2) x = cmul(x, q2[k >> 2]);
            else if (k & 3) x = cmul(x, p2[k & 3]);
            if constexpr (NOMID) wmid[k] = x;
            else lds[cell_b2(k)] = x;
        }
        if (!NOMID) lds_barrier();
        // ---- F3: gather n0 for (k2=hi, k1=lo); spectrum multiply; first inverse stage
#pragma unroll
        for (int r = 0; r < 16; r++) {
            if constexpr (NOMID) v[r] = wmid[r];
            else v[r] = lds[cell_c(r)];
        }
        dft16<-1>(v);
        {
            // spectrum multiply in place (bin k sits in v[P16(k)]), then the first inverse stage
            // with the transposed schedule, which consumes exactly that order: no shuffling.
#pragma unroll
            for (int k = 0; k < 16; k++) v[P16(k)] = cmul(v[P16(k)], hreg[k]);
            dft16_rev<+1>(v);
            // I1: element n0 goes back to the cell this thread read n0 from (no barrier needed)
#pragma unroll
            for (int k = 0; k < 16; k++) {
                if constexpr (NOMID) wmid[k] = v[k];
                else lds[cell_c(k)] = v[k];
            }
        }
        // every wave "looks at" the draw here, so that the compiler's wait for the atomic sits HERE on every
        // path: left to the one-lane branch below, the other waves' path keeps it pending and a vmcnt(0)
        // appears in front of the stores -- behind the next transform's rows
        asm volatile("" : "+v"(drawn));
        if (t == 0) s_next = draw_finish(drawn);
        lds_barrier();
        kt_next = __builtin_amdgcn_readfirstlane(s_next);
        // the transform after this one: (nch, nb), uniform
        const bool more = kt_next < a.total;
        const int nch = (int)(kt_next / nblk32);
        const long long nb = kt_next - (unsigned)nch * nblk32;
        // per-channel spectra: this transform's has just been multiplied in -- if the next transform belongs to another
        // channel its sixteen bins are requested HERE, under the two inverse stages and the stores still to come, and in
        // front of the next transform's rows (round 4; until then they were requested after the stores, at the top of the
        // next transform: with channel-major tickets every ~6th transform of a workgroup at 64 x 2^24, 3.6-5.7 % of the launch)
        if (HCH && more && nch != ch) load_spectrum(nch);
        // ---- I2: gather k1 for (k2=hi, n0=lo)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            v2f x;
            if constexpr (NOMID) x = wmid[r];
            else x = lds[cell_b2(r)];
            if ((r >> 2) && (r & 3)) x = tw2i(x, q2[r >> 2], p2[r & 3]);
            else if (r >> 2) x = cmul_conj(x, q2[r >> 2]);
            else if (r & 3) x = cmul_conj(x, p2[r & 3]);
            v[r] = x;
        }
        dft16<+1>(v);
        if (!NOMID) lds_barrier();
#pragma unroll
        for (int k = 0; k < 16; k++) lds[cell_b1(k)] = v[P16(k)];
        lds_barrier();
        // ---- I3: gather k2 for n_lo = t
#pragma unroll
        for (int r = 0; r < 16; r++) {
            v2f x = lds[base_a + row_a(r)];
            if ((r >> 2) && (r & 3)) x = tw2i(x, q1[r >> 2], p1[r & 3]);
            else if (r >> 2) x = cmul_conj(x, q1[r >> 2]);
            else if (r & 3) x = cmul_conj(x, p1[r & 3]);
            v[r] = x;
        }
        if constexpr (DMA) {
            // all I3 reads done -> the buffer is free: request the next transform's rows NOW, under
            // this transform's last DFT16 and its stores (WP: this wave's reads of its own region
            // are all that has to be done -- lgkmcnt, no barrier)
            if (WP) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            else lds_barrier();
            landed = more && interior(nb);
            if (landed) dma_rows(in_of(nch), nb);
            counted = landed && !OUT_TX10 && row0 == 1 && blk * a.advance + a.advance <= a.n;
            if constexpr (INTERLEAVE) dft16_head<+1>(v); else dft16<+1>(v);
        } else {
        if constexpr (INTERLEAVE) dft16_head<+1>(v); else dft16<+1>(v);
        lds_barrier();   // LDS free for the next transform
        }
        // (INTERLEAVE: the last four butterflies of that DFT16 run below, next to the stores of the rows each completes)

        if constexpr (PAIR) {
            if constexpr (INTERLEAVE) {
#pragma unroll
                for (int b = 0; b < 4; b++) dft16_tail<+1>(v, b);
            }
            const long long oA = 2 * blk * a.advance - a.hl, oB = oA + a.advance;
            const bool wholeA = 2 * blk * a.advance + a.advance <= a.n, wholeB = wholeA && oB + a.hl + a.advance <= a.n;
            if (OUT_TX10 && wholeB && row0 == 1) {
                if constexpr (OUT_TX10) {
                // real stream into the transmit wire format, the common transform (both segments whole): thirty
                // straight-line group stores through ONE buffer descriptor over the two segments' consecutive output
                // bytes; lane 4g of a quad writes the group, the other three lanes' offsets lie beyond the records
                char *const wb = out_c + (oA >> 2) * 5;                          // segment B's bytes follow at (advance / 4) * 5
                const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<uintptr_t>(wb));
                const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(reinterpret_cast<uintptr_t>(wb) >> 32));
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                    reinterpret_cast<void *>(((uintptr_t)hi << 32) | lo), 0, (2 * FFT_N / 4) * 5, 0x00020000);
                int sel = (t & 3u) ? (int)0x7FFF0000 : (int)((t >> 2) * 5u);
                asm volatile("" : "+v"(sel));
                const int segB = (a.advance >> 2) * 5;                           // uniform
#pragma unroll
                for (int r = 1; r < 16; r++) {
                    const v2f y = v[P16(r)];
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        const unsigned u = (unsigned)((int)(short)(int)((q ? y.y : y.x) * 511.0f) + 512) & 0x3FFu;
                        const unsigned u0 = quad_lane<0>(u), u1 = quad_lane<1>(u), u2 = quad_lane<2>(u), u3 = quad_lane<3>(u);
                        const unsigned w = ((u0 >> 8) | ((u1 >> 8) << 2) | ((u2 >> 8) << 4) | ((u3 >> 8) << 6)) | ((u0 & 0xFFu) << 8) |
                                           ((u1 & 0xFFu) << 16) | ((u2 & 0xFFu) << 24);
                        const int up = 320 * r;                                  // (256 r / 4) groups of 5 bytes
                        __builtin_amdgcn_raw_buffer_store_b32((int)w, rs, sel + up, q ? segB : 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)u3, rs, sel + up + 4, q ? segB : 0, 0);
                    }
                }
                }
            } else
#pragma unroll
            for (int r = 0; r < 16; r++) {
                if (r < row0) continue;
                const v2f y = v[P16(r)];
                if constexpr (OUT_TX10) {
                    // lanes 4g..4g+3 hold 4 consecutive samples: quantise in place, gather the four
                    // 10-bit codes into lane 4g, which writes the 5 bytes of group (row + t)/4
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        const long long o = (q ? oB : oA) + 256 * r;                 // multiple of 256
                        const unsigned u = (unsigned)((int)(short)(int)((q ? y.y : y.x) * 511.0f) + 512) & 0x3FFu;
                        const unsigned u0 = quad_lane<0>(u), u1 = quad_lane<1>(u), u2 = quad_lane<2>(u), u3 = quad_lane<3>(u);
                        // only whole groups of 4 are emitted (the reference's loop steps by 4)
                        if ((t & 3u) == 0 && o + (long long)t + 3 < a.n) {
                            unsigned char *d = reinterpret_cast<unsigned char *>(out_c) + ((o + (long long)t) >> 2) * 5;
                            store_group(d, u0, u1, u2, u3);
                        }
                    }
                } else if (wholeB) {          // both segments inside the stream (the common transform): no per-row tests
                    float *qa = reinterpret_cast<float *>(out_c + (oA + 256 * r) * 4) + t;
                    float *qb = reinterpret_cast<float *>(out_c + (oB + 256 * r) * 4) + t;
                    __builtin_nontemporal_store(ACC ? y.x + *qa : y.x, qa);
                    __builtin_nontemporal_store(ACC ? y.y + *qb : y.y, qb);
                } else {
                if (wholeA || oA + 256 * r + (long long)t < a.n) {
                    float *q = reinterpret_cast<float *>(out_c + (oA + 256 * r) * 4) + t;
                    __builtin_nontemporal_store(ACC ? y.x + *q : y.x, q);
                }
                if (wholeB || oB + 256 * r + (long long)t < a.n) {
                    float *q = reinterpret_cast<float *>(out_c + (oB + 256 * r) * 4) + t;
                    __builtin_nontemporal_store(ACC ? y.y + *q : y.y, q);
                }
                }
            }
        } else {
        const long long obase = blk * a.advance - a.hl;   // uniform; + 256*row + t
        constexpr int OSZ = OUT_C ? 8 : 4;
        const bool whole = blk * a.advance + a.advance <= a.n;
        char *const obase_p = out_c + obase * OSZ;           // uniform
        unsigned voff = t * OSZ;
        asm volatile("" : "+v"(voff));                       // keep the row offsets in the vector register, not in scalar pointers
        // The common transform -- one discarded row, all 3840 outputs inside the stream -- stores its 15 rows
        // unconditionally and in ONE basic block with the last DFT16, so the scheduler may issue a row's store as
        // soon as its radix-4 group is done (DESIGN.md 9 lead (ii)); the guarded form below costs two branches and
        // ~10 scalar instructions per row and starts only after the whole DFT16.  DIAG bit 2: always the guarded form.
        const bool plain_rows = !(DIAG & 4) && whole && row0 == 1;
        if (OUT_TX10 && OUT_C && plain_rows) {
            if constexpr (OUT_TX10 && OUT_C) {
            // complex stream into the transmit wire format, the common transform: fifteen straight-line group stores
            // through ONE buffer descriptor over the transform's 9600 output bytes -- the even lane of a pair writes the
            // group (its offset), the odd lane's offset lies beyond the descriptor's records and the hardware drops its
            // store: no exec mask, no branch and no bounds test per row (the guarded form below has all three)
            typedef int v1i;
            char *const wb = out_c + (obase >> 1) * 5;                          // uniform; row r adds 640 r bytes, rows >= 1 only
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<uintptr_t>(wb));
            const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(reinterpret_cast<uintptr_t>(wb) >> 32));
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                reinterpret_cast<void *>(((uintptr_t)hi << 32) | lo), 0, (FFT_N / 2) * 5, 0x00020000);
            int sel = (t & 1u) ? (int)0x7FFF0000 : (int)((t >> 1) * 5u);
            asm volatile("" : "+v"(sel));
#pragma unroll
            for (int r = 1; r < 16; r++) {
                const v2f y = v[P16(r)];
                const unsigned u0 = (unsigned)((int)(short)(int)(y.x * 511.0f) + 512) & 0x3FFu;
                const unsigned u1 = (unsigned)((int)(short)(int)(y.y * 511.0f) + 512) & 0x3FFu;
                const unsigned u2 = quad_odd(u0), u3 = quad_odd(u1);
                const unsigned w = ((u0 >> 8) | ((u1 >> 8) << 2) | ((u2 >> 8) << 4) | ((u3 >> 8) << 6)) | ((u0 & 0xFFu) << 8) |
                                   ((u1 & 0xFFu) << 16) | ((u2 & 0xFFu) << 24);
                const int up = 640 * r;                                          // (256 r / 2) groups of 5 bytes
                __builtin_amdgcn_raw_buffer_store_b32((v1i)w, rs, sel + (up & 2047), up & ~2047, 0);
                __builtin_amdgcn_raw_buffer_store_b8((unsigned char)u3, rs, sel + (up & 2047) + 4, up & ~2047, 0);
            }
            }
        } else if (plain_rows) {
            // Butterfly b completes rows b, b + 4, b + 8, b + 12 (fft16.h), so their stores could go out while
            // the remaining butterflies run (DESIGN.md 9 lead (ii)).  Written that way (DIAG bit 3, diagnostic
            // library only: variants y / u) the kernels SPILL -- 14 (LDS-DMA) to 26 (register loads) VGPRs, 22-54
            // with the stores pinned by scheduling barriers -- against none for the whole DFT16 first and the
            // fifteen stores behind it, which is what the product does.
#pragma unroll
            for (int b = 0; b < 4; b++) {
                if constexpr (INTERLEAVE) dft16_tail<+1>(v, b);
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int r = b + 4 * c;
                if (r == 0) continue;
                if constexpr (DIAG & 2) continue;
                v2f y = v[P16(r)];
                char *rp = DMA ? obase_p + (voff + (unsigned)(256 * r * OSZ)) : out_c + (obase + 256 * r) * OSZ + (size_t)t * OSZ;
                if constexpr (OUT_C) {
                    if constexpr (ACC) y += *reinterpret_cast<const v2f *>(rp);
                    __builtin_nontemporal_store(y, reinterpret_cast<v2f *>(rp));
                } else {
                    if constexpr (ACC) y.x += *reinterpret_cast<const float *>(rp);
                    __builtin_nontemporal_store(y.x, reinterpret_cast<float *>(rp));
                }
            }
            }
        } else {
        if constexpr (INTERLEAVE) {
#pragma unroll
            for (int b = 0; b < 4; b++) dft16_tail<+1>(v, b);
        }
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const long long orow = obase + 256 * r;          // uniform
            if constexpr (OUT_TX10 && OUT_C) {
                // complex stream (gr-simplefe/lib/sink_c_impl.cc:118-144): lanes 2g, 2g+1 hold two
                // consecutive samples = the four floats of one 5-byte group; lane 2g writes it
                if (r < row0) continue;
                const v2f y = v[P16(r)];
                const unsigned u0 = (unsigned)((int)(short)(int)(y.x * 511.0f) + 512) & 0x3FFu;
                const unsigned u1 = (unsigned)((int)(short)(int)(y.y * 511.0f) + 512) & 0x3FFu;
                const unsigned u2 = quad_odd(u0), u3 = quad_odd(u1);       // the odd partner's codes
                if ((t & 1u) == 0 && orow + (long long)t + 1 < a.n) {
                    unsigned char *d = reinterpret_cast<unsigned char *>(out_c) + ((orow + (long long)t) >> 1) * 5;
                    store_group(d, u0, u1, u2, u3);
                }
                continue;
            }
            if constexpr (DIAG & 2) continue;      // ablation: no output stores (folded below)
            if (r >= row0 && (whole || orow + (long long)t < a.n)) {
                v2f y = v[P16(r)];
                // ONE uniform base for the transform + a per-lane byte offset that steps by a row: fifteen
                // separate 64-bit row pointers would sit in 30 scalar registers across the store burst
                // (the LDS-DMA kernel only: its eight landing addresses already fill the scalar file; the
                // register-load kernels measured 5 % slower this way)
                char *rp = DMA ? obase_p + (voff + (unsigned)(256 * r * OSZ)) : out_c + orow * OSZ + (size_t)t * OSZ;
                if constexpr (OUT_C) {
                    if constexpr (ACC) y += *reinterpret_cast<const v2f *>(rp);
                    __builtin_nontemporal_store(y, reinterpret_cast<v2f *>(rp));
                } else {
                    if constexpr (ACC) y.x += *reinterpret_cast<const float *>(rp);
                    __builtin_nontemporal_store(y.x, reinterpret_cast<float *>(rp));
                }
            }
        }
        }
        if constexpr (DIAG & 2) {   // keeps every result alive behind one store that never happens
            v2f acc = v[0];
#pragma unroll
            for (int r = 1; r < 16; r++) acc += v[r];
            if (acc.x == 1.2345e38f) reinterpret_cast<v2f *>(out_c)[t] = acc;
        }
        }
        // ---- on to the next transform (drawn two stages ago, or blockIdx.x + k gridDim.x)
        if (!more) break;
        blk = nb;
        if (nch != ch) {
            ch = nch;
            set_channel(ch);
        }
    }
    SFE_FIR_STAMP(2);
}


#ifdef SFE_DIAG
#include "diag/fir_fft_diag.inc"
#endif  // SFE_DIAG

}  // namespace

#ifdef SFE_DIAG
// the clock workgroup 0 of the LAST FIR launch on the current device saw (the caller has synchronised): MHz, and its own span in ms
extern "C" int sfe_dsp_diag_fir_clock(double *mhz, double *ms)
{
    unsigned long long c[4];
    SFE_HIP(hipMemcpyFromSymbol(c, HIP_SYMBOL(g_fir_clk), sizeof c));
    const double dr = (double)(c[3] - c[1]);
    if (mhz) *mhz = dr > 0 ? (double)(c[2] - c[0]) / dr * 100.0 : 0.0;
    if (ms) *ms = dr / 1e5;
    return SFE_OK;
}
#endif

bool fir_fft_has_variants(const FirFftArgs &a, int in_complex, int out_complex, int in_u8, int out_tx10, int n_channels, int accumulate)
{
    // LDS-DMA moves 16-byte lanes: every channel's first sample must sit on a 16-byte boundary
    const bool dma_ok = (reinterpret_cast<uintptr_t>(a.in) & 15u) == 0 && (n_channels == 1 || (a.in_stride & 1) == 0);
    return in_complex && out_complex && !in_u8 && !out_tx10 && !accumulate && dma_ok;
}

int launch_fir_fft(const FirFftArgs &a0, int in_complex, int out_complex, int in_u8, int out_tx10, int n_channels,
                   hipStream_t s, int accumulate)
{
    FirFftArgs a = a0;
    if (a.nblk <= 0) return SFE_OK;
    if (a.hist_len < a.hl + a.shift || (a.hist_len & 255) || (a.shift & 255) || ((accumulate || a.shift) && out_tx10)) {
        set_error("fir_fft: bad partition arguments (hist_len=%d hl=%d shift=%d)", a.hist_len, a.hl, a.shift);
        return SFE_EINVAL;
    }
    if (a.hl <= 0 || (a.hl & 255) || a.hl >= FFT_N || a.advance != FFT_N - a.hl) {
        set_error("fir_fft: bad overlap rows (hl=%d advance=%d)", a.hl, a.advance);
        return SFE_EINVAL;
    }
    if (in_complex && !out_complex) {
        set_error("fir_fft: complex input with real output is not a defined combination");
        return SFE_EINVAL;
    }
    if ((out_tx10 || in_u8) && in_complex != out_complex) {
        set_error("fir_fft: wire-format input/output is built for real->real and complex->complex streams");
        return SFE_EINVAL;
    }
    // persistent workgroups: 2 x the resident count (4 per CU at 124 VGPRs / 34 KiB LDS; the finer
    // tail balance measured +2 %), shared over the channels
    int wg_per_cu = 8;
    const bool pair = !in_complex && !out_complex;     // real stream, real taps: two segments per transform
    // LDS-DMA moves 16-byte lanes: every channel's first sample must sit on a 16-byte boundary
    const bool dma_ok = (reinterpret_cast<uintptr_t>(a.in) & 15u) == 0 && (n_channels == 1 || (a.in_stride & 1) == 0);
    // data movement of an aligned cf32 stream: what the caller measured (api_fir.hip: fir_pick_variant), else register loads
    const int var = !dma_ok ? FIR_VAR_REG : (a.variant == FIR_VAR_DMA || a.variant == FIR_VAR_WP ? a.variant : FIR_VAR_REG);
#ifdef SFE_DIAG
    // SFE_FIR_VARIANT = "<waves 2-4><p|n>[s][h]" | "c" | "d" | "e",  SFE_FIR_DIAG = bit 0 no loads, bit 1 no stores,
    // SFE_FIR_WG_PER_CU: read per launch so scripts/ab_fir.py can interleave variants in one process
    const char *ev = getenv("SFE_FIR_VARIANT");
    const int diag = getenv("SFE_FIR_DIAG") ? atoi(getenv("SFE_FIR_DIAG")) & 3 : 0;
    if (const char *e = getenv("SFE_FIR_WG_PER_CU")) wg_per_cu = atoi(e) > 0 ? atoi(e) : wg_per_cu;
#endif
    long long nb = pair ? (a.nblk + 1) / 2 : a.nblk;
    a.nblk = nb;
    // ticketed kernels: ONE grid dimension, transforms of all channels drawn channel-major from a.ticket
    const long long total = nb * n_channels;
    // compute units of the launch's device (cached per device ordinal, common.h)
    const int cus = device_cu_count();
    long long gt = total < (long long)cus * wg_per_cu ? total : (long long)cus * wg_per_cu;
    if (total + gt >= 0xFFFFFFFFLL || !a.ticket) {
        set_error("fir_fft: %lld transforms in one launch exceed the ticket counter", total);
        return SFE_ERANGE;
    }
    a.total = (unsigned)total;
    a.tgroups = FIR_TICKET_GROUPS;
#ifdef SFE_DIAG
    if (const char *e = getenv("SFE_FIR_TGROUPS")) a.tgroups = atoi(e) >= 1 && atoi(e) <= FIR_TICKET_GROUPS_MAX ? atoi(e) : a.tgroups;
#endif
    if ((long long)a.tgroups > gt) a.tgroups = (unsigned)gt;      // every group needs a workgroup to draw for it
    a.ch_groups = a.hs_stride != 0 && a.tgroups == FIR_TICKET_GROUPS && n_channels >= (int)a.tgroups && n_channels % (int)a.tgroups == 0;
#ifdef SFE_DIAG
    if (const char *e = getenv("SFE_FIR_CH_GROUPS")) a.ch_groups = a.ch_groups && atoi(e) != 0;      // A/B: 0 = tickets across the channels as for a shared filter
#endif
    // runs of 8: seven of eight 2 KiB halos are re-read on the XCD whose L2 has just seen them
    // (FETCH_SIZE per launch 4.446 -> see profiles/r02; -1.5 % kernel time, `X` against `X^3`)
    a.tqs = 3;
    a.halo_keep = 0x8001u;      // rows 0 and 15; `X^3` against `X^3!1`: -0.5 ... -1.3 %
#ifdef SFE_DIAG
    if (const char *e = getenv("SFE_FIR_HALO_KEEP")) a.halo_keep = (unsigned)strtoul(e, nullptr, 0);      // any mask of rows
    if (const char *e = getenv("SFE_FIR_TQS")) a.tqs = atoi(e) >= 0 && atoi(e) <= 8 ? (unsigned)atoi(e) : 0u;
#endif
    const dim3 grid((unsigned)gt), block(256);
#define SFE_K(...) hipLaunchKernelGGL((fir_fft4096_kernel<__VA_ARGS__>), grid, block, 0, s, a)
#ifdef SFE_DIAG
    // the bare access-pattern kernels keep round 1's fixed-stride walk: blockIdx.x, + gridDim.x, ... per channel on a 2-D grid
    long long gx = nb;
    const long long cap = ((long long)cus * wg_per_cu + n_channels - 1) / n_channels;
    if (gx > cap) gx = cap < 1 ? 1 : cap;
    const dim3 grid2((unsigned)gx, (unsigned)n_channels);
#include "diag/fir_fft_launch_diag.inc"
#endif
    //    IN_C   OUT_C  IN_U8  PAIR   TX10   DMA    DIAG ACC   WP     HCH
    if (a.hs_stride) {          // per-channel taps: the channel's spectrum is reloaded into registers on a channel change (cf32 streams only)
        if (!in_complex || !out_complex || in_u8 || out_tx10) {
            set_error("fir_fft: per-channel taps are built for complex float32 streams");
            return SFE_EINVAL;
        }
        if (accumulate) SFE_K(true, true, false, false, false, false, 0, true, false, true);
        else if (var != FIR_VAR_REG) SFE_K(true, true, false, false, false, true, 0, false, false, true);
        else SFE_K(true, true, false, false, false, false, 0, false, false, true);
    } else if (accumulate) {           // partitions after the first: out += this partition's result
        if (in_complex) {
            if (in_u8) SFE_K(true, true, true, false, false, false, 0, true);
            else SFE_K(true, true, false, false, false, false, 0, true);
        } else if (out_complex) {
            SFE_K(false, true, false, false, false, false, 0, true);
        } else {
            if (in_u8) SFE_K(false, false, true, true, false, false, 0, true);
            else SFE_K(false, false, false, true, false, false, 0, true);
        }
    } else if (in_complex) {
        if (in_u8 && out_tx10) SFE_K(true, true, true, false, true);      // wire to wire
        else if (out_tx10) SFE_K(true, true, false, false, true);
        else if (in_u8) SFE_K(true, true, true);
        else if (var == FIR_VAR_DMA) SFE_K(true, true, false, false, false, true);      // LDS-DMA early request
        else if (var == FIR_VAR_WP) SFE_K(true, true, false, false, false, true, 0, false, true);      // ... into the wave-private layout
        else SFE_K(true, true);
    } else if (out_complex) {
        SFE_K(false, true);                                               // real data, complex taps
    } else {
        if (in_u8 && out_tx10) SFE_K(false, false, true, true, true);
        else if (out_tx10) SFE_K(false, false, false, true, true);
        else if (in_u8) SFE_K(false, false, true, true);
        else SFE_K(false, false, false, true);
    }
#undef SFE_K
    SFE_HIP(hipGetLastError());
    return SFE_OK;
}

}  // namespace sfe
