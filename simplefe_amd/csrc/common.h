// common.h -- shared declarations of libsfe_dsp.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/sfe_dsp.h"

typedef float v2f __attribute__((ext_vector_type(2)));   // one cf32 sample, packed-math friendly
typedef float v4f __attribute__((ext_vector_type(4)));

namespace sfe {

// Workgroup barrier for LDS hand-offs that does NOT drain global memory: __syncthreads() emits
// s_waitcnt vmcnt(0) lgkmcnt(0), which stalls every wave until its prefetch loads and streaming
// stores have completed; the LDS exchange only needs lgkmcnt(0).  (cdna_hip_programming.md,
// 'Pipelining across barriers'.)
__device__ __forceinline__ void lds_barrier()
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#endif
}

void set_error(const char *fmt, ...);
int hip_fail(hipError_t e, const char *what);
// Compute units of the CURRENT device (256 on an unpartitioned MI355X; a partitioned one reports
// its share), cached per device ordinal, thread-safe (api.hip).  The persistent grids are sized
// from it; 256 if the attribute cannot be read.
int device_cu_count();

// Every entry point that touches a handle runs on the handle's device and puts the caller's
// current device back afterwards (the library must not change the caller's HIP context state).
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int device)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) ok = hipSetDevice(device) == hipSuccess;
    }
    ~DeviceGuard()
    {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
};

#define SFE_HIP(call)                                         \
    do {                                                      \
        hipError_t e__ = (call);                              \
        if (e__ != hipSuccess) return ::sfe::hip_fail(e__, #call); \
    } while (0)

// ---- FIR -----------------------------------------------------------------------------
constexpr int FFT_N = 4096;         // in-LDS transform length (16 x 16 x 16)
constexpr int FFT_ROWS = 16;        // rows of 256 samples; thread t owns column t
constexpr int LDS_K2_STRIDE = 272;  // padded 256 (== 16 mod 32: conflict-free both layouts)
constexpr int LDS_K1_STRIDE = 17;
constexpr int WP_PITCH = 144;       // wave-private first/last layout: rows (i, i+8) of a wave's 64 columns, then 16 cells of pad
constexpr int WP_REGION = 8 * WP_PITCH;   // cells per wave

struct FirFftArgs {
    const void *in;       // channel 0 input
    void       *out;
    const void *hist;     // [n_channels][hl] samples preceding `in`
    void       *hist_out; // if non-null (needs n >= hl): the kernel also writes the NEXT call's
                          // history, in[n-hl .. n) as float32, saving the separate carry-over launch
    const v2f  *hs;       // [16][256] taps spectrum / 4096, per-thread order (see fir_fft.hip)
    const v2f  *tw1;      // [7][256]   rows 1..3: W_4096^(t k), rows 4..6: W_4096^(4 t k)
    const v2f  *tw2;      // [7][16]    rows 1..3: W_256^(n0 k),  rows 4..6: W_256^(4 n0 k)
    long long   n;        // samples per channel in this call
    long long   in_stride, out_stride;   // samples
    int         hl;       // overlap rows * 256 = FFT_N - advance (the taps this launch applies reach back hl samples)
    int         advance;  // valid outputs per transform
    int         hist_len; // samples of history in front of `in` (per channel), multiple of 256, >= hl + shift
    int         shift;    // this launch filters the stream delayed by `shift` samples (partition p: p * hl)
    long long   nblk;     // transforms per channel
    unsigned   *ticket;   // [FIR_TICKET_GROUPS_MAX][32] device counters (128 bytes apart) the persistent workgroups
                          // draw transforms from; zero between launches
    unsigned    total;    // set by the launcher: transforms over all channels (channel-major tickets)
    unsigned    tgroups;  // set by the launcher: counters in use (workgroup b draws from counter b % tgroups)
    unsigned    ch_groups; // set by the launcher (per-channel spectra, channels a multiple of tgroups): group g draws the transforms of
                           // channels g, g + tgroups, ... channel by channel, so that a workgroup changes channel -- reloads its
                           // spectrum registers -- once per channels / tgroups of the launch instead of every few transforms
    long long   hs_stride; // 0: one spectrum for every channel; else channel c's spectra start c*hs_stride elements after hs (per-channel taps: held in registers like the shared one and reloaded, 16 loads per thread, when the workgroup's next transform belongs to another channel)
    unsigned    tqs;      // set by the launcher: a group draws runs of 2^tqs CONSECUTIVE transforms (their halos meet in its L2)
    unsigned    halo_keep; // set by the launcher: mask of input rows loaded WITHOUT the nontemporal hint (0x8001: the two rows a neighbour re-reads)
    int         variant;  // how an aligned complex float32 stream's rows reach the transform (FIR_VAR_*); others ignore it
};
// Data-movement variants of the cf32 kernel (same arithmetic, bit-identical results):
//   REG  tickets + guarded register loads (16 global_load_dwordx2 per thread)
//   DMA  tickets + LDS-DMA requested early into the padded exchange layout, rows 0 / 15 kept in L2
//   WP   DMA into a wave-private exchange layout (two workgroup barriers fewer per transform)
// Which is fastest differs by a few percent between boxes of one pool (-5.7 % ... +3.7 % for DMA
// against REG), so the handle measures them on its first large call per device and shape
// (api_fir.hip: fir_pick_variant) instead of compiling one in.
enum { FIR_VAR_AUTO = -1, FIR_VAR_REG = 0, FIR_VAR_DMA = 1, FIR_VAR_WP = 2, FIR_VAR_COUNT = 3 };
// true when the launch described by (a, flags) has more than one data-movement variant
bool fir_fft_has_variants(const FirFftArgs &a, int in_complex, int out_complex, int in_u8, int out_tx10, int n_channels, int accumulate);
constexpr int FIR_TICKET_GROUPS = 8;        // one per XCD under round-robin workgroup placement
constexpr int FIR_TICKET_GROUPS_MAX = 64;
// in_u8: the input stream is the device wire format, u8 offset binary (one byte per real sample,
// an (I, Q) byte pair per complex sample); it is converted on load, (b - 128) * (1/127)
// (gr-simplefe/lib/source_c_impl.cc:121-132).  History stays float32.
// out_tx10: real output packed as 10-bit offset binary, 4 samples in 5 bytes (sink_f_impl.cc:117-143)
// accumulate: add to the output instead of overwriting it (partitions after the first)
int launch_fir_fft(const FirFftArgs &a, int in_complex, int out_complex, int in_u8, int out_tx10, int n_channels,
                   hipStream_t s, int accumulate = 0);

struct PolyArgs {
    const void *in;        // channel 0 input (n_in samples)
    void       *out;
    const void *hist;      // [n_channels][hl] samples preceding `in`
    const float *taps;     // [U][plen] phase-major real taps (or [2][U][plen] re,im planes)
    long long   n_in, in_stride, out_stride;
    int         hl;
    int         U, plen;
    // integer-step law: output k sits at upsampled position pos0 + k*step
    long long   pos0;
    int         step;
    long long   n_out;
    // scheduled law (general rate): per-output position / weight arrays
    const long long *sched_pos;
    const float     *sched_mu;
};
int launch_poly_int(const PolyArgs &a, int data_complex, int taps_complex, int exact,
                    int n_channels, hipStream_t s);

// Tiled integer-step kernel (the measured decimate / resample path).  With g = gcd(step, U),
// UP = U/g outputs are produced per SP = step/g input samples; output k = UP*m + r reads
// x[SP*m + o_r - j] against phase ph_r, (o_r, ph_r) fixed by pos0.  The host folds that into
// zero-padded tap rows G[r][q], q ascending in time, all of one length Lp (a multiple of SP),
// so every lane runs the same straight-line loop (polyphase.hip: poly_tiled_kernel).
struct PolyTiledPlan {
    int    SP = 0, UP = 0, Lp = 0, e_max = 0;
    float *d_G = nullptr;        // [UP][Lp]
    float *d_Gt = nullptr;       // [Lp][gt_pitch]: the same taps, one row of all UP phases per local time (poly_rt_kernel: one scalar load per tap)
    int    gt_pitch = 8;         // floats per row of d_Gt: 8, or UP rounded up to a multiple of 8 for the shapes with 9 ... 64 outputs per period
};
struct PolyTiledArgs {
    const void *in;
    void       *out;
    const void *hist;
    void       *hist_out = nullptr;   // if non-null (needs n_in >= hl): one extra workgroup per channel writes the next call's history
    unsigned    tiles = 0;            // set by the launcher: tiles per channel (blockIdx.x == tiles is that extra workgroup)
    unsigned    win = 0;              // > 1: workgroup i takes tile (i % win) * ceil(tiles / win) + i / win -- the resident workgroups
                                      // read `win` separate windows of the stream instead of one (experiment)
    unsigned    tlb_ahead = 0;        // > 0: one lane of every workgroup touches the input and the output `tlb_ahead` tiles ahead of its own,
                                      // so that the address translation of that part of the stream is resident when its workgroups start
    unsigned    tpw = 1;              // poly_stream_kernel: consecutive tiles per workgroup (the extra workgroup is blockIdx.x == ceil(tiles / tpw))
    const float *G;
    const float *Gt = nullptr;        // PolyTiledPlan::d_Gt
    long long   n_in, in_stride, out_stride, n_out;
    int         hl, Lp, e_max;
    // poly_rt_kernel (any SP, UP as launch arguments; set by the launcher)
    int         SP = 0, UP = 0, tm = 0, rowlen = 0;
    unsigned    sp_inv = 0;           // ceil(2^32 / SP): floor(s / SP) = mulhi(s, sp_inv) for the s a tile meets
    unsigned    y_off = 0;            // byte offset in LDS of the waves' output regions (0: none; UP >= 3, complex)
    // poly_rt_dma.hip: the input is the receive wire format (u8 offset binary; complex: byte pairs) -- the tile's raw bytes land at raw_off in LDS
    int         in_u8 = 0;
    unsigned    raw_off = 0;
    int         gt_pitch = 8;         // PolyTiledPlan::gt_pitch
};
// returns SFE_OK, or SFE_ESTATE when (SP, UP, Lp) has no tiled instantiation (caller falls
// back to launch_poly_int)
int launch_poly_tiled(const PolyTiledPlan &plan, const PolyTiledArgs &a, int data_complex, int exact,
                      int in_u8, int n_channels, hipStream_t s);
bool poly_tiled_supported(int SP, int UP, int Lp);     // a compile-time instantiation, or the runtime-shape kernel
bool poly_tiled_is_compiled(int SP, int UP, int Lp);    // a compile-time instantiation of poly_tiled_kernel
bool poly_tiled_u8_is_compiled(int SP, int UP, int Lp); // ... that also has a u8-input form

// f32-MFMA form of the same integer-step law (fused multiply-add numerics only).  Outputs are
// taken in groups of RG = UP*DM consecutive outputs (DM consecutive m, RG <= 16) that read a
// window of Kp input samples starting GS = SP*DM samples apart: D[16 x 16] += A[16 x 4] B[4 x 16]
// with rows = outputs of a group, columns = (group, re|im), K = window position.  The host
// builds A (taps, zero where a row does not reach a window position) in fragment order.
struct PolyMfmaPlan {
    int    GS = 0, RG = 0, Kp = 0, u_lo = 0;   // u_lo: window start relative to GS*g
    float  density = 0.0f;                     // useful MACs / issued MACs
    float *d_A = nullptr;                      // [Kp/4][64] fragments, K ascending = tap index ascending
};
struct PolyMfmaArgs {
    const void *in;
    void       *out;
    const void *hist;
    const float *A;
    long long   n_in, in_stride, out_stride, n_out;
    int         hl, GS, RG, Kp, u_lo;
    int         x_bytes;   // set by the launcher: bytes of one wave's sample / output tile slice
    int         a_bytes;   // set by the launcher: bytes of the tap-fragment table in front of the slices
    int         gs;        // set by the launcher: group spacing inside a column block (bank spread)
    long long   tiles;     // set by the launcher
};
int launch_poly_mfma(const PolyMfmaArgs &a, int n_channels, hipStream_t s);   // cf32 data only
bool poly_mfma_fits(int GS, int RG, int Kp);
// Transform-domain form of the integer-step law (cf32 or real f32 data, fused numerics): the UP output phases
// are UP filters on the SP input polyphase components, all at the low (1/SP) rate, so a block of
// 256 low-rate points costs SP forward and UP inverse 256-point FFTs plus UP*SP multiplies per
// bin -- ~100 flop per input sample for the 5/3, 381-tap headline shape instead of ~300 for the
// direct dot products, which takes that shape from VALU-bound to HBM-bound (poly_fft.hip).
struct PolyFftPlan {
    int    SP = 0, UP = 0, R = 0, Li = 0, e_max = 0;
    float *d_H = nullptr;        // [UP][SP][256] complex: spectra of the sub-filters, / 256
    float *d_tw = nullptr;       // [6][16] complex: W_256^(l k), k=1..3 and W_256^(4 l k), k=1..3
};
struct PolyFftArgs {
    const void *in;
    void       *out;
    const void *hist;
    void       *hist_out;        // if non-null (needs n_in >= hl): the kernel also writes the NEXT call's history,
                                 // in[n_in - hl .. n_in) as float32, saving the separate carry-over launch
    const float *H, *tw;
    long long   n_in, in_stride, out_stride, n_out;
    long long   n_pass;          // set by the launcher: passes of R segments per channel
    unsigned    total;           // set by the launcher (work counters): passes over all channels, channel-major tickets
    int         hl, e_max, ovl, V;
    unsigned   *ticket;          // [POLY_TICKET_GROUPS][32] work counters, zero between launches (null: fixed-stride walk)
    unsigned    tgroups;         // set by the launcher
    unsigned    tqs;             // set by the launcher: a counter deals runs of 2^tqs consecutive passes
    unsigned    halo_keep;       // set by the launcher: 1 = a pass's first and last staged rows are loaded without the nontemporal hint
};
constexpr unsigned POLY_TICKET_GROUPS = 8;
// returns SFE_ESTATE when (SP, UP) has no instantiation
int launch_poly_fft(const PolyFftPlan &plan, const PolyFftArgs &a, int data_complex, int in_u8, int n_channels, hipStream_t s);
// segments per pass for (SP, UP), 0 when the shape has no instantiation
int poly_fft_segments(int SP, int UP);
int launch_poly_sched(const PolyArgs &a, int data_complex, int exact, int n_channels,
                      hipStream_t s);

// General rate, run-length form (timelaw.h): the host plans each blksize-sample call of the
// reference as a handful of constant-increment runs; one workgroup per call expands its runs into
// (pos, mu) on the fly and evaluates the two polyphase dots from an LDS tile of the call's input.
struct SegChunk {
    long long in_off;     // first input sample of the call (relative to this launch's input)
    long long k_first;    // first output of the call (relative to this launch's output)
    int       m;          // input samples in the call
    int       n_out;      // outputs of the call
    int       seg_first;  // index of the call's first run in the run table
    int       n_seg;
};
struct PolySegArgs {
    const void *in;
    void       *out;
    const void *hist;
    const float *taps;          // [U][plen] phase-major
    const void *segs;           // TlSeg[]
    const SegChunk *chunks;
    long long   n_in, in_stride, out_stride;
    int         hl, U, plen, n_chunks, max_m;
    int         taps_global = 0;    // set by the launcher: the taps stay in memory (more of them than the LDS holds beside a call's samples)
    int         split = 1, tile_cap = 0;   // set by the launcher: workgroups per reference call, samples a workgroup's tile holds
    int         span_slack = 64;    // samples beyond max_m / split a part's outputs may reach: ceil(rate) + the recurrence's wobble (api_rs.hip)
};
// returns SFE_ESTATE when a call's tile does not fit in LDS (caller falls back to launch_poly_sched)
int launch_poly_seg(const PolySegArgs &a, int data_complex, int exact, int n_channels, hipStream_t s);

// The same law in the transform domain (poly_gen.hip): one forward 4096-point transform of an input block, U
// spectrum products and inverse transforms (all U phase samples of every input, libdsp/resample.cxx:100-114), the
// outputs picked and blended from LDS by the same runs.  Complex float32, fused numerics, rate >= 1.
struct PolyGenArgs {
    const void *in;
    void       *out;
    const void *hist;
    const v2f  *hs;             // [U][16][256]: the phases' spectra / 4096 in the FIR kernel's thread order
    const v2f  *tw1, *tw2;      // the FIR kernel's twiddle bases
    const void *segs;           // TlSeg[]
    const SegChunk *chunks;
    long long   n_in, in_stride, out_stride;
    int         hl, U, plen, ovl, blksize, n_chunks;
    int         adv;            // input samples a block owns (launcher: 4096 - ovl, fewer for rates below ~1)
    int         real;           // a real float32 stream: two consecutive blocks per transform (poly_gen.hip: REAL)
    int         in_u8;          // the stream is u8 offset binary, converted on load (poly_gen.hip: IN_U8)
    int         diag_T;         // (diagnostic library's prologue ablation only: outputs per block of the synthetic table)
};
// SFE_ESTATE: outside what the kernel takes (caller: launch_poly_seg)
int launch_poly_gen(const PolyGenArgs &a, int max_runs, float step, int n_channels, hipStream_t s);

// new_hist[i] = virtual[n_in - hl + i], virtual = old_hist ++ in  (per channel)
int launch_history_update(const void *in, long long n_in, long long in_stride,
                          const void *old_hist, void *new_hist, int hl, int elem_floats,
                          int n_channels, hipStream_t s, int in_u8 = 0);

// u8 offset binary -> float, bit-exact with the reference converter
__device__ __forceinline__ float u8_to_f32(unsigned b) { return ((float)b - 128.0f) * (1.0f / 127.0f); }

int launch_synth_fill(float *d, uint64_t n, uint32_t seed, uint32_t ch, uint64_t first,
                      hipStream_t s);
int launch_rx_u8_to_f32(const uint8_t *src, float *dst, size_t n, hipStream_t s);
int launch_tx_f32_to_10bit(const float *src, uint8_t *dst, size_t n_floats, hipStream_t s);

}  // namespace sfe
