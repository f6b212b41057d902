// host.h -- what the host-side translation units of libsfe_dsp.so share: the handles behind sfe_fir_t / sfe_rs_t,
// the plan caches, and the helpers that cross files.  HOST CODE ONLY (no kernels; not part of the kernel-source hash).
//   api.hip        errors, devices, memory, timers, the synthetic stream, the wire-format converters
//   api_plans.hip  folding (U, step, pos0) into the polyphase kernels' tap tables and spectra
//   api_fir.hip    the FIR handle: tables, partitions, variants, carried state, sfe_dsp_fir_*
//   api_rs.hip     the resample / decimate handle: time law, run memo, kernel choice, sfe_dsp_rs_*
//   api_pipe.hip   the pinned host pipe over either handle, sfe_dsp_*_pipe_*
//   group.hip      channel blocks over several devices, sfe_dsp_*_group_* (over the public C ABI only)
#pragma once
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <new>
#include <tuple>
#include <numeric>
#include <unordered_map>
#include <utility>
#include <vector>

#include "common.h"
#include "timelaw.h"

namespace sfe {

// DeviceGuard: common.h
#define SFE_ON_DEVICE(dev)                                   \
    DeviceGuard guard__(dev);                                \
    if (!guard__.ok) {                                       \
        set_error("cannot select device %d", (int)(dev));    \
        return SFE_EHIP;                                     \
    }

bool ranges_overlap(const void *a, size_t an, const void *b, size_t bn);      // [a, a + an) and [b, b + bn) share a byte
int use_device(int device);                                                    // hipSetDevice with the range / no-GPU checks
bool stream_is_capturing(hipStream_t s);

// ---- plan caches (api_plans.hip)
struct PlanCache {
    std::map<std::pair<int, long long>, PolyTiledPlan> plans;   // (step, pos0) -> plan
    void clear()
    {
        for (auto &kv : plans) {
            if (kv.second.d_G) (void)hipFree(kv.second.d_G);
            if (kv.second.d_Gt) (void)hipFree(kv.second.d_Gt);
        }
        plans.clear();
    }
};
struct FftPlanCache {
    std::map<std::pair<int, long long>, PolyFftPlan> plans;
    void clear()
    {
        for (auto &kv : plans) {
            if (kv.second.d_H) (void)hipFree(kv.second.d_H);
            if (kv.second.d_tw) (void)hipFree(kv.second.d_tw);
        }
        plans.clear();
    }
};
struct MfmaCache {
    std::map<std::pair<int, long long>, PolyMfmaPlan> plans;
    void clear()
    {
        for (auto &kv : plans)
            if (kv.second.d_A) (void)hipFree(kv.second.d_A);
        plans.clear();
    }
};
// nullptr: the shape has no such kernel (the caller takes the next one); *rc != SFE_OK: the upload failed
const PolyTiledPlan *get_tiled_plan(PlanCache &cache, const std::vector<float> &taps_pm, int U, int plen, int step, long long pos0,
                                    int *rc);
const PolyFftPlan *get_fft_plan(FftPlanCache &cache, const std::vector<float> &taps_pm, int U, int plen, int step, long long pos0,
                                int fft_mode, int *rc);
const PolyMfmaPlan *get_mfma_plan(MfmaCache &cache, const std::vector<float> &taps_pm, int U, int plen, int step, long long pos0,
                                  int *rc);

// ---- the FIR handle (api_fir.hip)
struct Fir {
    uint32_t magic = 0x46495231u;   // 'FIR1': catches stale or foreign handles
    int n_taps = 0, taps_complex = 0, data_complex = 0, out_complex = 0, n_channels = 1;
    int device = 0, algo = SFE_FIR_ALGO_AUTO, in_u8 = 0, out_tx10 = 0;
    int blk = 0, block_hint = 0;
    int hl = 0;                 // carried history per channel, samples (multiple of 256)
    int ovl = 0;                // overlap of one transform (multiple of 256): each launch applies `ovl` (+1) taps
    int variant = FIR_VAR_AUTO; // data movement of the cf32 kernel: measured per device and shape unless sfe_dsp_fir_set_variant fixed it
    int last_variant = FIR_VAR_AUTO, cal_runs = 0;      // what the last bulk call ran; calibrations this handle made
    float cal_ms[FIR_VAR_COUNT] = {0.0f, 0.0f, 0.0f};   // medians of this handle's last calibration, by variant
    int piped = 0;              // pipes alive over this handle: they froze its item formats, so the format setters refuse
    int per_channel = 0;        // taps given per channel ([n_channels][n_taps]): one spectrum set per channel
    int parts = 1;              // partitions of the tap vector, one launch each (filters longer than one overlap)
    bool fft_ok = false;
    v2f *d_hs = nullptr, *d_tw1 = nullptr, *d_tw2 = nullptr;
    unsigned *d_ticket = nullptr;   // work counter of the persistent FFT kernel (zero between launches)
    float *d_taps = nullptr;    // real taps for the direct kernel
    std::vector<float> h_taps;  // host copy (direct-kernel plan)
    std::vector<float> h_taps_all;   // every tap as given at create (complex pairs / per-channel rows included): re-planning
    PlanCache plans;
    void *d_hist[2] = {nullptr, nullptr};
    int cur = 0;
    bool captured = false;      // a call of this handle sits in a hipGraph that names d_hist[cur]: the state stays there (fir_run)
    bool started = false;       // samples have gone through since create / reset
    // class-compatible host block path
    float *h_buf = nullptr;     // pinned, block_hint+2 floats (blkconv.cxx:44 sizes it so)
    void *d_blk_in = nullptr, *d_blk_out = nullptr;
    void *h_blk_out = nullptr;  // pinned: the kernel's output of a zero-copy block (then copied over h_buf)
    // host-pointer streaming path (sfe_dsp_fir_process_host): chunked pinned + device staging
    void *h_stage = nullptr, *d_st_in = nullptr, *d_st_out = nullptr;
    void *h_stage_out = nullptr;
    size_t stage_samples = 0;
    // Calls of at most zc_max samples skip the two DMA copies: the kernel reads the pinned host buffer
    // and writes a pinned host buffer itself (one launch + one wait instead of copy, launch, copy, wait).
    // Only where the kernel reads its input once (parts == 1).  sfe_dsp_fir_set_zero_copy_max; 0 disables.
    size_t zc_max = (size_t)1 << 20;
    hipStream_t stream = nullptr;
    size_t hist_bytes() const { return (size_t)n_channels * hl * (data_complex ? 8 : 4); }
};
Fir *as_fir(void *h);
void fir_free(Fir *f);
int fir_create_impl(const float *taps, int n_taps, int taps_complex, int data_complex, int n_channels, int block_hint, int device,
                    int per_channel, sfe_fir_t *out);
int fir_run(Fir *f, const void *d_in, void *d_out, size_t n, size_t in_stride, size_t out_stride, hipStream_t s);

// ---- the resample / decimate handle (api_rs.hip)
struct Rs {
    uint32_t magic = 0x52533031u;   // 'RS01'
    int U = 1, n_taps = 0, plen = 0, blksize = 0, data_complex = 0, n_channels = 1;
    int device = 0, mode = SFE_RS_RESAMPLE, exact_stream = 0, in_u8 = 0;
    int piped = 0;                         // pipes alive over this handle (they froze its input format)
    int fft_mode = 0;                      // sfe_dsp_rs_set_algo: 1 force the transform-domain kernel, -1 never, 0 the calibrated rule
    int use_mfma = 0;                      // sfe_dsp_rs_set_algo(SFE_RS_ALGO_MFMA): the matrix-pipe form (measured slower; opt-in)
    int hl = 0;
    float *d_taps = nullptr;               // [U][plen] phase-major
    std::vector<float> h_taps_pm;          // host copy of the same (tiled plans)
    PlanCache plans;
    MfmaCache mfma_plans;
    FftPlanCache fft_plans;
    unsigned *d_ticket = nullptr;          // work counters of the transform-domain kernel
    struct Fir *gen_tables = nullptr;      // general rate in the transform domain (poly_gen.hip): the U phases' spectra and the
                                           // twiddle bases, built by the FIR's own table builder (one "channel" per phase)
    bool gen_tried = false;
    void *d_hist[2] = {nullptr, nullptr};
    int cur = 0;
    bool captured = false;                 // a call of this handle sits in a hipGraph that names d_hist[cur] (see fir_carry_state)
    sfe_rs_timestate ts = {0, 0.0f, 0};
    // class-compatible host path staging (one channel)
    void *d_in = nullptr, *d_out = nullptr;
    long long *d_pos = nullptr;
    float *d_mu = nullptr;
    size_t out_cap = 0, sched_cap = 0;
    void *h_stage = nullptr;               // pinned: in/out staging
    size_t h_stage_bytes = 0;
    // wire-format input on a shape no fused u8 kernel takes (steps beyond the tiled kernels, a general rate outside the transform kernel):
    // the call's bytes are converted ONCE into this grow-only float32 buffer and the float path runs (api_rs.hip: sfe_dsp_rs_process_stream)
    float *d_u8f = nullptr;
    size_t d_u8f_floats = 0;
    hipStream_t u8f_stream = nullptr;
    bool u8_refused = false;               // set by the implementation where it would have refused a u8 call, nothing launched, no state touched
    long long *h_pos = nullptr;            // pinned schedule staging
    float *h_mu = nullptr;
    void *d_segs = nullptr, *d_chunks = nullptr, *h_segs = nullptr, *h_chunks = nullptr;   // run-length plans
    size_t segs_cap = 0, chunks_cap = 0;
    hipEvent_t ev_plan = nullptr;          // recorded behind a call's plan uploads: the pinned staging is free again once it fires
    hipStream_t plan_stream = nullptr;     // ... on this stream (the device-side plan arrays are ordered by it)
    // General rate: the plan of one blksize-sample reference call depends only on the time state the call
    // starts in, and that state is a multiple of the float32 grid of the call's LAST binade inside
    // [-1, step) -- a few thousand possible values (blksize*U = 16384: 2^-10 apart) -- so plans are
    // memoised per start state: a 2^28-sample call replays 65 536 reference calls as table look-ups
    // instead of 65 536 x ~40 runs of float arithmetic (36 ms -> ~2 ms on the host), and the run table
    // lives on the device across calls (only what is new is uploaded).
    struct SegPlanRef {
        int seg_first, n_seg, n_out;
        sfe_rs_timestate after;
        int next = -1;                     // index of the plan for the state this call ends in, once it has been met:
                                           // a stream of full-size calls then walks the plans by index, no hashing
    };
    std::unordered_map<uint64_t, int> seg_memo;      // start state -> index into seg_refs
    std::vector<SegPlanRef> seg_refs;
    std::vector<TlSeg> seg_table;          // runs of the memoised calls, in the order they were first met
    size_t seg_uploaded = 0;               // leading entries of seg_table already in d_segs
    float memo_rate = 0.0f;                // the memo is for one (rate, blksize)
    int memo_m = 0;
    hipStream_t stream = nullptr;
    int esz() const { return data_complex ? 8 : 4; }
};
Rs *as_rs(void *h);
void rs_free(Rs *r);

// poly_rt_dma.hip (round 5): the runtime-shape tiled kernel for odd input steps, its tile fetched by LDS-DMA.  SFE_ESTATE: the shape or
// the buffers are outside what it takes -- the caller runs launch_poly_tiled.  (Declared here, not in common.h: host-side plumbing.)
int launch_poly_rt_dma(const PolyTiledPlan &plan, const PolyTiledArgs &a, int data_complex, int in_u8, int n_channels, hipStream_t s);
bool poly_rt_dma_window_shape(int SP, int UP);

}  // namespace sfe
