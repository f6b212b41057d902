// diag/alloc.hip -- DIAGNOSTIC LIBRARY ONLY (round 5: demoted from the product, DESIGN.md 9 -- the loss of data on a fresh mapping
// was never tied to a cause, and a built pair never reached what the best plain pair gives).  sfe_dsp_malloc_pair / sfe_dsp_probe_pair / sfe_dsp_free: a PAIR of device buffers for a stream call that reads one
// while it writes the other.  Host code, and one small kernel of its own (count_not_held_kernel: the check that a new mapping
// holds what is written to it); no product kernel lives here, which is why scripts/ leave this file out of the kernel hash.
//
// What such a pair gives is fixed when the memory is handed out ("the two modes"): physical memory comes in
// classes, in stretches of tens of GiB; a read stream and a write stream from the same class run ~8 % slower together (8 : 1
// mix; 10-18 % at 1 : 1) than streams from different classes, while each alone runs the same in both -- what one would expect
// of the DRAM ranks of the HBM stacks.  A large hipMalloc is stitched from whatever stretches are free, so a pair of plain
// allocations is a lottery.  Here the pair is BUILT: a pool of 1 GiB physical chunks (hipMemCreate) is classified against one
// of them with the bare 1 : 1 mix (util.hip: pair_probe_kernel), the input is mapped (hipMemMap) from the chunks most like the
// reference and the output from the chunks least like it, into two contiguous virtual ranges, and the rest of the pool is
// released.  Where the virtual-memory calls are not to be had, or the pool shows one class only, the fallback is a screening of
// plain allocations: up to `tries` candidates for the output (more behind 32 GiB spacers while they show no spread).
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdint.h>
#include <unistd.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "../common.h"
#include "sfe_dsp_diag.h"

namespace sfe {
namespace {

constexpr size_t CHUNK = (size_t)1 << 30;

struct Mapped {                                  // one buffer made of chunks: what sfe_dsp_free has to undo
    size_t bytes = 0;                            // of the reserved range
    int device = 0;                              // the device the chunks live on (the free must sync and unmap THERE)
    std::vector<hipMemGenericAllocationHandle_t> chunks;
};
std::mutex g_mu;
std::unordered_map<void *, Mapped> g_mapped;
#ifdef SFE_DIAG
void *g_diag_pool = nullptr;                     // SFE_PAIR_KEEP_POOL=1: the rest of the last classified pool, mapped instead of released
size_t g_diag_pool_chunks = 0;
#endif

// ---- the pair probe (sfe_dsp_probe_pair, sfe_dsp_malloc_pair; DESIGN.md 4.2 "the two modes") ----------------------
// What a pair of allocations gives a kernel that reads one while it writes the other is fixed when the memory is handed
// out: two classes of allocation, a read stream and a write stream from the same class run ~8 % slower together than a pair
// from different classes, while each stream alone runs the same in both (measured: profiles/r04/decimate_modes_parts.txt,
// decimate_modes_pairs.txt).  This kernel is the bare mix, one short-lived workgroup per tile as the bulk kernels launch:
// sixteen rows of 256 8-byte lanes read (32 KiB), `nw` rows of 256 sixteen-byte lanes written, contiguous.
__global__ __launch_bounds__(256) void pair_probe_kernel(const v2f *in, v4f *out, int nw)
{
    const v2f *p = in + (size_t)blockIdx.x * 4096 + threadIdx.x;
    v2f v[16];
#pragma unroll
    for (int u = 0; u < 16; u++) v[u] = __builtin_nontemporal_load(p + 256 * u);
    v2f acc = v[0];
#pragma unroll
    for (int u = 1; u < 16; u++) acc += v[u];
    v4f *q = out + (size_t)blockIdx.x * 256 * nw + threadIdx.x;
    for (int u = 0; u < nw; u++) __builtin_nontemporal_store((v4f){acc.x, acc.y, v[u & 15].x, v[u & 15].y}, q + 256 * u);
}

// one launch of the bare mix over (in, out): tiles of 32 KiB read, the output written in proportion (at least one row of
// 4 KiB per tile, at most 64)
int launch_pair_probe(const void *in, size_t in_bytes, void *out, size_t out_bytes, hipStream_t s)
{
    const size_t tiles = in_bytes / 32768;
    if (tiles == 0 || out_bytes < 4096) return SFE_OK;
    size_t nw = out_bytes / tiles / 4096;                     // rows of 4 KiB per tile that fit the output
    if (nw < 1) nw = 1;
    if (nw > 64) nw = 64;
    size_t t = tiles;
    while (t * nw * 4096 > out_bytes) t--;                     // (an output shorter than one row per tile: fewer tiles)
    if (t == 0 || t > 0x7fffffffu) return SFE_OK;
    hipLaunchKernelGGL(pair_probe_kernel, dim3((unsigned)t), dim3(256), 0, s, static_cast<const v2f *>(in), static_cast<v4f *>(out), (int)nw);
    SFE_HIP(hipGetLastError());
    return SFE_OK;
}


// median time of the bare read + write mix over the pair (util.hip: pair_probe_kernel), on the null stream
int probe_pair_ms(const void *d_in, size_t in_bytes, void *d_out, size_t out_bytes, float *ms)
{
    hipEvent_t e0, e1;
    SFE_HIP(hipEventCreate(&e0));
    if (hipError_t e = hipEventCreate(&e1); e != hipSuccess) {
        (void)hipEventDestroy(e0);
        return hip_fail(e, "probe_pair");
    }
    int rc = SFE_OK;
    float v[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 3 && rc == SFE_OK; i++) rc = launch_pair_probe(d_in, in_bytes, d_out, out_bytes, nullptr);
    for (int i = 0; i < 5 && rc == SFE_OK; i++) {
        hipError_t e = hipEventRecord(e0, nullptr);
        if (e == hipSuccess) rc = launch_pair_probe(d_in, in_bytes, d_out, out_bytes, nullptr);
        if (e == hipSuccess && rc == SFE_OK) e = hipEventRecord(e1, nullptr);
        if (e == hipSuccess && rc == SFE_OK) e = hipEventSynchronize(e1);
        if (e == hipSuccess && rc == SFE_OK) e = hipEventElapsedTime(&v[i], e0, e1);
        if (e != hipSuccess) rc = hip_fail(e, "probe_pair");
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc != SFE_OK) return rc;
    std::sort(v, v + 5);
    *ms = v[2];
    return SFE_OK;
}

void release_chunks(std::vector<hipMemGenericAllocationHandle_t> &h)
{
    for (auto c : h) (void)hipMemRelease(c);
    h.clear();
}

// a contiguous virtual range over the given chunks, read / write for `device`; nullptr on failure (nothing left mapped)
void *map_chunks(const std::vector<hipMemGenericAllocationHandle_t> &h, int device)
{
    void *va = nullptr;
    if (hipMemAddressReserve(&va, h.size() * CHUNK, CHUNK, nullptr, 0) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    size_t done = 0;
    bool ok = true;
    for (; done < h.size() && ok; done++) ok = hipMemMap(static_cast<char *>(va) + done * CHUNK, CHUNK, 0, h[done], 0) == hipSuccess;
    if (ok) {
        hipMemAccessDesc acc = {};
        acc.location.type = hipMemLocationTypeDevice;
        acc.location.id = device;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        ok = hipMemSetAccess(va, h.size() * CHUNK, &acc, 1) == hipSuccess;
    } else {
        done--;                                  // the one that failed was not mapped
    }
    if (!ok) {
        (void)hipGetLastError();
        for (size_t i = 0; i < done; i++) (void)hipMemUnmap(static_cast<char *>(va) + i * CHUNK, CHUNK);
        (void)hipMemAddressFree(va, h.size() * CHUNK);
        return nullptr;
    }
    return va;
}

void unmap_range(void *va, size_t n_chunks)
{
    for (size_t i = 0; i < n_chunks; i++) (void)hipMemUnmap(static_cast<char *>(va) + i * CHUNK, CHUNK);
    (void)hipMemAddressFree(va, n_chunks * CHUNK);
}

// A freshly mapped range must HOLD what is written to it before it is handed out.  Measured (profiles/r04/fir_modes_input.txt,
// blocks 19-21): a kernel launched right after hipMemMap + hipMemSetAccess of chunks into a range that was reserved, mapped,
// unmapped and reserved again moments before can find 50-85 % of the range not backed yet -- its stores are dropped and loads
// return zero (the sparse-range behaviour of a reservation), without a fault; a little later the same addresses hold data.  A
// caller that fills its input straight after sfe_dsp_malloc_pair would lose most of it.  So: a non-zero word is written to the
// whole range and every word of it is read back by a kernel, until two passes 5 ms apart find all of it held.
// (the one kernel of this file: host code's own check, not a product kernel)
__global__ __launch_bounds__(256) void count_not_held_kernel(const uint4 *p, size_t n16, uint32_t mark, unsigned *bad)
{
    unsigned mine = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const uint4 v = p[i];
        mine += (v.x != mark) | (v.y != mark) | (v.z != mark) | (v.w != mark);
    }
    if (mine) atomicAdd(bad, mine);
}

int settle_mapping(void *va, size_t bytes)
{
    const uint32_t mark = 0x3f800000u;           // 1.0f
    unsigned *d_bad = nullptr, bad = 1;
    SFE_HIP(hipMalloc(&d_bad, sizeof(unsigned)));
    int rc = SFE_ESTATE, clean = 0;
    for (int attempt = 0; attempt < 400 && clean < 2; attempt++) {          // up to ~2 s; two clean passes 5 ms apart
        hipError_t e = clean ? hipSuccess : hipMemsetD32(reinterpret_cast<hipDeviceptr_t>(va), (int)mark, bytes / 4);
        if (e == hipSuccess) e = hipMemset(d_bad, 0, sizeof(unsigned));
        if (e == hipSuccess) {
            hipLaunchKernelGGL(count_not_held_kernel, dim3(4096), dim3(256), 0, nullptr, static_cast<const uint4 *>(va), bytes / 16, mark, d_bad);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpy(&bad, d_bad, sizeof(unsigned), hipMemcpyDeviceToHost);
        if (e != hipSuccess) {
            (void)hipFree(d_bad);
            return hip_fail(e, "malloc_pair (settling a mapping)");
        }
        clean = bad == 0 ? clean + 1 : 0;
        if (clean < 2) usleep(5000);
    }
    (void)hipFree(d_bad);
    if (clean >= 2) {
        SFE_HIP(hipMemsetD32(reinterpret_cast<hipDeviceptr_t>(va), 0, bytes / 4));       // handed out zeroed, as fresh memory is
        SFE_HIP(hipDeviceSynchronize());
        rc = SFE_OK;
    } else {
        set_error("malloc_pair: a mapped range of %zu bytes did not come to hold what was written to it", bytes);
    }
    return rc;
}

// The built pair.  SFE_ESTATE: not possible here (no virtual-memory support, too little memory, one class only) -- the caller
// falls back to screening plain allocations.
int build_pair(size_t in_bytes, size_t out_bytes, int tries, void **d_in, void **d_out, float *ms_kept, float *ms_worst)
{
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) return SFE_ESTATE;
    const size_t n_in = (in_bytes + CHUNK - 1) / CHUNK, n_out = (out_bytes + CHUNK - 1) / CHUNK;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return SFE_ESTATE;
    size_t pool = n_in + n_out + (size_t)16 * tries;
    if (pool * CHUNK > free_b / 2) pool = free_b / 2 / CHUNK;
    if (pool < 2 * (n_in + n_out) || pool < 8) return SFE_ESTATE;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || gran == 0 || CHUNK % gran) {
        (void)hipGetLastError();
        return SFE_ESTATE;
    }
    std::vector<hipMemGenericAllocationHandle_t> h;
    for (size_t i = 0; i < pool; i++) {
        hipMemGenericAllocationHandle_t c;
        if (hipMemCreate(&c, CHUNK, &prop, 0) != hipSuccess) {
            (void)hipGetLastError();
            break;
        }
        h.push_back(c);
    }
    if (h.size() < 2 * (n_in + n_out) || h.size() < 8) {
        release_chunks(h);
        return SFE_ESTATE;
    }
    void *va = map_chunks(h, device);
    if (!va) {
        release_chunks(h);
        return SFE_ESTATE;
    }
    // every chunk against a reference chunk with the 1 : 1 mix: slow = the reference's class.  A reference that straddles two
    // stretches gives a smeared picture: up to three references are tried, the one with the widest spread is used.  The chip
    // is taken through its start-up transient first (~100 ms of the same launches): the first probes of a cold process
    // read slow whatever the class.
    const size_t n = h.size();
    int rc = SFE_OK;
    for (int i = 0; i < 300 && rc == SFE_OK; i++)
        rc = launch_pair_probe(va, CHUNK, static_cast<char *>(va) + ((size_t)1 + i % (n - 1)) * CHUNK, CHUNK, nullptr);
    std::vector<float> t(n, 0.0f), best_t;
    float best_spread = 0.0f, best_thr = 0.0f;
    for (size_t ref : {(size_t)0, n / 2, n - 1}) {
        for (size_t c = 0; c < n && rc == SFE_OK; c++)
            if (c != ref) rc = probe_pair_ms(static_cast<char *>(va) + ref * CHUNK, CHUNK, static_cast<char *>(va) + c * CHUNK, CHUNK, &t[c]);
        if (rc != SFE_OK) break;
        std::vector<float> v;
        for (size_t c = 0; c < n; c++)
            if (c != ref) v.push_back(t[c]);
        std::sort(v.begin(), v.end());
        const float p10 = v[v.size() / 10], p90 = v[v.size() - 1 - v.size() / 10];      // (outliers at either end do not count)
        t[ref] = p90;                            // the reference is of its own class
        if (p90 / p10 > best_spread) {
            best_spread = p90 / p10;
            best_thr = 0.5f * (p10 + p90);
            best_t = t;
        }
        if (best_spread >= 1.05f) break;         // two clean levels (the 1 : 1 mix's are ~8 % apart)
    }
    unmap_range(va, n);
    if (rc != SFE_OK || best_spread < 1.04f) {   // one class only in this pool (or the probe failed)
        release_chunks(h);
        return rc != SFE_OK ? rc : SFE_ESTATE;
    }
    // the reference's class (slow against it) and the rest; inside each, the chunks nearest the class's median first
    std::vector<size_t> same, other;
    for (size_t c = 0; c < n; c++) (best_t[c] > best_thr ? same : other).push_back(c);
    auto by_median = [&](std::vector<size_t> &g) {
        if (g.empty()) return;
        std::vector<float> v;
        for (size_t c : g) v.push_back(best_t[c]);
        std::sort(v.begin(), v.end());
        const float m = v[v.size() / 2];
        std::sort(g.begin(), g.end(), [&](size_t a, size_t b) { return fabsf(best_t[a] - m) < fabsf(best_t[b] - m); });
    };
    by_median(same);
    by_median(other);
    // the input from the larger class, the output from the other; and an output from the input's own class to verify against
    std::vector<size_t> &gin = same.size() >= other.size() ? same : other, &gout = same.size() >= other.size() ? other : same;
    if (gin.size() < n_in + n_out || gout.size() < n_out) {
        release_chunks(h);
        return SFE_ESTATE;
    }
    Mapped min, mout, mcheck;
    std::vector<char> used(n, 0);
    for (size_t i = 0; i < n_in; i++) { min.chunks.push_back(h[gin[i]]); used[gin[i]] = 1; }
    for (size_t i = 0; i < n_out; i++) { mout.chunks.push_back(h[gout[i]]); used[gout[i]] = 1; }
    for (size_t i = 0; i < n_out; i++) mcheck.chunks.push_back(h[gin[n_in + i]]);
#ifdef SFE_DIAG
    if (getenv("SFE_PAIR_DEBUG")) {              // diagnostics (libsfe_dsp_diag.so only): which chunks of the pool (in creation order) went where
        fprintf(stderr, "build_pair: pool %zu, spread %.3f, same-class %zu other %zu; in <-", n, best_spread, same.size(), other.size());
        for (size_t i = 0; i < n_in; i++) fprintf(stderr, " %zu (%.3f)", gin[i], best_t[gin[i]]);
        fprintf(stderr, "; out <-");
        for (size_t i = 0; i < n_out; i++) fprintf(stderr, " %zu (%.3f)", gout[i], best_t[gout[i]]);
        fprintf(stderr, "\n  t:");
        for (size_t c = 0; c < n; c++) fprintf(stderr, " %.3f", best_t[c]);
        fprintf(stderr, "\n");
    }
#endif
    void *pin = map_chunks(min.chunks, device), *pout = pin ? map_chunks(mout.chunks, device) : nullptr;
    void *pchk = pout ? map_chunks(mcheck.chunks, device) : nullptr;
    float kept = 0.0f, same_class = 0.0f;
    bool ok = pin && pout && pchk;
    ok = ok && settle_mapping(pin, n_in * CHUNK) == SFE_OK && settle_mapping(pout, n_out * CHUNK) == SFE_OK &&
         settle_mapping(pchk, n_out * CHUNK) == SFE_OK;       // (the check range too: a probe over a range not backed yet reads fast)
    if (ok && in_bytes >= 32768 && out_bytes >= 4096) {
        ok = probe_pair_ms(pin, in_bytes, pout, out_bytes, &kept) == SFE_OK && probe_pair_ms(pin, in_bytes, pchk, out_bytes, &same_class) == SFE_OK;
        ok = ok && kept < 0.97f * same_class;    // the built pair must beat a pair of one class, or the classes were misread
    }
    if (pchk) unmap_range(pchk, n_out);
    std::vector<hipMemGenericAllocationHandle_t> rest;
    for (size_t c = 0; c < n; c++)
        if (!used[c]) rest.push_back(h[c]);
#ifdef SFE_DIAG
    if (ok && getenv("SFE_PAIR_KEEP_POOL") && !rest.empty()) {       // diagnostics: the rest stays, mapped as one range (sfe_dsp_diag_last_pool)
        void *pp = map_chunks(rest, device);
        if (pp) {
            Mapped mp;
            mp.bytes = rest.size() * CHUNK;
            mp.chunks = rest;
            std::lock_guard<std::mutex> lk(g_mu);
            mp.device = device;
            g_mapped[pp] = mp;
            g_diag_pool = pp;
            g_diag_pool_chunks = rest.size();
            rest.clear();
        }
    }
#endif
    release_chunks(rest);                        // the rest of the pool goes back
    if (!ok) {
        if (pin) unmap_range(pin, n_in);
        if (pout) unmap_range(pout, n_out);
        release_chunks(min.chunks);
        release_chunks(mout.chunks);
        return SFE_ESTATE;
    }
    min.bytes = n_in * CHUNK;
    mout.bytes = n_out * CHUNK;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        min.device = mout.device = device;
        g_mapped[pin] = min;
        g_mapped[pout] = mout;
    }
    *d_in = pin;
    *d_out = pout;
    if (ms_kept) *ms_kept = kept;
    if (ms_worst) *ms_worst = same_class;                  // what the same input takes with an output of its own class
    return SFE_OK;
}

// plain allocations, screened: the fallback
int screen_pair(size_t in_bytes, size_t out_bytes, int tries, void **d_in, void **d_out, float *ms_kept, float *ms_worst)
{
    void *in = nullptr, *cand[16] = {nullptr}, *spacer[4] = {nullptr};
    float ms[16];
    SFE_HIP(hipMalloc(&in, in_bytes ? in_bytes : 16));
    int n = 0, rc = SFE_OK, best = 0, n_spacers = 0;
    float worst = 0.0f;
    const bool probe = tries > 1 && in_bytes >= 32768 && out_bytes >= 4096;
    const size_t SPACER = (size_t)32 << 30;
    auto spaced = [&]() {                        // the classes run in stretches of tens of GiB: step over one
        size_t free_b = 0, total_b = 0;
        if (n_spacers < 4 && hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > SPACER + 2 * out_bytes + in_bytes) {
            if (hipMalloc(&spacer[n_spacers], SPACER) == hipSuccess) n_spacers++;
            else (void)hipGetLastError();
        }
    };
    // every candidate stays allocated until the choice is made: a freed one's pages would come straight back.  Up to `tries`
    // candidates; twice as many, each pair of the further ones behind a 32 GiB spacer, while they show no spread (within 4 %:
    // all of one class -- a fresh process tends to be handed what the last one freed)
    for (; n < (probe ? 2 * tries : 1); n++) {
        if (n >= tries && ms[best] < 0.96f * worst) break;
        if (n >= tries && (n - tries) % 2 == 0) spaced();
        const hipError_t e = hipMalloc(&cand[n], out_bytes ? out_bytes : 16);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            if (n == 0) rc = hip_fail(e, "malloc_pair");
            break;                                             // out of memory further on: choose among what there is
        }
        ms[n] = 0.0f;
        if (probe) rc = probe_pair_ms(in, in_bytes, cand[n], out_bytes, &ms[n]);
        if (rc != SFE_OK) {
            n++;
            break;
        }
        if (ms[n] < ms[best]) best = n;
        if (ms[n] > worst) worst = ms[n];
    }
    // still no spread: the class of a pair is an exclusive-or of its two allocations' -- one more allocation for the INPUT,
    // from another stretch, kept if the pair is at least 4 % faster
    if (rc == SFE_OK && probe && n >= 2 && ms[best] >= 0.96f * worst) {
        spaced();
        void *alt = nullptr;
        if (hipMalloc(&alt, in_bytes) == hipSuccess) {
            float t = 0.0f;
            if (probe_pair_ms(alt, in_bytes, cand[best], out_bytes, &t) == SFE_OK && t < 0.96f * ms[best]) {
                (void)hipFree(in);
                in = alt;
                ms[best] = t;
            } else {
                (void)hipFree(alt);
            }
        } else {
            (void)hipGetLastError();
        }
    }
    for (int i = 0; i < n; i++)
        if (rc != SFE_OK || i != best) (void)hipFree(cand[i]);
    for (int i = 0; i < n_spacers; i++) (void)hipFree(spacer[i]);
    if (rc != SFE_OK) {
        (void)hipFree(in);
        return rc;
    }
    *d_in = in;
    *d_out = cand[best];
    if (ms_kept) *ms_kept = ms[best];
    if (ms_worst) *ms_worst = worst;
    return SFE_OK;
}

}  // namespace

bool chunk_mapped(const void *p)
{
    std::lock_guard<std::mutex> lk(g_mu);
    for (const auto &kv : g_mapped) {
        const char *b = static_cast<const char *>(kv.first);
        if (static_cast<const char *>(p) >= b && static_cast<const char *>(p) < b + kv.second.bytes) return true;
    }
    return false;
}

}  // namespace sfe

using namespace sfe;

extern "C" {

int sfe_dsp_probe_pair(const void *d_in, size_t in_bytes, void *d_out, size_t out_bytes, float *ms)
{
    if (!d_in || !d_out || !ms || in_bytes < 32768 || out_bytes < 4096 || (reinterpret_cast<uintptr_t>(d_in) & 7) ||
        (reinterpret_cast<uintptr_t>(d_out) & 15)) {
        set_error("probe_pair: needs an input of >= 32 KiB (8-byte aligned) and an output of >= 4 KiB (16-byte aligned)");
        return SFE_EINVAL;
    }
    return probe_pair_ms(d_in, in_bytes, d_out, out_bytes, ms);
}

int sfe_dsp_malloc_pair(size_t in_bytes, size_t out_bytes, int tries, void **d_in, void **d_out, float *ms_kept, float *ms_worst)
{
    if (!d_in || !d_out || tries < 1 || tries > 8) {
        set_error("malloc_pair: null argument or tries outside 1 .. 8");
        return SFE_EINVAL;
    }
    *d_in = *d_out = nullptr;
    // worth building for streams of a GiB and more (the effect is a large stream's; chunks are 1 GiB)
    if (tries > 1 && in_bytes >= CHUNK && out_bytes >= CHUNK / 4) {
        const int rc = build_pair(in_bytes, out_bytes, tries, d_in, d_out, ms_kept, ms_worst);
        if (rc != SFE_ESTATE) return rc;
    }
    return screen_pair(in_bytes, out_bytes, tries, d_in, d_out, ms_kept, ms_worst);
}

int sfe_dsp_malloc_pair_screened(size_t in_bytes, size_t out_bytes, int tries, void **d_in, void **d_out, float *ms_kept, float *ms_worst)
{
    if (!d_in || !d_out || tries < 1 || tries > 8) {
        set_error("malloc_pair_screened: null argument or tries outside 1 .. 8");
        return SFE_EINVAL;
    }
    *d_in = *d_out = nullptr;
    return screen_pair(in_bytes, out_bytes, tries, d_in, d_out, ms_kept, ms_worst);
}

#ifdef SFE_DIAG
// diagnostics (libsfe_dsp_diag.so, SFE_PAIR_KEEP_POOL=1): the rest of the pool the last built pair was chosen from, as one
// mapped range of n_chunks x 1 GiB in creation order (without the chunks the pair took); free it with sfe_dsp_free
extern "C" int sfe_dsp_diag_last_pool(void **va, size_t *n_chunks)
{
    if (!va || !n_chunks) return SFE_EINVAL;
    *va = g_diag_pool;
    *n_chunks = g_diag_pool_chunks;
    g_diag_pool = nullptr;
    g_diag_pool_chunks = 0;
    return SFE_OK;
}
#endif

int sfe_dsp_mem_kind(const void *dptr, int *kind)
{
    if (!dptr || !kind) return SFE_EINVAL;
    *kind = chunk_mapped(dptr) ? 1 : 0;
    return SFE_OK;
}

int sfe_dsp_free(void *dptr)
{
    if (!dptr) return SFE_OK;
    Mapped m;
    bool mapped = false;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_mapped.find(dptr);
        if (it != g_mapped.end()) {
            m = it->second;
            mapped = true;
        }
    }
    if (!mapped) {
        SFE_HIP(hipFree(dptr));
        return SFE_OK;
    }
    DeviceGuard guard(m.device);                 // the chunks' own device, whatever is current (group calls switch devices)
    SFE_HIP(hipDeviceSynchronize());             // (hipFree waits for the device by itself; the unmapping does not)
    {
        std::lock_guard<std::mutex> lk(g_mu);    // forgotten only once the device is known to be done with it
        g_mapped.erase(dptr);
    }
    unmap_range(dptr, m.chunks.size());
    release_chunks(m.chunks);
    return SFE_OK;
}

}  // extern "C"
