// dec_mode_parts.hip -- the decimator's two modes, taken apart IN ONE PROCESS on the process's own pair of
// buffers (hipMalloc: 8 GiB in, 1 GiB out): the library's decimate by 8 (64 taps, 2^30 cf32), a bare read of the same input, a
// bare write of the same output, and the bare 8 : 1 mix over both.  Run as N fresh processes (scripts/dec_mode_parts.sh):
// which of the bare patterns, if any, moves with the decimator's mode?
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/dec_mode_parts.hip -o scripts/probes/dec_mode_parts -Iinclude -Lsimplefe_amd -lsfe_dsp -Wl,-rpath,$PWD/simplefe_amd
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "sfe_dsp.h"

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define SK(x) do { int r_ = (x); if (r_ != SFE_OK) { fprintf(stderr, "%s: %s\n", #x, sfe_dsp_last_error()); exit(1); } } while (0)

// one workgroup per tile of 4096 samples, as the decimator launches: 16 x 8-byte loads per lane
__global__ __launch_bounds__(256) void k_read(const v2f *in, v2f *sink)
{
    const v2f *p = in + (size_t)blockIdx.x * 4096 + threadIdx.x;
    v2f v[16];
#pragma unroll
    for (int u = 0; u < 16; u++) v[u] = __builtin_nontemporal_load(p + 256 * u);
    v2f acc = v[0];
#pragma unroll
    for (int u = 1; u < 16; u++) acc += v[u];
    if (acc.x == 1.2345e38f) sink[threadIdx.x] = acc;
}
// the tile's 512 outputs: one 16-byte store per lane
__global__ __launch_bounds__(256) void k_write(v4f *out)
{
    const v4f w = {(float)threadIdx.x, 1.0f, 2.0f, 3.0f};
    __builtin_nontemporal_store(w, out + (size_t)blockIdx.x * 256 + threadIdx.x);
}
__global__ __launch_bounds__(256) void k_mix(const v2f *in, v4f *out)
{
    const v2f *p = in + (size_t)blockIdx.x * 4096 + threadIdx.x;
    v2f v[16];
#pragma unroll
    for (int u = 0; u < 16; u++) v[u] = __builtin_nontemporal_load(p + 256 * u);
    v2f acc = v[0];
#pragma unroll
    for (int u = 1; u < 16; u++) acc += v[u];
    __builtin_nontemporal_store((v4f){acc.x, acc.y, acc.x, acc.y}, out + (size_t)blockIdx.x * 256 + threadIdx.x);
}

// the same mix with PLAIN (write-back) stores, and with plain loads too
template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_mix_v(const v2f *in, v4f *out)
{
    const v2f *p = in + (size_t)blockIdx.x * 4096 + threadIdx.x;
    v2f v[16];
#pragma unroll
    for (int u = 0; u < 16; u++) v[u] = NTL ? __builtin_nontemporal_load(p + 256 * u) : p[256 * u];
    v2f acc = v[0];
#pragma unroll
    for (int u = 1; u < 16; u++) acc += v[u];
    const v4f w = {acc.x, acc.y, acc.x, acc.y};
    if (NTS) __builtin_nontemporal_store(w, out + (size_t)blockIdx.x * 256 + threadIdx.x);
    else out[(size_t)blockIdx.x * 256 + threadIdx.x] = w;
}

template <typename F>
static double med(F launch)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < 10; i++) launch();
    std::vector<float> v;
    for (int rep = 0; rep < 9; rep++) {
        CK(hipEventRecord(a));
        for (int i = 0; i < 3; i++) launch();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        v.push_back(ms / 3);
    }
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

// `pairs`: ONE process, two inputs (8 GiB each) and four outputs (1 GiB each): the bare 8 : 1 mix and the decimator for every
// (input, output) combination -- does the mode follow the input, the output, or the pair?
static int pairs()
{
    const size_t N = (size_t)1 << 30, CAP = N / 8 + 8;
    std::vector<float> taps(64);
    for (int i = 0; i < 64; i++) {
        const double k = i - 31.5, x = 0.9 / 8 * k;
        taps[i] = (float)((fabs(x) < 1e-9 ? 1.0 : sin(M_PI * x) / (M_PI * x)) * (0.54 - 0.46 * cos(2 * M_PI * i / 63.0)) / 8.9);
    }
    void *in[2], *out[4];
    for (int i = 0; i < 2; i++) {
        CK(hipMalloc(&in[i], N * 8));
        SK(sfe_dsp_synth_fill(in[i], 2 * N, 20240601u, 0, 0, nullptr));
    }
    for (int j = 0; j < 4; j++) CK(hipMalloc(&out[j], CAP * 8));
    sfe_rs_t r;
    SK(sfe_dsp_rs_create(taps.data(), 64, 1, 4096, 1, 1, 0, SFE_RS_DECIMATE, &r));
    size_t k = 0;
    for (int i = 0; i < 60; i++) SK(sfe_dsp_rs_process_stream(r, in[0], N, N, out[0], CAP, CAP, 8.0f, &k, nullptr));
    const unsigned tiles = (unsigned)(N / 4096);
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 4; j++) {
            const double mx = med([&] { hipLaunchKernelGGL(k_mix, dim3(tiles), dim3(256), 0, 0, (const v2f *)in[i], (v4f *)out[j]); });
            const double dec = med([&] { SK(sfe_dsp_rs_process_stream(r, in[i], N, N, out[j], CAP, CAP, 8.0f, &k, nullptr)); });
            const double m_ps = med([&] { hipLaunchKernelGGL((k_mix_v<true, false>), dim3(tiles), dim3(256), 0, 0, (const v2f *)in[i], (v4f *)out[j]); });
            const double m_pp = med([&] { hipLaunchKernelGGL((k_mix_v<false, false>), dim3(tiles), dim3(256), 0, 0, (const v2f *)in[i], (v4f *)out[j]); });
            const double m_pl = med([&] { hipLaunchKernelGGL((k_mix_v<false, true>), dim3(tiles), dim3(256), 0, 0, (const v2f *)in[i], (v4f *)out[j]); });
            printf("in%d %p out%d %p   mix 8:1 %.4f ms   decimate %.4f ms   mix with plain stores %.4f   plain loads and stores %.4f   plain loads, nt stores %.4f\n", i, in[i], j, out[j], mx, dec, m_ps, m_pp, m_pl);
        }
    sfe_dsp_rs_destroy(r);
    return 0;
}

// `sweep`: ONE arena (one hipMalloc of 26 GiB, physically contiguous as far as a process can arrange it): the input at its
// start, the output at 8 GiB + off for a ladder of offsets, and the input moved too -- which ADDRESS BITS of the pair decide
// the mode?
static int sweep()
{
    const size_t N = (size_t)1 << 30, G = (size_t)1 << 30, M = (size_t)1 << 20;
    char *arena = nullptr;
    CK(hipMalloc(&arena, 26 * G));
    SK(sfe_dsp_synth_fill(arena, (26 * G) / 4, 20240601u, 0, 0, nullptr));
    const unsigned tiles = (unsigned)(N / 4096);
    auto mix = [&](size_t in_off, size_t out_off) {
        return med([&] { hipLaunchKernelGGL(k_mix, dim3(tiles), dim3(256), 0, 0, (const v2f *)(arena + in_off), (v4f *)(arena + out_off)); });
    };
    for (int i = 0; i < 40; i++) hipLaunchKernelGGL(k_mix, dim3(tiles), dim3(256), 0, 0, (const v2f *)arena, (v4f *)(arena + 8 * G));
    printf("arena %p\n", arena);
    const size_t offs[] = {0, 2 * M, 16 * M, 64 * M, 128 * M, 256 * M, 512 * M, 768 * M, 1 * G, 1 * G + 512 * M, 2 * G, 3 * G, 4 * G, 5 * G, 6 * G, 7 * G, 8 * G, 12 * G, 16 * G};
    for (size_t o : offs) printf("input at 0, output at 8 GiB + %6zu MiB: mix %.4f ms\n", o / M, mix(0, 8 * G + o));
    // the input moved with the output fixed behind both
    const size_t ioffs[] = {0, 64 * M, 256 * M, 512 * M, 1 * G, 2 * G, 4 * G, 8 * G};
    for (size_t o : ioffs) printf("input at %6zu MiB, output at 24 GiB: mix %.4f ms\n", o / M, mix(o, 24 * G));
    return 0;
}

// one workgroup per 4096-sample tile: 16 x 8-byte loads, 16 x 8-byte stores (the FIR's 1 : 1 mix)
__global__ __launch_bounds__(256) void k_copy11(const v2f *in, v2f *out)
{
    const v2f *p = in + (size_t)blockIdx.x * 4096 + threadIdx.x;
    v2f *q = out + (size_t)blockIdx.x * 4096 + threadIdx.x;
    v2f v[16];
#pragma unroll
    for (int u = 0; u < 16; u++) v[u] = __builtin_nontemporal_load(p + 256 * u);
#pragma unroll
    for (int u = 0; u < 16; u++) __builtin_nontemporal_store(v[u], q + 256 * u);
}

// `copy`: does the FIR's 1 : 1 mix know of the classes too?  Two inputs and four outputs of 2 GiB each (2^28 cf32), the bare
// copy and the library's 256-tap FIR for every combination.
static int copy_pairs()
{
    const size_t N = (size_t)1 << 28;
    std::vector<float> taps(256);
    for (int i = 0; i < 256; i++) {
        const double k = i - 127.5, x = 0.2 * k;
        taps[i] = (float)((fabs(x) < 1e-9 ? 1.0 : sin(M_PI * x) / (M_PI * x)) * (0.54 - 0.46 * cos(2 * M_PI * i / 255.0)) * 0.2);
    }
    void *in[2], *out[4];
    for (int i = 0; i < 2; i++) {
        CK(hipMalloc(&in[i], N * 8));
        SK(sfe_dsp_synth_fill(in[i], 2 * N, 20240601u, 0, 0, nullptr));
    }
    for (int j = 0; j < 4; j++) CK(hipMalloc(&out[j], N * 8));
    sfe_fir_t f;
    SK(sfe_dsp_fir_create(taps.data(), 256, 0, 1, 1, 0, 0, &f));
    for (int i = 0; i < 100; i++) SK(sfe_dsp_fir_process_stream(f, in[0], out[0], N, N, N, nullptr));
    const unsigned tiles = (unsigned)(N / 4096);
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 4; j++) {
            const double cp = med([&] { hipLaunchKernelGGL(k_copy11, dim3(tiles), dim3(256), 0, 0, (const v2f *)in[i], (v2f *)out[j]); });
            const double fir = med([&] { SK(sfe_dsp_fir_process_stream(f, in[i], out[j], N, N, N, nullptr)); });
            printf("in%d %p out%d %p   copy 1:1 %.4f ms (%.2f TB/s)   FIR %.4f ms (frac %.3f)\n", i, in[i], j, out[j], cp, 16.0 * N / cp / 1e9, fir,
                   16.0 * N / fir / 1e9 / 8000.0);
        }
    sfe_dsp_fir_destroy(f);
    return 0;
}

// `big`: ONE arena of 192 GiB: the input at its start, the output at 8 GiB steps across all of it -- if a class were a high
// physical-address bit (>= 2^35), a contiguous arena this large would show both classes at some offset
static int big()
{
    const size_t N = (size_t)1 << 30, G = (size_t)1 << 30;
    char *arena = nullptr;
    size_t total = 192;
    while (total >= 64 && hipMalloc(&arena, total * G) != hipSuccess) {
        (void)hipGetLastError();
        total -= 32;
    }
    if (!arena) return 1;
    printf("arena %p, %zu GiB\n", arena, total);
    CK(hipMemset(arena, 1, 8 * G));
    const unsigned tiles = (unsigned)(N / 4096);
    auto mix = [&](size_t in_off, size_t out_off) {
        return med([&] { hipLaunchKernelGGL(k_mix, dim3(tiles), dim3(256), 0, 0, (const v2f *)(arena + in_off), (v4f *)(arena + out_off)); });
    };
    for (int i = 0; i < 40; i++) hipLaunchKernelGGL(k_mix, dim3(tiles), dim3(256), 0, 0, (const v2f *)arena, (v4f *)(arena + 8 * G));
    for (size_t o = 8; o + 1 <= total; o += 8) printf("input at 0, output at %3zu GiB: mix %.4f ms\n", o, mix(0, o * G));
    // and a second allocation beside the arena, for the other class if it exists at all in this process
    void *other = nullptr;
    CK(hipMalloc(&other, 2 * G));
    const double t = med([&] { hipLaunchKernelGGL(k_mix, dim3(tiles), dim3(256), 0, 0, (const v2f *)arena, (v4f *)other); });
    printf("input at 0 of the arena, output in ANOTHER allocation: mix %.4f ms\n", t);
    return 0;
}

// `vmm`: can a pair be BUILT?  64 physical chunks of 1 GiB (hipMemCreate), each classified against chunk 0 with the bare 1 : 1
// copy; then an 8 GiB input stitched from chunks of one class and a 1 GiB output from the other (hipMemMap into contiguous
// virtual ranges), against an input and an output both from one class: the 8 : 1 mix and the decimator on both pairs.
static int vmm()
{
    const size_t G = (size_t)1 << 30, N = (size_t)1 << 30;
    const int NCH = 64;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    std::vector<hipMemGenericAllocationHandle_t> h(NCH);
    char *va = nullptr;
    CK(hipMemAddressReserve((void **)&va, NCH * G, G, nullptr, 0));
    int got = 0;
    for (; got < NCH; got++) {
        if (hipMemCreate(&h[got], G, &prop, 0) != hipSuccess) { (void)hipGetLastError(); break; }
        CK(hipMemMap(va + got * G, G, 0, h[got], 0));
    }
    CK(hipMemSetAccess(va, got * G, &acc, 1));
    CK(hipMemset(va, 1, got * G));
    // classify: copy chunk 0 -> chunk c (1 GiB : 1 GiB)
    const unsigned tiles1 = (unsigned)(G / 8 / 4096);
    std::vector<double> t(got, 0.0);
    for (int c = 1; c < got; c++)
        t[c] = med([&] { hipLaunchKernelGGL(k_copy11, dim3(tiles1), dim3(256), 0, 0, (const v2f *)va, (v2f *)(va + c * G)); });
    double lo = 1e9, hi = 0;
    for (int c = 1; c < got; c++) { lo = std::min(lo, t[c]); hi = std::max(hi, t[c]); }
    printf("%d chunks; copy chunk 0 -> chunk c: min %.4f max %.4f ms\n  ", got, lo, hi);
    for (int c = 1; c < got; c++) printf("%.3f ", t[c]);
    printf("\n");
    std::vector<int> same, other;            // same class as chunk 0 = the slow copies
    same.push_back(0);
    for (int c = 1; c < got; c++) (t[c] > 0.5 * (lo + hi) ? same : other).push_back(c);
    printf("class of chunk 0: %zu chunks, other class: %zu chunks\n  ", same.size(), other.size());
    for (int c = 1; c < got; c++) printf("%c", t[c] > 0.5 * (lo + hi) ? 'S' : 'o');
    printf("\n");
    if (hi < 1.04 * lo || same.size() < 9 || other.size() < 9) { printf("no two classes with nine chunks each here\n"); return 0; }
    // stitch: input A = 8 chunks of `same`, output X = 1 chunk of `other` (different classes), output Y = 1 chunk of `same`
    char *vin = nullptr, *vx = nullptr, *vy = nullptr, *vin2 = nullptr;
    CK(hipMemAddressReserve((void **)&vin, 8 * G, G, nullptr, 0));
    CK(hipMemAddressReserve((void **)&vin2, 8 * G, G, nullptr, 0));
    CK(hipMemAddressReserve((void **)&vx, 2 * G, G, nullptr, 0));
    CK(hipMemAddressReserve((void **)&vy, 2 * G, G, nullptr, 0));
    for (int i = 0; i < 8; i++) CK(hipMemMap(vin + i * G, G, 0, h[same[i]], 0));            // (a chunk may be mapped twice)
    for (int i = 0; i < 8; i++) CK(hipMemMap(vin2 + i * G, G, 0, h[other[i]], 0));
    for (int i = 0; i < 2; i++) CK(hipMemMap(vx + i * G, G, 0, h[other[8 + i - (other.size() < 10 ? 1 : 0) * i]], 0));
    for (int i = 0; i < 2; i++) CK(hipMemMap(vy + i * G, G, 0, h[same[8 + i - (same.size() < 10 ? 1 : 0) * i]], 0));
    CK(hipMemSetAccess(vin, 8 * G, &acc, 1));
    CK(hipMemSetAccess(vin2, 8 * G, &acc, 1));
    CK(hipMemSetAccess(vx, 2 * G, &acc, 1));
    CK(hipMemSetAccess(vy, 2 * G, &acc, 1));
    const unsigned tiles = (unsigned)(N / 4096);
    auto mix = [&](const char *i, char *o) { return med([&] { hipLaunchKernelGGL(k_mix, dim3(tiles), dim3(256), 0, 0, (const v2f *)i, (v4f *)o); }); };
    printf("input of class 0, output of class 1: mix %.4f ms\n", mix(vin, vx));
    printf("input of class 0, output of class 0: mix %.4f ms\n", mix(vin, vy));
    printf("input of class 1, output of class 0: mix %.4f ms\n", mix(vin2, vy));
    printf("input of class 1, output of class 1: mix %.4f ms\n", mix(vin2, vx));
    return 0;
}

int main(int argc, char **argv)
{
    if (argc > 1 && argv[1][0] == 'v') return vmm();
    if (argc > 1 && argv[1][0] == 'b') return big();
    if (argc > 1 && argv[1][0] == 'c') return copy_pairs();
    if (argc > 1 && argv[1][0] == 's') return sweep();
    if (argc > 1 && argv[1][0] == 'p') return pairs();
    const size_t N = (size_t)1 << 30, CAP = N / 8 + 8;
    std::vector<float> taps(64);
    for (int i = 0; i < 64; i++) {
        const double k = i - 31.5, x = 0.9 / 8 * k;
        taps[i] = (float)((fabs(x) < 1e-9 ? 1.0 : sin(M_PI * x) / (M_PI * x)) * (0.54 - 0.46 * cos(2 * M_PI * i / 63.0)) / 8.9);
    }
    void *in = nullptr, *out = nullptr;
    CK(hipMalloc(&in, N * 8));
    CK(hipMalloc(&out, CAP * 8));
    SK(sfe_dsp_synth_fill(in, 2 * N, 20240601u, 0, 0, nullptr));
    sfe_rs_t r;
    SK(sfe_dsp_rs_create(taps.data(), 64, 1, 4096, 1, 1, 0, SFE_RS_DECIMATE, &r));
    size_t k = 0;
    for (int i = 0; i < 60; i++) SK(sfe_dsp_rs_process_stream(r, in, N, N, out, CAP, CAP, 8.0f, &k, nullptr));     // the chip's start-up transient
    const unsigned tiles = (unsigned)(N / 4096);
    const double dec = med([&] { SK(sfe_dsp_rs_process_stream(r, in, N, N, out, CAP, CAP, 8.0f, &k, nullptr)); });
    const double rd = med([&] { hipLaunchKernelGGL(k_read, dim3(tiles), dim3(256), 0, 0, (const v2f *)in, (v2f *)out); });
    const double wr = med([&] { hipLaunchKernelGGL(k_write, dim3(tiles), dim3(256), 0, 0, (v4f *)out); });
    const double mx = med([&] { hipLaunchKernelGGL(k_mix, dim3(tiles), dim3(256), 0, 0, (const v2f *)in, (v4f *)out); });
    const double dec2 = med([&] { SK(sfe_dsp_rs_process_stream(r, in, N, N, out, CAP, CAP, 8.0f, &k, nullptr)); });
    printf("in %p out %p  decimate %.4f ms (again %.4f)  read 8 GiB %.4f ms (%.2f TB/s)  write 1 GiB %.4f ms (%.2f TB/s)  mix 8:1 %.4f ms (%.2f TB/s)\n", in, out, dec,
           dec2, rd, 8.0 * N / rd / 1e9, wr, 8.0 * (N / 8) / wr / 1e9, mx, 9.0 * N / mx / 1e9);
    sfe_dsp_rs_destroy(r);
    return 0;
}
