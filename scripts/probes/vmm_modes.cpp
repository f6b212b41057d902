// vmm_modes.cpp -- does the decimator's slow mode follow HOW ITS BUFFERS ARE BACKED PHYSICALLY?
//
// Same process, same kernel (libsfe_dsp.so's decimate by 8, 64 taps, 2^30 cf32 -> 2^27), three ways of getting the
// 8 GiB + 1 GiB:
//   M  hipMalloc                                   (what torch / sfe_dsp_malloc do)
//   V  the virtual-memory API, ONE physical allocation per buffer (hipMemCreate of the whole size), mapped
//   C  the virtual-memory API, one physical allocation per CHUNK of `chunk` bytes (2 MiB .. 1 GiB), mapped back to back
//      into one reserved address range: every chunk is physically contiguous and aligned by construction, so the
//      driver can map it with fragments of at least the chunk's size whatever the state of the rest of the memory
// Each variant is allocated, timed (HIP events, median of 9 x 3 launches) and freed, twice, interleaved.
//   build: hipcc -O2 scripts/probes/vmm_modes.cpp -o scripts/probes/vmm_modes -Isimplefe_amd/../include -Lsimplefe_amd -lsfe_dsp -Wl,-rpath,$PWD/simplefe_amd
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "sfe_dsp.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define SK(x) do { int r_ = (x); if (r_ != SFE_OK) { fprintf(stderr, "%s: %s\n", #x, sfe_dsp_last_error()); exit(1); } } while (0)

struct Vm {
    void *va = nullptr, *res = nullptr;
    size_t size = 0, res_size = 0;
    std::vector<hipMemGenericAllocationHandle_t> h;
};

static Vm vm_alloc(size_t bytes, size_t chunk, size_t va_shift = 0)
{
    Vm v;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    if (chunk == 0) chunk = bytes;
    chunk = (chunk + gran - 1) / gran * gran;
    v.size = (bytes + chunk - 1) / chunk * chunk;
    // va_shift: the mapping starts that many bytes behind a chunk-aligned address (is it the alignment of the VIRTUAL range
    // that matters, or the size of the physical pieces?)
    void *res = nullptr;
    CK(hipMemAddressReserve(&res, v.size + va_shift, chunk < ((size_t)1 << 30) ? chunk : ((size_t)1 << 30), nullptr, 0));
    v.res = res;
    v.res_size = v.size + va_shift;
    v.va = (char *)res + va_shift;
    for (size_t off = 0; off < v.size; off += chunk) {
        hipMemGenericAllocationHandle_t h;
        CK(hipMemCreate(&h, chunk, &prop, 0));
        CK(hipMemMap((char *)v.va + off, chunk, 0, h, 0));
        v.h.push_back(h);
    }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(v.va, v.size, &acc, 1));
    return v;
}

static void vm_free(Vm &v)
{
    CK(hipMemUnmap(v.va, v.size));
    for (auto h : v.h) CK(hipMemRelease(h));
    CK(hipMemAddressFree(v.res, v.res_size));
    v = Vm();
}

int main(int argc, char **argv)
{
    const size_t N = (size_t)1 << 30, CAP = N / 8 + 8;
    std::vector<float> taps(64);
    for (int i = 0; i < 64; i++) {                       // any low-pass will do: the time does not depend on the values
        const double k = i - 31.5, x = 0.9 / 8 * k;
        taps[i] = (float)((fabs(x) < 1e-9 ? 1.0 : sin(M_PI * x) / (M_PI * x)) * (0.54 - 0.46 * cos(2 * M_PI * i / 63.0)) / 8.9);
    }
    sfe_rs_t r;
    SK(sfe_dsp_rs_create(taps.data(), 64, 1, 4096, 1, 1, 0, SFE_RS_DECIMATE, &r));
    sfe_timer_t tm;
    SK(sfe_dsp_timer_create(&tm));
    auto time_it = [&](void *in, void *out) -> double {
        size_t k = 0;
        for (int i = 0; i < 40; i++) SK(sfe_dsp_rs_process_stream(r, in, N, N, out, CAP, CAP, 8.0f, &k, nullptr));
        std::vector<float> v;
        for (int rep = 0; rep < 9; rep++) {
            SK(sfe_dsp_timer_start(tm, nullptr));
            for (int i = 0; i < 3; i++) SK(sfe_dsp_rs_process_stream(r, in, N, N, out, CAP, CAP, 8.0f, &k, nullptr));
            SK(sfe_dsp_timer_stop(tm, nullptr));
            float ms = 0;
            SK(sfe_dsp_timer_elapsed_ms(tm, &ms));
            v.push_back(ms / 3);
        }
        std::sort(v.begin(), v.end());
        return v[v.size() / 2];
    };
    struct Variant { const char *name; size_t chunk; int kind; size_t shift; };       // kind 0 hipMalloc, 1 VMM
    const Variant set_a[] = {{"M  hipMalloc", 0, 0, 0},
                             {"V  hipMemCreate, one allocation per buffer", 0, 1, 0},
                             {"C  hipMemCreate, 2 MiB chunks", (size_t)2 << 20, 1, 0},
                             {"C  hipMemCreate, 32 MiB chunks", (size_t)32 << 20, 1, 0},
                             {"C  hipMemCreate, 1 GiB chunks", (size_t)1 << 30, 1, 0}};
    const Variant set_b[] = {{"M  hipMalloc", 0, 0, 0},
                             {"C  hipMemCreate, 128 MiB chunks", (size_t)128 << 20, 1, 0},
                             {"C  hipMemCreate, 256 MiB chunks", (size_t)256 << 20, 1, 0},
                             {"C  hipMemCreate, 512 MiB chunks", (size_t)512 << 20, 1, 0},
                             {"C  hipMemCreate, 1 GiB chunks", (size_t)1 << 30, 1, 0},
                             {"C  1 GiB chunks, mapped 2 MiB off alignment", (size_t)1 << 30, 1, (size_t)2 << 20},
                             {"C  hipMemCreate, 2 GiB chunks", (size_t)2 << 30, 1, 0}};
    const int rounds = argc > 1 ? atoi(argv[1]) : 2;
    const bool second = argc > 2 && argv[2][0] == 'b';
    const Variant *vars = second ? set_b : set_a;
    const int nvars = second ? 7 : 5;
    for (int round = 0; round < rounds; round++)
        for (int vi_ = 0; vi_ < nvars; vi_++) {
            const Variant &v = vars[vi_];
            void *in = nullptr, *out = nullptr;
            Vm vi, vo;
            if (v.kind == 0) {
                CK(hipMalloc(&in, N * 8));
                CK(hipMalloc(&out, CAP * 8));
            } else {
                vi = vm_alloc(N * 8, v.chunk, v.shift);
                vo = vm_alloc(CAP * 8, v.chunk, v.shift);
                in = vi.va;
                out = vo.va;
            }
            SK(sfe_dsp_synth_fill(in, 2 * N, 20240601u, 0, 0, nullptr));
            SK(sfe_dsp_rs_reset(r));
            const double ms = time_it(in, out);
            printf("round %d  %-46s in %p out %p  median %.4f ms  frac %.3f\n", round, v.name, in, out, ms, 9.0 * N / (ms * 1e-3) / 8e12);
            fflush(stdout);
            if (v.kind == 0) {
                CK(hipFree(in));
                CK(hipFree(out));
            } else {
                vm_free(vi);
                vm_free(vo);
            }
        }
    sfe_dsp_rs_destroy(r);
    return 0;
}
