// hbm_mix.hip -- what this box's HBM gives for pure reads, pure writes and a 1:1 copy, with the
// lane widths and cache policies the FIR kernel could use.  The FIR kernel's memory floor is the
// copy figure, not the read figure: the guide's ~6.3 TB/s is a read rate.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/hbm_mix.hip -o scripts/probes/hbm_mix && scripts/probes/hbm_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); exit(1); } } while (0)

template <typename T, bool NT, int U>
__global__ __launch_bounds__(256) void k_read(const T *in, T *out, size_t n)
{
    T acc = {};
    const size_t stride = (size_t)gridDim.x * 256 * U;
    for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i + 256 * (U - 1) < n; i += stride) {
        T v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = NT ? __builtin_nontemporal_load(in + i + 256 * u) : in[i + 256 * u];
#pragma unroll
        for (int u = 0; u < U; u++) acc += v[u];
    }
    if (acc[0] == 1.2345e38f) out[threadIdx.x] = acc;
}
template <typename T, bool NT, int U>
__global__ __launch_bounds__(256) void k_write(T *out, size_t n)
{
    T v = {};
    v[0] = (float)threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256 * U;
    for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i + 256 * (U - 1) < n; i += stride) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (NT) __builtin_nontemporal_store(v, out + i + 256 * u);
            else out[i + 256 * u] = v;
        }
    }
}
template <typename T, bool NTL, bool NTS, int U>
__global__ __launch_bounds__(256) void k_copy(const T *in, T *out, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256 * U;
    for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i + 256 * (U - 1) < n; i += stride) {
        T v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = NTL ? __builtin_nontemporal_load(in + i + 256 * u) : in[i + 256 * u];
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (NTS) __builtin_nontemporal_store(v[u], out + i + 256 * u);
            else out[i + 256 * u] = v[u];
        }
    }
}

// 8 parts read, 1 part written (the decimator's mix): a workgroup reads CH*256 8-byte lanes and writes CH*32
template <int CH>
__global__ __launch_bounds__(256) void k_mix81(const v2f *in, v2f *out, size_t n_tiles)
{
    for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const v2f *p = in + tile * (size_t)(CH * 256) + threadIdx.x;
        v2f v[CH];
#pragma unroll
        for (int u = 0; u < CH; u++) v[u] = __builtin_nontemporal_load(p + 256 * u);
        v2f acc = v[0];
#pragma unroll
        for (int u = 1; u < CH; u++) acc += v[u];
        // the tile's CH*32 output samples = CH*16 sixteen-byte lanes, contiguous, as the decimator's store is
        // (every lane stores, so that no lane's loads are dead: 8 bytes per lane for CH = 8, 16 for CH = 16, 2 x 16 for CH = 32)
        if constexpr (CH == 8) {
            __builtin_nontemporal_store(acc, out + tile * (size_t)(CH * 32) + threadIdx.x);
        } else {
            const v4f w = {acc.x, acc.y, acc.x, acc.y};
            for (unsigned q = threadIdx.x; q < (unsigned)(CH * 16); q += 256)
                __builtin_nontemporal_store(w, reinterpret_cast<v4f *>(out + tile * (size_t)(CH * 32)) + q);
        }
    }
}

// 5 parts read, 3 parts written (the 5/3 resampler's mix): a workgroup reads 10 x 256 8-byte lanes and writes 6 x 256
// (the resampler's pass is 2310 samples in, 1386 out), all loads first, as that kernel issues them
__global__ __launch_bounds__(256) void k_mix53(const v2f *in, v2f *out, size_t n_tiles)
{
    for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const v2f *p = in + tile * (size_t)(10 * 256) + threadIdx.x;
        v2f v[10];
#pragma unroll
        for (int u = 0; u < 10; u++) v[u] = __builtin_nontemporal_load(p + 256 * u);
        v2f *q = out + tile * (size_t)(6 * 256) + threadIdx.x;
#pragma unroll
        for (int u = 0; u < 6; u++) __builtin_nontemporal_store(v[u] + v[9 - u], q + 256 * u);
    }
}

// 1 part read, UP parts written (the interpolators' mix, x2 / x4 / x8): a workgroup reads NR x 256 8-byte lanes and writes
// NR UP / 2 x 256 sixteen-byte lanes, contiguous (what poly_rt_kernel's wave-private output regions produce)
template <int UP, int NR>
__global__ __launch_bounds__(256) void k_mix1u(const v2f *in, v4f *out, size_t n_tiles)
{
    for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const v2f *p = in + tile * (size_t)(NR * 256) + threadIdx.x;
        v2f v[NR];
#pragma unroll
        for (int u = 0; u < NR; u++) v[u] = __builtin_nontemporal_load(p + 256 * u);
        v4f *q = out + tile * (size_t)(NR * UP / 2 * 256) + threadIdx.x;
#pragma unroll
        for (int u = 0; u < NR * UP / 2; u++) {
            const v2f a = v[u % NR], b = v[(u + 1) % NR];
            __builtin_nontemporal_store((v4f){a.x, a.y, b.x, b.y}, q + 256 * u);
        }
    }
}

template <typename F>
static void timeit(const char *name, double bytes, F launch)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const char *re = getenv("HBM_MIX_R");            // HBM_MIX_R=2: a short run for counter collection
    const int R = re ? atoi(re) : 30;
    for (int i = 0; i < (re ? 1 : 5); i++) launch();
    float best = 1e9f, sum = 0;
    for (int i = 0; i < R; i++) {
        CK(hipEventRecord(a));
        launch();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
        sum += ms;
    }
    printf("%-44s mean %.4f ms (%.2f TB/s)   best %.4f ms (%.2f TB/s)\n", name, sum / R, bytes / (sum / R) / 1e9, best, bytes / best / 1e9);
}

int main(int argc, char **argv)
{
    const size_t bytes = (size_t)2 << 30;      // 2 GiB in, 2 GiB out: the FIR headline's buffers
    void *in, *out;
    CK(hipMalloc(&in, bytes));
    CK(hipMalloc(&out, bytes));
    CK(hipMemset(in, 1, bytes));
    CK(hipMemset(out, 0, bytes));
    if (argc > 1 && argv[1][0] == '8') {       // the decimator's 8:1 mix by tile size: one workgroup per tile of CH x 256 samples
        const size_t n8 = bytes / 8;
        printf("-- one workgroup per tile\n");
        timeit("mix 8:1, tile 4096 samples (16 loads per lane)", 1.125 * bytes, [&] { hipLaunchKernelGGL((k_mix81<16>), dim3((unsigned)(n8 / (16 * 256))), dim3(256), 0, 0, (const v2f *)in, (v2f *)out, n8 / (16 * 256)); });
        timeit("mix 8:1, tile 2048 samples (8 loads per lane)", 1.125 * bytes, [&] { hipLaunchKernelGGL((k_mix81<8>), dim3((unsigned)(n8 / (8 * 256))), dim3(256), 0, 0, (const v2f *)in, (v2f *)out, n8 / (8 * 256)); });
        timeit("mix 8:1, tile 8192 samples (32 loads per lane)", 1.125 * bytes, [&] { hipLaunchKernelGGL((k_mix81<32>), dim3((unsigned)(n8 / (32 * 256))), dim3(256), 0, 0, (const v2f *)in, (v2f *)out, n8 / (32 * 256)); });
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'p') {       // a copy IN PLACE (every line read, then written back) against the same copy between two buffers
        for (int g : {2048, 8192, 65536}) {
            printf("-- grid %d x 256 threads\n", g);
            timeit("copy   8 B lanes, nt, 16 deep, in -> out", 2.0 * bytes, [&] { hipLaunchKernelGGL((k_copy<v2f, true, true, 16>), dim3(g), dim3(256), 0, 0, (const v2f *)in, (v2f *)out, bytes / 8); });
            timeit("copy   8 B lanes, nt, 16 deep, in place", 2.0 * bytes, [&] { hipLaunchKernelGGL((k_copy<v2f, true, true, 16>), dim3(g), dim3(256), 0, 0, (const v2f *)in, (v2f *)in, bytes / 8); });
            timeit("copy  16 B lanes, nt, 8 deep, in place", 2.0 * bytes, [&] { hipLaunchKernelGGL((k_copy<v4f, true, true, 8>), dim3(g), dim3(256), 0, 0, (const v4f *)in, (v4f *)in, bytes / 16); });
        }
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'i') {       // the interpolators' mixes: 2^27 / 2^26 / 2^25 samples in so that the output stays 2 GiB
        printf("-- one workgroup per tile (32 KiB written per workgroup)\n");
        { const size_t t = (bytes / 2 / 8) / (8 * 256); timeit("mix 1:2, 8 x 8 B loads, 16 x 16 B stores", 1.5 * bytes, [&] { hipLaunchKernelGGL((k_mix1u<2, 8>), dim3((unsigned)t), dim3(256), 0, 0, (const v2f *)in, (v4f *)out, t); }); }
        { const size_t t = (bytes / 4 / 8) / (4 * 256); timeit("mix 1:4, 4 x 8 B loads, 16 x 16 B stores", 1.25 * bytes, [&] { hipLaunchKernelGGL((k_mix1u<4, 4>), dim3((unsigned)t), dim3(256), 0, 0, (const v2f *)in, (v4f *)out, t); }); }
        { const size_t t = (bytes / 8 / 8) / (2 * 256); timeit("mix 1:8, 2 x 8 B loads, 16 x 16 B stores", 1.125 * bytes, [&] { hipLaunchKernelGGL((k_mix1u<8, 2>), dim3((unsigned)t), dim3(256), 0, 0, (const v2f *)in, (v4f *)out, t); }); }
        return 0;
    }
    if (argc > 1 && argv[1][0] == '4') {       // the real-data FIR's lane width: 4-byte lanes against 8 and 16 (same bytes, same grid)
        for (int g : {2048, 8192}) {
            printf("-- grid %d x 256 threads\n", g);
            timeit("copy   4 B lanes, nt load + nt store, 32 deep", 2.0 * bytes, [&] { hipLaunchKernelGGL((k_copy<float, true, true, 32>), dim3(g), dim3(256), 0, 0, (const float *)in, (float *)out, bytes / 4); });
            timeit("copy   8 B lanes, nt load + nt store, 16 deep", 2.0 * bytes, [&] { hipLaunchKernelGGL((k_copy<v2f, true, true, 16>), dim3(g), dim3(256), 0, 0, (const v2f *)in, (v2f *)out, bytes / 8); });
            timeit("copy  16 B lanes, nt load + nt store, 8 deep", 2.0 * bytes, [&] { hipLaunchKernelGGL((k_copy<v4f, true, true, 8>), dim3(g), dim3(256), 0, 0, (const v4f *)in, (v4f *)out, bytes / 16); });
        }
        return 0;
    }
    if (argc > 1 && argv[1][0] == '5') {       // only the resampler's mix (profiles/r03/hbm_mix_5to3.txt)
        const size_t tiles = bytes / 8 / (10 * 256);
        for (int g : {1024, 2048, 8192, 65536, (int)tiles}) {
            printf("-- grid %d x 256 threads\n", g);
            timeit("mix 5:3, 10 x 8 B loads, 6 x 8 B stores", 1.6 * (double)(tiles * 10 * 256 * 8), [&] { hipLaunchKernelGGL(k_mix53, dim3(g), dim3(256), 0, 0, (const v2f *)in, (v2f *)out, tiles); });
        }
        return 0;
    }
    const int grids[] = {2048, 8192, 65536, 262144};
    for (int g : grids) {
        printf("-- grid %d x 256 threads\n", g);
        const size_t n16 = bytes / 16, n8 = bytes / 8;
        timeit("read  16 B lanes, nt, 4 in flight", (double)bytes, [&] { hipLaunchKernelGGL((k_read<v4f, true, 4>), dim3(g), dim3(256), 0, 0, (const v4f *)in, (v4f *)out, n16); });
        timeit("read   8 B lanes, nt, 8 in flight", (double)bytes, [&] { hipLaunchKernelGGL((k_read<v2f, true, 8>), dim3(g), dim3(256), 0, 0, (const v2f *)in, (v2f *)out, n8); });
        timeit("write 16 B lanes, nt", (double)bytes, [&] { hipLaunchKernelGGL((k_write<v4f, true, 4>), dim3(g), dim3(256), 0, 0, (v4f *)out, n16); });
        timeit("write  8 B lanes, nt", (double)bytes, [&] { hipLaunchKernelGGL((k_write<v2f, true, 8>), dim3(g), dim3(256), 0, 0, (v2f *)out, n8); });
        timeit("write 16 B lanes, plain", (double)bytes, [&] { hipLaunchKernelGGL((k_write<v4f, false, 4>), dim3(g), dim3(256), 0, 0, (v4f *)out, n16); });
        timeit("copy  16 B lanes, nt load + nt store", 2.0 * bytes, [&] { hipLaunchKernelGGL((k_copy<v4f, true, true, 4>), dim3(g), dim3(256), 0, 0, (const v4f *)in, (v4f *)out, n16); });
        timeit("copy  16 B lanes, plain", 2.0 * bytes, [&] { hipLaunchKernelGGL((k_copy<v4f, false, false, 4>), dim3(g), dim3(256), 0, 0, (const v4f *)in, (v4f *)out, n16); });
        timeit("copy  16 B lanes, nt load + plain store", 2.0 * bytes, [&] { hipLaunchKernelGGL((k_copy<v4f, true, false, 4>), dim3(g), dim3(256), 0, 0, (const v4f *)in, (v4f *)out, n16); });
        timeit("copy   8 B lanes, nt load + nt store, 16 deep", 2.0 * bytes, [&] { hipLaunchKernelGGL((k_copy<v2f, true, true, 16>), dim3(g), dim3(256), 0, 0, (const v2f *)in, (v2f *)out, n8); });
        {
            const size_t tiles = n8 / (16 * 256);           // 16 loads per lane: the decimator's 4096-sample tile
            timeit("mix 8:1, 16 x 8 B loads, one 16 B store per 2 lanes", 1.125 * bytes, [&] { hipLaunchKernelGGL((k_mix81<16>), dim3(g), dim3(256), 0, 0, (const v2f *)in, (v2f *)out, tiles); });
        }
    }
    return 0;
}
