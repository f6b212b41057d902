// util.hip -- synthetic stream generator and the gr-simplefe wire-format converters (gfx950).
#include "common.h"

#pragma clang fp contract(off)   // converters are bit-exact restatements: no FMA contraction

namespace sfe {
namespace {

// simplefe_amd/synth.py:hash32 is the host twin of this function -- keep them identical.
__device__ __forceinline__ uint32_t hash32(uint32_t seed, uint32_t ch, uint64_t idx)
{
    uint32_t x = (uint32_t)idx * 0x9E3779B9u + (uint32_t)(idx >> 32) * 0x7F4A7C15u +
                 seed * 0x85EBCA6Bu + ch * 0xC2B2AE35u + 0x165667B1u;
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}

__device__ __forceinline__ float synth_val(uint32_t seed, uint32_t ch, uint64_t idx)
{
    return (float)((int32_t)hash32(seed, ch, idx) >> 8) * 1.1920928955078125e-07f;   // 2^-23
}

__global__ __launch_bounds__(256) void synth_fill_kernel(float *d, uint64_t n, uint32_t seed,
                                                         uint32_t ch, uint64_t first)
{
    // 4 floats per lane, 16-byte stores; grid-stride
    const uint64_t n4 = n >> 2;
    for (uint64_t q = (uint64_t)blockIdx.x * 256 + threadIdx.x; q < n4; q += (uint64_t)gridDim.x * 256) {
        const uint64_t i = q << 2;
        v4f v = {synth_val(seed, ch, first + i), synth_val(seed, ch, first + i + 1),
                 synth_val(seed, ch, first + i + 2), synth_val(seed, ch, first + i + 3)};
        reinterpret_cast<v4f *>(d)[q] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const uint64_t i = (n4 << 2) + threadIdx.x;
        d[i] = synth_val(seed, ch, first + i);
    }
}

// RX: (b - 128) * (1/127)   gr-simplefe/lib/source_c_impl.cc:121-132, source_f_impl.cc:120-129
__global__ __launch_bounds__(256) void rx_u8_kernel(const uint8_t *src, float *dst, size_t n)
{
    const float qinv = 1.0f / 127.0f;
    const size_t n4 = n >> 2;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n4; q += (size_t)gridDim.x * 256) {
        const uint32_t w = reinterpret_cast<const uint32_t *>(src)[q];
        v4f v = {(float)((int)(w & 0xFF) - 128) * qinv, (float)((int)((w >> 8) & 0xFF) - 128) * qinv,
                 (float)((int)((w >> 16) & 0xFF) - 128) * qinv, (float)((int)(w >> 24) - 128) * qinv};
        reinterpret_cast<v4f *>(dst)[q] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t i = (n4 << 2) + threadIdx.x;
        dst[i] = (float)((int)src[i] - 128) * qinv;
    }
}

// TX: ((short)(x*511) + 512) & 0x3FF, 4 samples -> 5 bytes
//     gr-simplefe/lib/sink_f_impl.cc:117-143 (== sink_c_impl.cc:118-144, bpsk.cxx:76-101)
__device__ __forceinline__ uint32_t q10(float x)
{
    const int v = (int)(short)(int)(x * 511.0f);   // C float->short: truncate toward zero
    return (uint32_t)(v + 512) & 0x3FFu;
}

// One thread packs FOUR groups: 16 floats in (four 16-byte loads), 20 bytes out as five dwords
// (20 k is 4-byte aligned whenever dst is); the last n_groups % 4 groups go out bytewise.
__global__ __launch_bounds__(256) void tx_10bit_kernel(const float *src, uint8_t *dst, size_t n_groups)
{
    const size_t n_quads = n_groups / 4;
    const bool dst_aligned = (reinterpret_cast<uintptr_t>(dst) & 3) == 0;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n_quads; q += (size_t)gridDim.x * 256) {
        uint8_t b[20];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const v4f x = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(src) + 4 * q + k);
            const uint32_t u0 = q10(x.x), u1 = q10(x.y), u2 = q10(x.z), u3 = q10(x.w);
            b[5 * k + 0] = (uint8_t)((u0 >> 8) | ((u1 >> 8) << 2) | ((u2 >> 8) << 4) | ((u3 >> 8) << 6));
            b[5 * k + 1] = (uint8_t)(u0 & 0xFF);
            b[5 * k + 2] = (uint8_t)(u1 & 0xFF);
            b[5 * k + 3] = (uint8_t)(u2 & 0xFF);
            b[5 * k + 4] = (uint8_t)(u3 & 0xFF);
        }
        uint8_t *o = dst + q * 20;
        if (dst_aligned) {
#pragma unroll
            for (int w = 0; w < 5; w++)
                reinterpret_cast<uint32_t *>(o)[w] = (uint32_t)b[4 * w] | ((uint32_t)b[4 * w + 1] << 8) | ((uint32_t)b[4 * w + 2] << 16) |
                                                     ((uint32_t)b[4 * w + 3] << 24);
        } else {
#pragma unroll
            for (int i = 0; i < 20; i++) o[i] = b[i];
        }
    }
    // tail groups (fewer than four), one thread each
    const size_t g = n_quads * 4 + (size_t)blockIdx.x * 256 + threadIdx.x;
    if (g < n_groups) {
        const v4f x = reinterpret_cast<const v4f *>(src)[g];
        const uint32_t u0 = q10(x.x), u1 = q10(x.y), u2 = q10(x.z), u3 = q10(x.w);
        uint8_t *o = dst + g * 5;
        o[0] = (uint8_t)((u0 >> 8) | ((u1 >> 8) << 2) | ((u2 >> 8) << 4) | ((u3 >> 8) << 6));
        o[1] = (uint8_t)(u0 & 0xFF);
        o[2] = (uint8_t)(u1 & 0xFF);
        o[3] = (uint8_t)(u2 & 0xFF);
        o[4] = (uint8_t)(u3 & 0xFF);
    }
}

inline unsigned grid_for(size_t items)
{
    size_t b = (items + 255) / 256;
    if (b < 1) b = 1;
    if (b > 256 * 8) b = 256 * 8;
    return (unsigned)b;
}

}  // namespace

int launch_synth_fill(float *d, uint64_t n, uint32_t seed, uint32_t ch, uint64_t first, hipStream_t s)
{
    if (n == 0) return SFE_OK;
    if (reinterpret_cast<uintptr_t>(d) & 15) {
        set_error("synth_fill: destination must be 16-byte aligned");
        return SFE_EINVAL;
    }
    hipLaunchKernelGGL(synth_fill_kernel, dim3(grid_for(n / 4)), dim3(256), 0, s, d, n, seed, ch, first);
    SFE_HIP(hipGetLastError());
    return SFE_OK;
}

int launch_rx_u8_to_f32(const uint8_t *src, float *dst, size_t n, hipStream_t s)
{
    if (n == 0) return SFE_OK;
    if ((reinterpret_cast<uintptr_t>(src) & 3) || (reinterpret_cast<uintptr_t>(dst) & 15)) {
        set_error("rx_u8_to_f32: src must be 4-byte and dst 16-byte aligned");
        return SFE_EINVAL;
    }
    hipLaunchKernelGGL(rx_u8_kernel, dim3(grid_for(n / 4)), dim3(256), 0, s, src, dst, n);
    SFE_HIP(hipGetLastError());
    return SFE_OK;
}

int launch_tx_f32_to_10bit(const float *src, uint8_t *dst, size_t n_floats, hipStream_t s)
{
    const size_t groups = n_floats / 4;   // the reference consumes whole groups of 4 (i += 4)
    if (groups == 0) return SFE_OK;
    if (reinterpret_cast<uintptr_t>(src) & 15) {
        set_error("tx_f32_to_10bit: src must be 16-byte aligned");
        return SFE_EINVAL;
    }
    hipLaunchKernelGGL(tx_10bit_kernel, dim3(grid_for(groups / 4 + 1)), dim3(256), 0, s, src, dst, groups);
    SFE_HIP(hipGetLastError());
    return SFE_OK;
}

}  // namespace sfe
