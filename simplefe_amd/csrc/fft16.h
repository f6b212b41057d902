// fft16.h -- register-resident 16-point complex DFT (radix-4 x radix-4) on packed cf32.
// Host+device so tests/host/test_fft16.cpp can check it against a naive DFT without a GPU.
#pragma once
#include "common.h"

namespace sfe {

__host__ __device__ __forceinline__ v2f cmul(v2f a, v2f w)
{
    // (a.x w.x - a.y w.y, a.y w.x + a.x w.y) as one packed mul + one packed fma
    v2f t = a * (v2f){w.x, w.x};
    return __builtin_elementwise_fma((v2f){a.y, a.x}, (v2f){-w.y, w.y}, t);
}
__host__ __device__ __forceinline__ v2f cmul_conj(v2f a, v2f w)
{
    v2f t = a * (v2f){w.x, w.x};
    return __builtin_elementwise_fma((v2f){a.y, a.x}, (v2f){w.y, -w.y}, t);
}

template <int DIR>
__host__ __device__ __forceinline__ void dft4(v2f &a0, v2f &a1, v2f &a2, v2f &a3)
{
    v2f s0 = a0 + a2, d0 = a0 - a2, s1 = a1 + a3, d1 = a1 - a3;
    v2f r = DIR < 0 ? (v2f){d1.y, -d1.x} : (v2f){-d1.y, d1.x};   // -+ j * d1
    a0 = s0 + s1;
    a2 = s0 - s1;
    a1 = d0 + r;
    a3 = d0 - r;
}

// multiply by W_16^(DIR*m), constants folded at compile time
template <int DIR, int M>
__host__ __device__ __forceinline__ v2f tw16(v2f a)
{
    constexpr float C1 = 0.92387953251128673848f, S1 = 0.38268343236508978178f;
    constexpr float R = 0.70710678118654752440f;
    constexpr float sg = DIR < 0 ? -1.0f : 1.0f;     // sign of the imaginary part
    if constexpr (M == 0) return a;
    else if constexpr (M == 4) return DIR < 0 ? (v2f){a.y, -a.x} : (v2f){-a.y, a.x};
    else if constexpr (M == 2) return (v2f){R, R} * (DIR < 0 ? (v2f){a.x + a.y, a.y - a.x}
                                                              : (v2f){a.x - a.y, a.y + a.x});
    else if constexpr (M == 6) return (v2f){R, R} * (DIR < 0 ? (v2f){a.y - a.x, -a.x - a.y}
                                                              : (v2f){-a.x - a.y, a.x - a.y});
    else if constexpr (M == 1) return cmul(a, (v2f){C1, sg * S1});
    else if constexpr (M == 3) return cmul(a, (v2f){S1, sg * C1});
    else /* M == 9 */ return cmul(a, (v2f){-C1, -sg * S1});
}

// 16-point DFT in registers.  Result element k is left in v[P16(k)].
__host__ __device__ constexpr int P16(int k) { return (k >> 2) | ((k & 3) << 2); }

template <int DIR>
__host__ __device__ __forceinline__ void dft16(v2f (&v)[16])
{
#pragma unroll
    for (int a = 0; a < 4; a++) dft4<DIR>(v[a], v[a + 4], v[a + 8], v[a + 12]);
    // v[a + 4b] *= W16^(a*b)
    v[1 + 4] = tw16<DIR, 1>(v[1 + 4]);
    v[1 + 8] = tw16<DIR, 2>(v[1 + 8]);
    v[1 + 12] = tw16<DIR, 3>(v[1 + 12]);
    v[2 + 4] = tw16<DIR, 2>(v[2 + 4]);
    v[2 + 8] = tw16<DIR, 4>(v[2 + 8]);
    v[2 + 12] = tw16<DIR, 6>(v[2 + 12]);
    v[3 + 4] = tw16<DIR, 3>(v[3 + 4]);
    v[3 + 8] = tw16<DIR, 6>(v[3 + 8]);
    v[3 + 12] = tw16<DIR, 9>(v[3 + 12]);
#pragma unroll
    for (int b = 0; b < 4; b++) dft4<DIR>(v[4 * b], v[4 * b + 1], v[4 * b + 2], v[4 * b + 3]);
}

}  // namespace sfe
