// fft16.h -- register-resident 16-point complex DFT (radix-4 x radix-4) on packed cf32.
//
// On the device every complex primitive is ONE or TWO VOP3P instructions: gfx950's
// v_pk_{add,mul,fma}_f32 take per-lane operand swizzles (op_sel / op_sel_hi) and sign flips
// (neg_lo / neg_hi), which is exactly what a complex multiply or a +-j rotation needs.  hipcc
// does not find those forms from C (it adds v_xor / v_mov / an extra v_pk_add per primitive:
// 3-4 instructions per complex multiply, 10 per radix-4 butterfly), so they are written as
// inline asm; the host build (tests/host/test_fft16.cpp checks this header against a naive
// DFT without a GPU) uses the plain C forms.
#pragma once
#include "common.h"

namespace sfe {

// The W16 constants of the DFT16 are operands of packed instructions, which take no 64-bit literal: register pairs.  In a
// kernel that loops over transforms the compiler materialises them once, outside the loop -- up to eight VGPR pairs that
// live for the whole kernel.  A file that defines SFE_W16_SCALAR before including this header has them in SGPR pairs
// instead (one scalar source per packed instruction is allowed); poly_gen.hip does, where sixteen VGPRs decide the occupancy.
#ifdef SFE_W16_SCALAR
#define SFE_W16_C "s"
#else
#define SFE_W16_C "v"
#endif
#if defined(__HIP_DEVICE_COMPILE__)
// a * w
// (each primitive is ONE asm statement: hipcc pads an s_nop after every asm statement whose
// result the next instruction reads, so two statements per complex multiply cost two pads)
__device__ __forceinline__ v2f cmul(v2f a, v2f w)
{
    v2f t, r;
    asm("v_pk_mul_f32 %1, %2, %3 op_sel_hi:[1,0]\n\t"                                        // t = (ax wx, ay wx)
        "v_pk_fma_f32 %0, %2, %3, %1 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"        // (-ay wy, ax wy) + t
        : "=v"(r), "=&v"(t) : "v"(a), "v"(w));
    return r;
}
// a * w, w one of the W16 constants
__device__ __forceinline__ v2f cmul_k(v2f a, v2f w)
{
    v2f t, r;
    asm("v_pk_mul_f32 %1, %2, %3 op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %2, %3, %1 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"
        : "=v"(r), "=&v"(t) : "v"(a), SFE_W16_C(w));
    return r;
}
// acc + a * w
__device__ __forceinline__ v2f cmac(v2f acc, v2f a, v2f w)
{
    v2f t, r;
    asm("v_pk_fma_f32 %1, %3, %4, %2 op_sel_hi:[1,0,1]\n\t"                                 // t = (ax wx, ay wx) + acc
        "v_pk_fma_f32 %0, %3, %4, %1 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"        // (-ay wy, ax wy) + t
        : "=v"(r), "=&v"(t) : "v"(acc), "v"(a), "v"(w));
    return r;
}
// a * conj(w)
__device__ __forceinline__ v2f cmul_conj(v2f a, v2f w)
{
    v2f t, r;
    asm("v_pk_mul_f32 %1, %2, %3 op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %2, %3, %1 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]"        // (ay wy, -ax wy) + t
        : "=v"(r), "=&v"(t) : "v"(a), "v"(w));
    return r;
}
// (a * q) * p and (a * conj q) * conj p: the factored twiddles, four instructions, one statement
__device__ __forceinline__ v2f cmul2(v2f a, v2f q, v2f p)
{
    v2f t, u, r;
    asm("v_pk_mul_f32 %1, %3, %4 op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %2, %3, %4, %1 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n\t"
        "v_pk_mul_f32 %1, %2, %5 op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %2, %5, %1 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"
        : "=v"(r), "=&v"(t), "=&v"(u) : "v"(a), "v"(q), "v"(p));
    return r;
}
__device__ __forceinline__ v2f cmul2_conj(v2f a, v2f q, v2f p)
{
    v2f t, u, r;
    asm("v_pk_mul_f32 %1, %3, %4 op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %2, %3, %4, %1 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]\n\t"
        "v_pk_mul_f32 %1, %2, %5 op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %2, %5, %1 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]"
        : "=v"(r), "=&v"(t), "=&v"(u) : "v"(a), "v"(q), "v"(p));
    return r;
}
// s (a -+ j a), then optionally rotated by -+j: the W16^2 / W16^6 multiplies in two instructions
__device__ __forceinline__ v2f w16_2_fwd(v2f a, v2f s)    // s (ax+ay, ay-ax)
{
    v2f t, r;
    asm("v_pk_add_f32 %1, %2, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"
        "v_pk_mul_f32 %0, %1, %3 op_sel_hi:[1,0]" : "=v"(r), "=&v"(t) : "v"(a), SFE_W16_C(s));
    return r;
}
__device__ __forceinline__ v2f w16_2_inv(v2f a, v2f s)    // s (ax-ay, ay+ax)
{
    v2f t, r;
    asm("v_pk_add_f32 %1, %2, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"
        "v_pk_mul_f32 %0, %1, %3 op_sel_hi:[1,0]" : "=v"(r), "=&v"(t) : "v"(a), SFE_W16_C(s));
    return r;
}
__device__ __forceinline__ v2f w16_6_fwd(v2f a, v2f s)    // -j * s (ax+ay, ay-ax)
{
    v2f t, r;
    asm("v_pk_add_f32 %1, %2, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"
        "v_pk_mul_f32 %0, %1, %3 op_sel:[1,0] op_sel_hi:[0,0] neg_hi:[1,0]" : "=v"(r), "=&v"(t) : "v"(a), SFE_W16_C(s));
    return r;
}
__device__ __forceinline__ v2f w16_6_inv(v2f a, v2f s)    // +j * s (ax-ay, ay+ax)
{
    v2f t, r;
    asm("v_pk_add_f32 %1, %2, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"
        "v_pk_mul_f32 %0, %1, %3 op_sel:[1,0] op_sel_hi:[0,0] neg_lo:[1,0]" : "=v"(r), "=&v"(t) : "v"(a), SFE_W16_C(s));
    return r;
}
// a - j b = (ax + by, ay - bx)
__device__ __forceinline__ v2f add_mj(v2f a, v2f b)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a + j b = (ax - by, ay + bx)
__device__ __forceinline__ v2f add_pj(v2f a, v2f b)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// -j a = (ay, -ax)          j a = (-ay, ax)
__device__ __forceinline__ v2f rot_mj(v2f a)
{
    v2f r;
    asm("v_pk_add_f32 %0, 0, %1 op_sel:[0,1] op_sel_hi:[0,0] neg_hi:[0,1]" : "=v"(r) : "v"(a));
    return r;
}
__device__ __forceinline__ v2f rot_pj(v2f a)
{
    v2f r;
    asm("v_pk_add_f32 %0, 0, %1 op_sel:[0,1] op_sel_hi:[0,0] neg_lo:[0,1]" : "=v"(r) : "v"(a));
    return r;
}
// s * a (real scale; only the low half of s is read), optionally followed by -j / +j
__device__ __forceinline__ v2f scale(v2f a, v2f s)
{
    v2f r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(s));
    return r;
}
__device__ __forceinline__ v2f scale_mj(v2f a, v2f s)     // s * (ay, -ax)
{
    v2f r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,0] neg_hi:[1,0]" : "=v"(r) : "v"(a), "v"(s));
    return r;
}
__device__ __forceinline__ v2f scale_pj(v2f a, v2f s)     // s * (-ay, ax)
{
    v2f r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,0] neg_lo:[1,0]" : "=v"(r) : "v"(a), "v"(s));
    return r;
}
#else
__host__ __device__ __forceinline__ v2f cmul(v2f a, v2f w) { return (v2f){a.x * w.x - a.y * w.y, a.y * w.x + a.x * w.y}; }
__host__ __device__ __forceinline__ v2f cmul_conj(v2f a, v2f w) { return (v2f){a.x * w.x + a.y * w.y, a.y * w.x - a.x * w.y}; }
__host__ __device__ __forceinline__ v2f cmul_k(v2f a, v2f w) { return cmul(a, w); }
__host__ __device__ __forceinline__ v2f cmac(v2f acc, v2f a, v2f w) { return (v2f){acc.x + a.x * w.x - a.y * w.y, acc.y + a.y * w.x + a.x * w.y}; }
__host__ __device__ __forceinline__ v2f add_mj(v2f a, v2f b) { return (v2f){a.x + b.y, a.y - b.x}; }
__host__ __device__ __forceinline__ v2f add_pj(v2f a, v2f b) { return (v2f){a.x - b.y, a.y + b.x}; }
__host__ __device__ __forceinline__ v2f rot_mj(v2f a) { return (v2f){a.y, -a.x}; }
__host__ __device__ __forceinline__ v2f rot_pj(v2f a) { return (v2f){-a.y, a.x}; }
__host__ __device__ __forceinline__ v2f scale(v2f a, v2f s) { return (v2f){a.x * s.x, a.y * s.x}; }
__host__ __device__ __forceinline__ v2f scale_mj(v2f a, v2f s) { return (v2f){a.y * s.x, -a.x * s.x}; }
__host__ __device__ __forceinline__ v2f scale_pj(v2f a, v2f s) { return (v2f){-a.y * s.x, a.x * s.x}; }
__host__ __device__ __forceinline__ v2f cmul2(v2f a, v2f q, v2f p) { return cmul(cmul(a, q), p); }
__host__ __device__ __forceinline__ v2f cmul2_conj(v2f a, v2f q, v2f p) { return cmul_conj(cmul_conj(a, q), p); }
__host__ __device__ __forceinline__ v2f w16_2_fwd(v2f a, v2f s) { return scale(add_mj(a, a), s); }
__host__ __device__ __forceinline__ v2f w16_2_inv(v2f a, v2f s) { return scale(add_pj(a, a), s); }
__host__ __device__ __forceinline__ v2f w16_6_fwd(v2f a, v2f s) { return scale_mj(add_mj(a, a), s); }
__host__ __device__ __forceinline__ v2f w16_6_inv(v2f a, v2f s) { return scale_pj(add_pj(a, a), s); }
#endif

// radix-4 butterfly, DIR = -1 forward (W4 = -j), +1 inverse: 8 packed adds.  (Written as one
// in-place asm statement it costs MORE: the "+v" ties make hipcc copy registers around it.)
template <int DIR>
__host__ __device__ __forceinline__ void dft4(v2f &a0, v2f &a1, v2f &a2, v2f &a3)
{
    const v2f s0 = a0 + a2, d0 = a0 - a2, s1 = a1 + a3, d1 = a1 - a3;
    a0 = s0 + s1;
    a2 = s0 - s1;
    a1 = DIR < 0 ? add_mj(d0, d1) : add_pj(d0, d1);
    a3 = DIR < 0 ? add_pj(d0, d1) : add_mj(d0, d1);
}

// multiply by W_16^(DIR*m), m in {0,1,2,3,4,6,9}: 0, 1 or 2 instructions
template <int DIR, int M>
__host__ __device__ __forceinline__ v2f tw16(v2f a)
{
    constexpr float C1 = 0.92387953251128673848f, S1 = 0.38268343236508978178f;
    constexpr float R = 0.70710678118654752440f;
    constexpr float sg = DIR < 0 ? -1.0f : 1.0f;     // sign of the imaginary part
    if constexpr (M == 0) return a;
    else if constexpr (M == 4) return DIR < 0 ? rot_mj(a) : rot_pj(a);
    else if constexpr (M == 2)   // R(1 -+ j) a = R (a -+ j a)
        return DIR < 0 ? w16_2_fwd(a, (v2f){R, R}) : w16_2_inv(a, (v2f){R, R});
    else if constexpr (M == 6)   // W^6 = -+j W^2
        return DIR < 0 ? w16_6_fwd(a, (v2f){R, R}) : w16_6_inv(a, (v2f){R, R});
    else if constexpr (M == 1) return cmul_k(a, (v2f){C1, sg * S1});
    else if constexpr (M == 3) return cmul_k(a, (v2f){S1, sg * C1});
    else /* M == 9 */ return cmul_k(a, (v2f){-C1, -sg * S1});
}

// 16-point DFT in registers.  Result element k is left in v[P16(k)].
__host__ __device__ constexpr int P16(int k) { return (k >> 2) | ((k & 3) << 2); }

// dft16 = dft16_head, then the four butterflies dft16_tail<DIR>(v, b), b = 0..3: butterfly b completes
// the result elements b, b + 4, b + 8, b + 12 (in v[4b], v[4b + 1], v[4b + 2], v[4b + 3]), so a caller can
// put each element to use -- store it -- while the remaining butterflies still run (fir_fft.hip).
template <int DIR>
__host__ __device__ __forceinline__ void dft16_head(v2f (&v)[16])
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
}
template <int DIR>
__host__ __device__ __forceinline__ void dft16_tail(v2f (&v)[16], int b)
{
    dft4<DIR>(v[4 * b], v[4 * b + 1], v[4 * b + 2], v[4 * b + 3]);
}
template <int DIR>
__host__ __device__ __forceinline__ void dft16(v2f (&v)[16])
{
    dft16_head<DIR>(v);
#pragma unroll
    for (int b = 0; b < 4; b++) dft16_tail<DIR>(v, b);
}

// The transposed schedule: takes input element n in v[P16(n)] (i.e. exactly what dft16 leaves
// behind) and leaves result element k in v[k].  dft16 followed by dft16_rev therefore needs no
// register shuffling in between -- used for the spectrum-multiply + first inverse stage.
template <int DIR>
__host__ __device__ __forceinline__ void dft16_rev(v2f (&v)[16])
{
#pragma unroll
    for (int b = 0; b < 4; b++) dft4<DIR>(v[4 * b], v[4 * b + 1], v[4 * b + 2], v[4 * b + 3]);
    // v[c + 4b] *= W16^(b*c)
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
    for (int c = 0; c < 4; c++) dft4<DIR>(v[c], v[c + 4], v[c + 8], v[c + 12]);
}

}  // namespace sfe
