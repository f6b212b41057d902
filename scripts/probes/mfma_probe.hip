// Probe: issue rate of v_mfma_f32_16x16x4_f32 with (a) operands in registers, (b) the B operand
// read from LDS per MFMA and A per 4 MFMAs (the poly_mfma_kernel K loop), at 4 workgroups per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256, 4) void k(float *out, int iters, int gs2)
{
    extern __shared__ float lds[];
    const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (unsigned i = tid; i < 9000; i += 256) lds[i] = (float)(i & 15) * 0.001f;
    __syncthreads();
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    const unsigned j = lane & 15, kq = lane >> 4;
    const float *bp = lds + gs2 * (wave * 32 + (j >> 1) * 2) + (j & 1) + 2 * (151 - kq);
    const float *ap = lds + 7000 + lane;
    float av = 1.0f + lane, b0 = 0.5f, b1 = 0.25f, b2 = 0.125f, b3 = 2.0f;
    for (int it = 0; it < iters; it++) {
        const float *bq = bp;
        for (int ks = 0; ks < 38; ks++) {
            if (MODE == 1) {
                av = ap[(ks & 7) * 64];
                b0 = bq[0]; b1 = bq[gs2]; b2 = bq[16 * gs2]; b3 = bq[17 * gs2];
                bq -= 8;
            }
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b2, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b3, a3, 0, 0, 0);
        }
    }
    out[blockIdx.x * 256 + tid] = a0[0] + a1[1] + a2[2] + a3[3];
}
int main()
{
    float *d;
    (void)hipMalloc(&d, 1024 * 256 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 200;
    for (int mode = 0; mode < 2; mode++)
        for (int rep = 0; rep < 2; rep++) {
            (void)hipEventRecord(e0, 0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1024), dim3(256), 36352, 0, d, iters, 50);
            else hipLaunchKernelGGL(k<1>, dim3(1024), dim3(256), 36352, 0, d, iters, 50);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            const double mf = 1024.0 * 4 * iters * 38 * 4;      // MFMAs
            printf("mode %d: %.3f ms, %.1f MFMA-cycles(32)/SIMD busy-equivalent: %.3f ms at 2.1 GHz; %.1f TFLOP/s\n", mode, ms,
                   0.0, mf / 1024.0 * 32.0 / 2.1e6, mf * 2048.0 / ms / 1e9);
        }
    return 0;
}
