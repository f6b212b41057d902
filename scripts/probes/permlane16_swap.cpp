// What v_permlane16_swap_b32 does on gfx950, lane by lane (round 5: the real-stream FIR pairs lanes with it).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/pl16 scripts/probes/permlane16_swap.cpp && /tmp/pl16
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *p)
{
    const unsigned l = threadIdx.x;
    const auto r = __builtin_amdgcn_permlane16_swap(100u + l, 200u + l, false, false);      // first operand 100 + lane, second 200 + lane
    p[l] = r[0];
    p[64 + l] = r[1];
}
int main()
{
    unsigned *d, h[128];
    if (hipMalloc(&d, sizeof h) != hipSuccess) return 1;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    if (hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    for (int half = 0; half < 2; half++) {
        printf("result[%d]:", half);
        for (int l = 0; l < 64; l += 8) printf(" lane %2d: %u", l, h[64 * half + l]);
        printf("\n");
    }
    return 0;
}
