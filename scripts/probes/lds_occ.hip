// Probe: how many 256-thread workgroups the occupancy API admits per CU vs dynamic LDS size,
// and whether a census kernel really sees them co-resident.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void k(int *count, int *maxc, int spin)
{
    extern __shared__ char s[];
    s[threadIdx.x] = 1;
    if (threadIdx.x == 0) {
        int c = atomicAdd(count, 1) + 1;
        atomicMax(maxc, c);
        long long t0 = clock64();
        while (clock64() - t0 < spin) {}
        atomicAdd(count, -1);
    }
    __syncthreads();
}
int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("sharedMemPerBlock %zu maxSharedMemoryPerMultiProcessor %zu CUs %d\n", p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor, p.multiProcessorCount);
    int *d;
    hipMalloc(&d, 8);
    for (int kb : {16, 24, 30, 32, 34, 36, 40, 48, 52, 56, 64}) {
        size_t sh = (size_t)kb * 1024;
        int nb = -1;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, 256, sh);
        hipMemset(d, 0, 8);
        hipLaunchKernelGGL(k, dim3(256 * 8), dim3(256), sh, 0, d, d + 1, 2000000);
        hipDeviceSynchronize();
        int h[2];
        hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("lds %2d KiB: API %d blocks/CU, census max co-resident %d (%.2f per CU)\n", kb, nb, h[1], h[1] / 256.0);
    }
    return 0;
}
