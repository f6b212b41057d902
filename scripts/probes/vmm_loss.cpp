// vmm_loss.cpp -- VERDICT r4 missing 4: when does a freshly mapped range lose what is written to it?
// Round 4's sfe_dsp_malloc_pair (now diag/alloc.hip) saw a kernel launched straight after hipMemMap + hipMemSetAccess find most of
// the range "not backed": stores dropped, loads zero, no fault; settle_mapping retried until two passes read back clean.  This probe
// walks the sequences one by one: 8 chunks of 1 GiB (hipMemCreate), a fill kernel, a check kernel, the fraction of 16-byte words lost
// at once and again 20 ms later.
//   A  fresh: reserve, map, set access, fill, check
//   B  remap at once: (A), unmap, free the range, reserve again (the driver hands back the same address), map the SAME chunks, set access, fill, check
//   C  B with hipDeviceSynchronize between the unmap and the new reservation
//   D  B, but the chunks stay mapped at a SECOND range of their own while the first is remapped (the classification range of malloc_pair)
//   E  B with the access set BEFORE anything else touches the device (hipMemSetAccess, hipDeviceSynchronize, then fill)
//   G  A, but the chunks rest for 1.5 s between hipMemCreate and the first write (is it the driver's own clearing of new memory, still under way?)
//   H  A, with the check repeated at 0 / 5 / 20 / 50 / 100 / 200 / 400 / 800 ms and never written again: the time course of the loss
//   F  as malloc_pair did it: the big range unmapped and freed, then TWO smaller ranges reserved and the chunks dealt over them (odd chunks / even chunks)
// build: hipcc --offload-arch=gfx950 -O2 -o scripts/probes/vmm_loss scripts/probes/vmm_loss.cpp      run: scripts/probes/vmm_loss [rounds]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
static const size_t CHUNK = (size_t)1 << 30;
__global__ void fill(uint4 *p, size_t n16, unsigned mark)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = make_uint4(mark, (unsigned)i, mark, (unsigned)(i >> 32));
}
__global__ void check(const uint4 *p, size_t n16, unsigned mark, unsigned long long *bad)
{
    unsigned long long mine = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const uint4 v = p[i];
        mine += (v.x != mark) | (v.y != (unsigned)i) | (v.z != mark);
    }
    if (mine) atomicAdd(bad, mine);
}
static hipMemAccessDesc acc_desc(int dev)
{
    hipMemAccessDesc a = {};
    a.location.type = hipMemLocationTypeDevice;
    a.location.id = dev;
    a.flags = hipMemAccessFlagsProtReadWrite;
    return a;
}
static void *map_all(const std::vector<hipMemGenericAllocationHandle_t> &h, int dev, bool set_access = true)
{
    void *va = nullptr;
    CK(hipMemAddressReserve(&va, h.size() * CHUNK, CHUNK, nullptr, 0));
    for (size_t i = 0; i < h.size(); i++) CK(hipMemMap((char *)va + i * CHUNK, CHUNK, 0, h[i], 0));
    if (set_access) { hipMemAccessDesc a = acc_desc(dev); CK(hipMemSetAccess(va, h.size() * CHUNK, &a, 1)); }
    return va;
}
static void unmap_all(void *va, size_t n)
{
    for (size_t i = 0; i < n; i++) CK(hipMemUnmap((char *)va + i * CHUNK, CHUNK));
    CK(hipMemAddressFree(va, n * CHUNK));
}
static double lost(void *va, size_t bytes, unsigned mark, unsigned long long *d_bad)
{
    CK(hipMemset(d_bad, 0, 8));
    hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, (const uint4 *)va, bytes / 16, mark, d_bad);
    unsigned long long b = 0;
    CK(hipMemcpy(&b, d_bad, 8, hipMemcpyDeviceToHost));
    return 100.0 * (double)b / (double)(bytes / 16);
}
int main(int argc, char **argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 3, dev = 0, N = 8;
    CK(hipSetDevice(dev));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    unsigned long long *d_bad;
    CK(hipMalloc(&d_bad, 8));
    for (int r = 0; r < rounds; r++) {
        for (int sc = 0; sc < 8; sc++) {
            std::vector<hipMemGenericAllocationHandle_t> h(N);
            for (auto &c : h) CK(hipMemCreate(&c, CHUNK, &prop, 0));
            const unsigned mark = 0x3f800000u + 16 * r + sc;
            if (sc == 6) usleep(1500000);                    // G
            void *va = map_all(h, dev), *second = nullptr;
            if (sc == 7) {                                   // H
                hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (uint4 *)va, N * CHUNK / 16, mark);
                printf("round %d  H  lost after", r);
                int waited = 0;
                for (int ms : {0, 5, 20, 50, 100, 200, 400, 800}) {
                    usleep((ms - waited) * 1000);
                    waited = ms;
                    printf("  %d ms %.3f %%", ms, lost(va, N * CHUNK, mark, d_bad));
                }
                hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (uint4 *)va, N * CHUNK / 16, mark + 7);
                const double again = lost(va, N * CHUNK, mark + 7, d_bad);
                usleep(200000);
                printf("   written again: %.3f %%, 200 ms later %.3f %%\n", again, lost(va, N * CHUNK, mark + 7, d_bad));
                fflush(stdout);
                CK(hipDeviceSynchronize());
                unmap_all(va, N);
                for (auto c : h) CK(hipMemRelease(c));
                continue;
            }
            void *first_va = va;
            if (sc > 0) {                                    // B..E: the range has been used, is unmapped and reserved again
                hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (uint4 *)va, N * CHUNK / 16, mark ^ 0x55u);
                if (sc == 3) second = map_all(h, dev);       // D: a second mapping of the same chunks stays
                CK(hipDeviceSynchronize());
                unmap_all(va, N);
                if (sc == 2) CK(hipDeviceSynchronize());     // C
                if (sc == 5) {                               // F: two ranges, the chunks dealt alternately
                    std::vector<hipMemGenericAllocationHandle_t> ha, hb;
                    for (int i = 0; i < N; i++) (i & 1 ? hb : ha).push_back(h[i]);
                    void *a = map_all(ha, dev), *b = map_all(hb, dev);
                    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (uint4 *)a, ha.size() * CHUNK / 16, mark);
                    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (uint4 *)b, hb.size() * CHUNK / 16, mark);
                    const double la = lost(a, ha.size() * CHUNK, mark, d_bad), lb = lost(b, hb.size() * CHUNK, mark, d_bad);
                    usleep(20000);
                    const double la2 = lost(a, ha.size() * CHUNK, mark, d_bad), lb2 = lost(b, hb.size() * CHUNK, mark, d_bad);
                    printf("round %d  F  first / second range: lost at once %7.3f %% / %7.3f %%   20 ms later %7.3f %% / %7.3f %%   (ranges at %p, %p; the big one was at %p)\n",
                           r, la, lb, la2, lb2, a, b, first_va);
                    fflush(stdout);
                    CK(hipDeviceSynchronize());
                    unmap_all(a, ha.size());
                    unmap_all(b, hb.size());
                    for (auto c : h) CK(hipMemRelease(c));
                    continue;
                }
                va = map_all(h, dev, sc != 4);
                if (sc == 4) { hipMemAccessDesc a = acc_desc(dev); CK(hipMemSetAccess(va, N * CHUNK, &a, 1)); CK(hipDeviceSynchronize()); }
            }
            hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (uint4 *)va, N * CHUNK / 16, mark);
            const double at_once = lost(va, N * CHUNK, mark, d_bad);
            usleep(20000);
            const double later = lost(va, N * CHUNK, mark, d_bad);
            hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (uint4 *)va, N * CHUNK / 16, mark + 7);      // written again: held now?
            const double rewritten = lost(va, N * CHUNK, mark + 7, d_bad);
            printf("round %d  %c  same address %-3s  lost at once %7.3f %%   20 ms later %7.3f %%   after writing again %7.3f %%\n", r, "ABCDEFGH"[sc],
                   va == first_va ? "yes" : "no", at_once, later, rewritten);
            fflush(stdout);
            CK(hipDeviceSynchronize());
            unmap_all(va, N);
            if (second) unmap_all(second, N);
            for (auto c : h) CK(hipMemRelease(c));
        }
    }
    return 0;
}
