// fir_chunk_matrix.cpp -- the FIR's own two modes (~10 % apart, sustained, per process, not predicted by the
// bare read + write mix): WHICH physical memory makes them?
//
// One process, one pool of P physical chunks of 1 GiB (hipMemCreate), mapped back to back into one address range in the order
// they were created.  A "slot" is two consecutive chunks = 2 GiB = the headline's 2^28 cf32.  For every ordered pair of slots
// (a, b), a != b: the library's 256-tap FIR on 2^28 samples reading slot a and writing slot b (median of 5 launches after 3),
// and the bare 1 : 1 mix over the same two slots (sfe_dsp_probe_pair).  Then the same by single chunks with 2^27 samples.
// If the FIR's mode is a property of the pair like the mix's, its matrix has the mix's block structure; if it belongs to the
// input or the output alone, rows or columns; if to neither, it is not the memory.
// Modes (argv: P = chunks in the pool, then a letter; profiles/r04/fir_modes_input.txt says which block each made):
//   (none) / s  the matrix over slots of two chunks / and over single chunks      o  24 objects on two fixed pairs of slots
//   t  rests of 0-8 s, then launches in groups of 25                               l  three library pairs in a process that already holds memory
//   L  bench.py's order: the library's pair first, the object after               x  buffers from chunks that are not neighbours
//   y / z  the library's order of events by hand with random chunks (pool kept / released)
//   c  every chunk of a fresh pool as the input, one at a time                    k  physically contiguous allocations against plain ones
//   w  WHEN is fresh memory wiped after a fill: never (the zeros were a mapping not backed yet, not a late clear)
// Outcome: the "fast mode" these were written to find was the FIR reading zeros (blocks 19-22); kept as the record of the search.
//   build: hipcc -O2 scripts/probes/fir_chunk_matrix.cpp -o scripts/probes/fir_chunk_matrix -Iinclude -Lsimplefe_amd -lsfe_dsp -Wl,-rpath,$PWD/simplefe_amd
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
#include <time.h>

#include <algorithm>
#include <vector>

#include "sfe_dsp.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define SK(x) do { int r_ = (x); if (r_ != SFE_OK) { fprintf(stderr, "%s: %s\n", #x, sfe_dsp_last_error()); exit(1); } } while (0)

int main(int argc, char **argv)
{
    const size_t CHUNK = (size_t)1 << 30;
    const int P = argc > 1 ? atoi(argv[1]) : 24;
    const bool singles = argc > 2 && argv[2][0] == 's';
    if (argc > 2 && (argv[2][0] == 'y' || argv[2][0] == 'z')) {
        // the library's order of events by hand: all chunks created first, mapped as one range, ~1000 bare-mix launches over
        // it, unmapped; then buffers put together from RANDOM chunks (no classification).  y: 40 trials, the pool kept;
        // z: one trial, the rest of the pool released first (what sfe_dsp_malloc_pair does).  The object is made after the
        // first buffers exist.
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        CK(hipSetDevice(0));
        CK(hipFree(nullptr));                    // the runtime up before the first virtual-memory call
        size_t gran = 0;
        CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
        std::vector<hipMemGenericAllocationHandle_t> h(P);
        for (int i = 0; i < P; i++) CK(hipMemCreate(&h[i], CHUNK, &prop, 0));
        void *all = nullptr;
        CK(hipMemAddressReserve(&all, P * CHUNK, CHUNK, nullptr, 0));
        for (int i = 0; i < P; i++) CK(hipMemMap((char *)all + i * CHUNK, CHUNK, 0, h[i], 0));
        CK(hipMemSetAccess(all, P * CHUNK, &acc, 1));
        float dummy = 0;
        for (int i = 0; i < 130; i++) SK(sfe_dsp_probe_pair(all, CHUNK, (char *)all + (1 + i % (P - 1)) * CHUNK, CHUNK, &dummy));
        CK(hipDeviceSynchronize());
        CK(hipMemUnmap(all, P * CHUNK));
        CK(hipMemAddressFree(all, P * CHUNK));
        unsigned rng = argc > 3 ? (unsigned)atoi(argv[3]) : 12345u;
        auto next = [&]() { rng = rng * 1664525u + 1013904223u; return (int)((rng >> 8) % (unsigned)P); };
        const size_t n = (size_t)1 << 28;
        sfe_fir_t f = nullptr;
        sfe_timer_t tm;
        SK(sfe_dsp_timer_create(&tm));
        std::vector<float> taps(256);
        for (int i = 0; i < 256; i++) {
            const double k = i - 127.5, x = 0.2 * k;
            taps[i] = (float)((fabs(x) < 1e-9 ? 1.0 : sin(M_PI * x) / (M_PI * x)) * (0.54 - 0.46 * cos(2 * M_PI * i / 255.0)) * 0.2);
        }
        const int trials = argv[2][0] == 'y' ? 40 : 1;
        for (int trial = 0; trial < trials; trial++) {
            int c[4];
            for (int i = 0; i < 4; i++) {
                bool again;
                do {
                    c[i] = next();
                    again = false;
                    for (int j = 0; j < i; j++) again |= c[j] == c[i];
                } while (again);
            }
            if (argv[2][0] == 'z')
                for (int i = 0; i < P; i++)
                    if (i != c[0] && i != c[1] && i != c[2] && i != c[3]) CK(hipMemRelease(h[i]));
            void *bi = nullptr, *bo = nullptr;
            CK(hipMemAddressReserve(&bi, 2 * CHUNK, CHUNK, nullptr, 0));
            CK(hipMemAddressReserve(&bo, 2 * CHUNK, CHUNK, nullptr, 0));
            CK(hipMemMap(bi, CHUNK, 0, h[c[0]], 0));
            CK(hipMemMap((char *)bi + CHUNK, CHUNK, 0, h[c[1]], 0));
            CK(hipMemMap(bo, CHUNK, 0, h[c[2]], 0));
            CK(hipMemMap((char *)bo + CHUNK, CHUNK, 0, h[c[3]], 0));
            CK(hipMemSetAccess(bi, 2 * CHUNK, &acc, 1));
            CK(hipMemSetAccess(bo, 2 * CHUNK, &acc, 1));
            SK(sfe_dsp_synth_fill(bi, 2 * n, 20240601u, 0, 0, nullptr));
            if (!f) SK(sfe_dsp_fir_create(taps.data(), 256, 0, 1, 1, 0, 0, &f));
            for (int k = 0; k < (trial ? 10 : 40); k++) SK(sfe_dsp_fir_process_stream(f, bi, bo, n, n, n, nullptr));
            SK(sfe_dsp_timer_start(tm, nullptr));
            for (int k = 0; k < 20; k++) SK(sfe_dsp_fir_process_stream(f, bi, bo, n, n, n, nullptr));
            SK(sfe_dsp_timer_stop(tm, nullptr));
            float ms = 0, mix = 0;
            SK(sfe_dsp_timer_elapsed_ms(tm, &ms));
            SK(sfe_dsp_probe_pair(bi, 2 * CHUNK, bo, 2 * CHUNK, &mix));
            printf("%c: in (%2d,%2d) out (%2d,%2d)  FIR %.4f ms  bare mix %.4f\n", argv[2][0], c[0], c[1], c[2], c[3], ms / 20, mix);
            fflush(stdout);
            CK(hipDeviceSynchronize());
            CK(hipMemUnmap(bi, 2 * CHUNK));
            CK(hipMemUnmap(bo, 2 * CHUNK));
            CK(hipMemAddressFree(bi, 2 * CHUNK));
            CK(hipMemAddressFree(bo, 2 * CHUNK));
        }
        return 0;
    }
    if (argc > 2 && argv[2][0] == 'w') {
        // WHEN does freshly created memory stop being wiped?  P chunks created and mapped, filled at once with non-zero data
        // (sfe_dsp_synth_fill), then 16 windows of 4 KiB of every chunk read back every ~20 ms for 4 s: a window that reads all
        // zero was cleared by the driver AFTER the fill.  One line per chunk whose count of zero windows changes.
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        CK(hipSetDevice(0));
        CK(hipFree(nullptr));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        auto now = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; };
        const double t0 = now();
        std::vector<hipMemGenericAllocationHandle_t> h(P);
        for (int i = 0; i < P; i++) CK(hipMemCreate(&h[i], CHUNK, &prop, 0));
        const double t_created = now();
        void *all = nullptr;
        CK(hipMemAddressReserve(&all, P * CHUNK, CHUNK, nullptr, 0));
        for (int i = 0; i < P; i++) CK(hipMemMap((char *)all + i * CHUNK, CHUNK, 0, h[i], 0));
        CK(hipMemSetAccess(all, P * CHUNK, &acc, 1));
        const double t_mapped = now();
        SK(sfe_dsp_synth_fill(all, P * CHUNK / 4, 20240601u, 0, 0, nullptr));
        CK(hipDeviceSynchronize());
        const double t_filled = now();
        printf("%d chunks created in %.3f s, mapped %.3f s later, filled (non-zero) %.3f s after that\n", P, t_created - t0, t_mapped - t_created, t_filled - t_mapped);
        std::vector<int> zeros(P, 0);
        std::vector<unsigned char> buf(4096);
        int polls = 0;
        while (now() - t_filled < 4.0) {
            for (int i = 0; i < P; i++) {
                int z = 0;
                for (int w = 0; w < 16; w++) {
                    CK(hipMemcpy(buf.data(), (char *)all + i * CHUNK + (size_t)w * (CHUNK / 16) + 8192, 4096, hipMemcpyDeviceToHost));
                    bool any = false;
                    for (int b = 0; b < 4096 && !any; b++) any = buf[b] != 0;
                    z += any ? 0 : 1;
                }
                if (z != zeros[i]) {
                    printf("  %.3f s after the fill: chunk %2d has %2d of 16 windows zero\n", now() - t_filled, i, z);
                    zeros[i] = z;
                }
            }
            polls++;
            usleep(5000);
        }
        int wiped = 0;
        for (int i = 0; i < P; i++) wiped += zeros[i] ? 1 : 0;
        printf("%d polls; %d of %d chunks were wiped after the fill\n", polls, wiped, P);
        return 0;
    }
    if (argc > 2 && argv[2][0] == 'k') {
        // physically CONTIGUOUS buffers (hipExtMallocWithFlags, hipDeviceMallocContiguous) against plain hipMalloc, as the
        // FIR's input and output: is the fast mode what a contiguous input gives?
        CK(hipSetDevice(0));
        const size_t n = (size_t)1 << 28;
        void *ci = nullptr, *co = nullptr, *mi = nullptr, *mo = nullptr;
        const bool contiguous_first = argc > 3 && argv[3][0] == '1';
        if (contiguous_first) {
            CK(hipExtMallocWithFlags(&ci, n * 8, hipDeviceMallocContiguous));
            CK(hipExtMallocWithFlags(&co, n * 8, hipDeviceMallocContiguous));
        }
        CK(hipMalloc(&mi, n * 8));
        CK(hipMalloc(&mo, n * 8));
        if (!contiguous_first) {
            CK(hipExtMallocWithFlags(&ci, n * 8, hipDeviceMallocContiguous));
            CK(hipExtMallocWithFlags(&co, n * 8, hipDeviceMallocContiguous));
        }
        SK(sfe_dsp_synth_fill(ci, 2 * n, 20240601u, 0, 0, nullptr));
        SK(sfe_dsp_synth_fill(mi, 2 * n, 20240601u, 0, 0, nullptr));
        std::vector<float> taps(256);
        for (int i = 0; i < 256; i++) {
            const double k = i - 127.5, x = 0.2 * k;
            taps[i] = (float)((fabs(x) < 1e-9 ? 1.0 : sin(M_PI * x) / (M_PI * x)) * (0.54 - 0.46 * cos(2 * M_PI * i / 255.0)) * 0.2);
        }
        sfe_fir_t f;
        SK(sfe_dsp_fir_create(taps.data(), 256, 0, 1, 1, 0, 0, &f));
        sfe_timer_t tm;
        SK(sfe_dsp_timer_create(&tm));
        auto run = [&](const void *in, void *out) -> float {
            for (int k = 0; k < 20; k++) SK(sfe_dsp_fir_process_stream(f, in, out, n, n, n, nullptr));
            SK(sfe_dsp_timer_start(tm, nullptr));
            for (int k = 0; k < 40; k++) SK(sfe_dsp_fir_process_stream(f, in, out, n, n, n, nullptr));
            SK(sfe_dsp_timer_stop(tm, nullptr));
            float ms = 0;
            SK(sfe_dsp_timer_elapsed_ms(tm, &ms));
            return ms / 40;
        };
        for (int k = 0; k < 100; k++) SK(sfe_dsp_fir_process_stream(f, mi, mo, n, n, n, nullptr));
        float a = run(ci, co), b = run(ci, mo), c = run(mi, co), d = run(mi, mo), a2 = run(ci, co), d2 = run(mi, mo);
        float m1 = 0, m2 = 0, m3 = 0, m4 = 0;
        SK(sfe_dsp_probe_pair(ci, n * 8, co, n * 8, &m1));
        SK(sfe_dsp_probe_pair(ci, n * 8, mo, n * 8, &m2));
        SK(sfe_dsp_probe_pair(mi, n * 8, co, n * 8, &m3));
        SK(sfe_dsp_probe_pair(mi, n * 8, mo, n * 8, &m4));
        printf("contiguous %s: FIR  C->C %.4f  C->m %.4f  m->C %.4f  m->m %.4f  (again C->C %.4f  m->m %.4f)   bare mix %.4f %.4f %.4f %.4f   C in %p out %p, m in %p out %p\n",
               contiguous_first ? "first" : "after the plain ones", a, b, c, d, a2, d2, m1, m2, m3, m4, ci, co, mi, mo);
        return 0;
    }
    if (argc > 2 && argv[2][0] == 'c') {
        // every chunk of a fresh process's pool as the FIR's INPUT, one at a time (2^27 samples = 1 GiB; the output a fixed
        // distance away): are there chunks the FIR reads faster, and how many?
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        CK(hipSetDevice(0));
        CK(hipFree(nullptr));
        std::vector<hipMemGenericAllocationHandle_t> h(P);
        for (int i = 0; i < P; i++) CK(hipMemCreate(&h[i], CHUNK, &prop, 0));
        void *all = nullptr;
        CK(hipMemAddressReserve(&all, P * CHUNK, CHUNK, nullptr, 0));
        for (int i = 0; i < P; i++) CK(hipMemMap((char *)all + i * CHUNK, CHUNK, 0, h[i], 0));
        CK(hipMemSetAccess(all, P * CHUNK, &acc, 1));
        char *va = (char *)all;
        SK(sfe_dsp_synth_fill(va, P * CHUNK / 4, 20240601u, 0, 0, nullptr));
        std::vector<float> taps(256);
        for (int i = 0; i < 256; i++) {
            const double k = i - 127.5, x = 0.2 * k;
            taps[i] = (float)((fabs(x) < 1e-9 ? 1.0 : sin(M_PI * x) / (M_PI * x)) * (0.54 - 0.46 * cos(2 * M_PI * i / 255.0)) * 0.2);
        }
        sfe_fir_t f;
        SK(sfe_dsp_fir_create(taps.data(), 256, 0, 1, 1, 0, 0, &f));
        sfe_timer_t tm;
        SK(sfe_dsp_timer_create(&tm));
        const size_t n = (size_t)1 << 27;
        for (int k = 0; k < 200; k++) SK(sfe_dsp_fir_process_stream(f, va, va + (P / 2) * CHUNK, n, n, n, nullptr));
        for (int pass = 0; pass < 2; pass++) {
            printf("pass %d, input chunk 0..%d -> output chunk (i + %d) %% %d, ms:", pass, P - 1, pass ? P / 3 : P / 2, P);
            for (int i = 0; i < P; i++) {
                const void *in = va + i * CHUNK;
                void *out = va + ((i + (pass ? P / 3 : P / 2)) % P) * CHUNK;
                for (int k = 0; k < 5; k++) SK(sfe_dsp_fir_process_stream(f, in, out, n, n, n, nullptr));
                SK(sfe_dsp_timer_start(tm, nullptr));
                for (int k = 0; k < 10; k++) SK(sfe_dsp_fir_process_stream(f, in, out, n, n, n, nullptr));
                SK(sfe_dsp_timer_stop(tm, nullptr));
                float ms = 0;
                SK(sfe_dsp_timer_elapsed_ms(tm, &ms));
                printf(" %.3f", ms / 10);
            }
            printf("\n");
        }
        fflush(stdout);
        return 0;
    }
    if (argc > 2 && argv[2][0] == 'L') {
        // bench.py's order of events: the library's pair is the FIRST device memory the process asks for, the object comes after
        const size_t n = (size_t)1 << 28;
        void *in = nullptr, *out = nullptr;
        float kept = 0, worst = 0;
        SK(sfe_dsp_malloc_pair(n * 8, n * 8, 4, &in, &out, &kept, &worst));
        SK(sfe_dsp_synth_fill(in, 2 * n, 20240601u, 0, 0, nullptr));
        std::vector<float> taps(256);
        for (int i = 0; i < 256; i++) {
            const double k = i - 127.5, x = 0.2 * k;
            taps[i] = (float)((fabs(x) < 1e-9 ? 1.0 : sin(M_PI * x) / (M_PI * x)) * (0.54 - 0.46 * cos(2 * M_PI * i / 255.0)) * 0.2);
        }
        sfe_fir_t f;
        SK(sfe_dsp_fir_create(taps.data(), 256, 0, 1, 1, 0, 0, &f));
        sfe_timer_t tm;
        SK(sfe_dsp_timer_create(&tm));
        for (int k = 0; k < 40; k++) SK(sfe_dsp_fir_process_stream(f, in, out, n, n, n, nullptr));
        float ms[3];
        for (int r = 0; r < 3; r++) {
            SK(sfe_dsp_timer_start(tm, nullptr));
            for (int k = 0; k < 50; k++) SK(sfe_dsp_fir_process_stream(f, in, out, n, n, n, nullptr));
            SK(sfe_dsp_timer_stop(tm, nullptr));
            SK(sfe_dsp_timer_elapsed_ms(tm, &ms[r]));
        }
        printf("pair first, object after: FIR %.4f %.4f %.4f ms   bare mix %.4f (own class %.4f)\n", ms[0] / 50, ms[1] / 50, ms[2] / 50, kept, worst);
        return 0;
    }
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    void *res = nullptr;
    CK(hipMemAddressReserve(&res, P * CHUNK, CHUNK, nullptr, 0));
    char *va = (char *)res;
    std::vector<hipMemGenericAllocationHandle_t> h(P);
    for (int i = 0; i < P; i++) {
        CK(hipMemCreate(&h[i], CHUNK, &prop, 0));
        CK(hipMemMap(va + i * CHUNK, CHUNK, 0, h[i], 0));
    }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, P * CHUNK, &acc, 1));
    SK(sfe_dsp_synth_fill(va, P * CHUNK / 4, 20240601u, 0, 0, nullptr));
    CK(hipDeviceSynchronize());

    std::vector<float> taps(256);
    for (int i = 0; i < 256; i++) {
        const double k = i - 127.5, x = 0.2 * k;
        taps[i] = (float)((fabs(x) < 1e-9 ? 1.0 : sin(M_PI * x) / (M_PI * x)) * (0.54 - 0.46 * cos(2 * M_PI * i / 255.0)) * 0.2);
    }
    sfe_fir_t f;
    SK(sfe_dsp_fir_create(taps.data(), 256, 0, 1, 1, 0, 0, &f));
    sfe_timer_t tm;
    SK(sfe_dsp_timer_create(&tm));
    auto fir_ms = [&](const void *in, void *out, size_t n) -> float {
        for (int i = 0; i < 3; i++) SK(sfe_dsp_fir_process_stream(f, in, out, n, n, n, nullptr));
        float v[5];
        for (int rep = 0; rep < 5; rep++) {
            SK(sfe_dsp_timer_start(tm, nullptr));
            SK(sfe_dsp_fir_process_stream(f, in, out, n, n, n, nullptr));
            SK(sfe_dsp_timer_stop(tm, nullptr));
            SK(sfe_dsp_timer_elapsed_ms(tm, &v[rep]));
        }
        std::sort(v, v + 5);
        return v[2];
    };
    // through the start-up transient
    for (int i = 0; i < 200; i++) SK(sfe_dsp_fir_process_stream(f, va, va + 2 * CHUNK, (size_t)1 << 28, (size_t)1 << 28, (size_t)1 << 28, nullptr));
    CK(hipDeviceSynchronize());

    if (argc > 2 && argv[2][0] == 'o') {
        // the same pair of slots, many OBJECTS (each with its own spectrum, twiddle tables and ticket counters, with a few small
        // allocations of other sizes between them so that they land elsewhere): does the mode belong to the object's small buffers?
        const int NO = 24;
        if (P < 12) { fprintf(stderr, "the objects mode reads chunks 0-1 and 8-9 and writes 4-5 and 2-3: P >= 12\n"); return 1; }
        std::vector<sfe_fir_t> objs;
        std::vector<void *> pads;
        for (int i = 0; i < NO; i++) {
            sfe_fir_t g;
            SK(sfe_dsp_fir_create(taps.data(), 256, 0, 1, 1, 0, 0, &g));
            objs.push_back(g);
            void *pad = nullptr;
            CK(hipMalloc(&pad, (size_t)4096 * (1 + (i * 37) % 61)));
            pads.push_back(pad);
        }
        const size_t n = (size_t)1 << 28;
        for (int round = 0; round < 3; round++) {
            printf("# round %d: 24 objects on slots 0 -> 2, then on 4 -> 1 (back-to-back launches, 20 after 10), ms\n", round);
            for (int which = 0; which < 2; which++) {
                const void *in = va + (which ? 8 : 0) * CHUNK;
                void *out = va + (which ? 2 : 4) * CHUNK;
                for (int i = 0; i < NO; i++) {
                    for (int k = 0; k < 10; k++) SK(sfe_dsp_fir_process_stream(objs[i], in, out, n, n, n, nullptr));
                    SK(sfe_dsp_timer_start(tm, nullptr));
                    for (int k = 0; k < 20; k++) SK(sfe_dsp_fir_process_stream(objs[i], in, out, n, n, n, nullptr));
                    SK(sfe_dsp_timer_stop(tm, nullptr));
                    float ms = 0;
                    SK(sfe_dsp_timer_elapsed_ms(tm, &ms));
                    printf(" %.4f", ms / 20);
                }
                printf("\n");
            }
        }
        fflush(stdout);
        return 0;
    }
    auto fir_run = [&](const void *in, void *out, size_t n, int reps) -> float {
        for (int k = 0; k < 10; k++) SK(sfe_dsp_fir_process_stream(f, in, out, n, n, n, nullptr));
        SK(sfe_dsp_timer_start(tm, nullptr));
        for (int k = 0; k < reps; k++) SK(sfe_dsp_fir_process_stream(f, in, out, n, n, n, nullptr));
        SK(sfe_dsp_timer_stop(tm, nullptr));
        float ms = 0;
        SK(sfe_dsp_timer_elapsed_ms(tm, &ms));
        return ms / reps;
    };
    if (argc > 2 && argv[2][0] == 't') {
        // is the fast mode what the chip does after a REST?  idle for a while, then launches back to back, timed in groups of 25
        const size_t n = (size_t)1 << 28;
        const double rests[] = {0.0, 0.05, 0.2, 1.0, 3.0, 8.0, 0.0};
        for (double rest : rests) {
            CK(hipDeviceSynchronize());
            if (rest > 0) usleep((useconds_t)(rest * 1e6));
            printf("after %4.2f s idle, groups of 25 launches, ms per launch:", rest);
            for (int g = 0; g < 40; g++) {
                SK(sfe_dsp_timer_start(tm, nullptr));
                for (int k = 0; k < 25; k++) SK(sfe_dsp_fir_process_stream(f, va, va + 2 * CHUNK, n, n, n, nullptr));
                SK(sfe_dsp_timer_stop(tm, nullptr));
                float ms = 0;
                SK(sfe_dsp_timer_elapsed_ms(tm, &ms));
                printf(" %.3f", ms / 25);
            }
            printf("\n");
            fflush(stdout);
        }
        return 0;
    }
    if (argc > 2 && argv[2][0] == 'l') {
        // what bench.py does: pairs from the library (built from classified chunks), the FIR on each
        const size_t n = (size_t)1 << 28;
        for (int k = 0; k < 3; k++) {
            void *in = nullptr, *out = nullptr;
            float kept = 0, worst = 0;
            SK(sfe_dsp_malloc_pair(n * 8, n * 8, 4, &in, &out, &kept, &worst));
            SK(sfe_dsp_synth_fill(in, 2 * n, 20240601u, 0, 0, nullptr));
            const float a = fir_run(in, out, n, 20), b = fir_run(in, out, n, 20);
            printf("library pair %d: FIR %.4f %.4f ms   bare mix %.4f (own class %.4f)\n", k, a, b, kept, worst);
            fflush(stdout);
        }
        return 0;
    }
    if (argc > 2 && argv[2][0] == 'x') {
        // buffers put together from chunks that are NOT neighbours: in = (a, b), out = (c, d), mapped afresh each time
        const size_t n = (size_t)1 << 28;
        unsigned rng = argc > 3 ? (unsigned)atoi(argv[3]) : 12345u;
        auto next = [&]() { rng = rng * 1664525u + 1013904223u; return (int)((rng >> 8) % (unsigned)P); };
        CK(hipMemUnmap(va, P * CHUNK));
        for (int trial = 0; trial < 48; trial++) {
            int c[4];
            if (trial % 4 == 0) {                 // every fourth: neighbours, as the matrix had them
                c[0] = 2 * (next() % (P / 2)); c[1] = c[0] + 1;
                do { c[2] = 2 * (next() % (P / 2)); } while (c[2] == c[0]);
                c[3] = c[2] + 1;
            } else {
                for (int i = 0; i < 4; i++) {
                    bool again;
                    do {
                        c[i] = next();
                        again = false;
                        for (int j = 0; j < i; j++) again |= c[j] == c[i];
                    } while (again);
                }
            }
            void *bi = nullptr, *bo = nullptr;
            CK(hipMemAddressReserve(&bi, 2 * CHUNK, CHUNK, nullptr, 0));
            CK(hipMemAddressReserve(&bo, 2 * CHUNK, CHUNK, nullptr, 0));
            CK(hipMemMap(bi, CHUNK, 0, h[c[0]], 0));
            CK(hipMemMap((char *)bi + CHUNK, CHUNK, 0, h[c[1]], 0));
            CK(hipMemMap(bo, CHUNK, 0, h[c[2]], 0));
            CK(hipMemMap((char *)bo + CHUNK, CHUNK, 0, h[c[3]], 0));
            CK(hipMemSetAccess(bi, 2 * CHUNK, &acc, 1));
            CK(hipMemSetAccess(bo, 2 * CHUNK, &acc, 1));
            const float ms = fir_run(bi, bo, n, 10);
            float mix = 0;
            SK(sfe_dsp_probe_pair(bi, 2 * CHUNK, bo, 2 * CHUNK, &mix));
            printf("in (%2d,%2d) out (%2d,%2d)  FIR %.4f ms  bare mix %.4f%s\n", c[0], c[1], c[2], c[3], ms, mix, trial % 4 == 0 ? "   neighbours" : "");
            fflush(stdout);
            CK(hipDeviceSynchronize());
            CK(hipMemUnmap(bi, 2 * CHUNK));
            CK(hipMemUnmap(bo, 2 * CHUNK));
            CK(hipMemAddressFree(bi, 2 * CHUNK));
            CK(hipMemAddressFree(bo, 2 * CHUNK));
        }
        return 0;
    }
    for (int pass = 0; pass < (singles ? 2 : 1); pass++) {
        const int per = pass == 0 ? 2 : 1, S = P / per;
        const size_t bytes = per * CHUNK, n = bytes / 8;
        std::vector<float> tf(S * S, 0.f), tmx(S * S, 0.f);
        for (int a = 0; a < S; a++)
            for (int b = 0; b < S; b++) {
                if (a == b) continue;
                tf[a * S + b] = fir_ms(va + a * bytes, va + b * bytes, n);
                SK(sfe_dsp_probe_pair(va + a * bytes, bytes, va + b * bytes, bytes, &tmx[a * S + b]));
            }
        for (int which = 0; which < 2; which++) {
            printf("# %s, slots of %d GiB (row = slot read, column = slot written), ms\n", which == 0 ? "256-tap FIR (fir_fft4096_kernel)" : "bare 1 : 1 mix (pair_probe_kernel)", per);
            printf("     ");
            for (int b = 0; b < S; b++) printf(" %5d", b);
            printf("\n");
            for (int a = 0; a < S; a++) {
                printf("%4d:", a);
                for (int b = 0; b < S; b++) {
                    if (a == b) printf("     -");
                    else printf(" %.3f", (which == 0 ? tf : tmx)[a * S + b]);
                }
                printf("\n");
            }
        }
        // a second reading of the first row, at the end: does a pair keep its time within the process?
        printf("# row 0 of the FIR's matrix read again:");
        for (int b = 1; b < S; b++) printf(" %.3f", fir_ms(va, va + b * bytes, n));
        printf("\n");
        fflush(stdout);
    }
    sfe_dsp_fir_destroy(f);
    return 0;
}
