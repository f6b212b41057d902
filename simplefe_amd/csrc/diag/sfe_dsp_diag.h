/* sfe_dsp_diag.h -- entry points of the DIAGNOSTIC library only (libsfe_dsp_diag.so, simplefe_amd/build.py build_lib(diag=True));
 * the product library exports none of them and include/sfe_dsp.h does not declare them.
 *
 * Round 5: sfe_dsp_malloc_pair and friends were product exports in round 4 and were demoted (VERDICT r4 weak 3 / missing 4,
 * DESIGN.md 9): a built pair removed the worst case of the allocation lottery but never reached what the best plain pair
 * gives, cost 5-7 s per call, and its first version lost a caller's data on a fresh mapping for a reason that was worked
 * around (settle_mapping) rather than explained.  They stay here for the measurements scripts/ make with them. */
#ifndef SFE_DSP_DIAG_H
#define SFE_DSP_DIAG_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
/* A PAIR of device buffers for a stream call that reads one while it writes the other.  What such a pair gives is fixed
 * when the memory is handed out ("the two modes"): physical memory comes in classes, in stretches of tens of
 * GiB; a read stream and a write stream from the same class run ~8 % slower together (8 : 1 mix; 10-18 % at 1 : 1) than
 * streams from different classes, each alone the same in both -- a property of the platform's memory, seen by any
 * streaming kernel, the bare copy included.  A large hipMalloc is stitched from whatever is free: a plain pair is a lottery.
 * probe_pair: the median time of five launches of the bare mix over (d_in, d_out) -- every 32 KiB read, the output
 *   written in proportion -- on the null stream, synchronous; the output's contents are overwritten.
 * malloc_pair: for streams of a GiB and more the pair is BUILT: a pool of 1 GiB physical chunks (the HIP virtual-memory
 *   calls; 16 x tries beyond what the buffers need, released afterwards; a few seconds) is classified with the bare mix, the
 *   input mapped from one class and the output from another into two contiguous ranges, and the result verified against an
 *   output of the input's own class.  Such memory is ordinary device memory to kernels, to every sfe_dsp_* call and to
 *   hipMemcpy / hipMemset (checked across chunk boundaries); it is sized in whole GiB.  Smaller streams, and processes where
 *   the build is not possible or does not verify (one class only in the pool), get malloc_pair_screened.
 * malloc_pair_screened: plain hipMalloc memory: the input, then up to `tries` (1 .. 8) candidates for the output, each
 *   probed against it, the fastest kept, the others freed (all stay allocated until the choice is made).  While the
 *   candidates show no spread (within 4 %: all of one class) up to as many again are tried, each pair of them behind a 32 GiB
 *   spacer that is freed with the losers, and in the end one more allocation for the input.  tries = 1: two plain allocations.
 * ms_kept / ms_worst (may be NULL): the probe time of the pair returned, and of the slowest candidate (built pairs: of the
 *   same input with an output of its own class).  Free both buffers with sfe_dsp_free. */
int sfe_dsp_probe_pair(const void *d_in, size_t in_bytes, void *d_out, size_t out_bytes, float *ms);
int sfe_dsp_malloc_pair(size_t in_bytes, size_t out_bytes, int tries, void **d_in, void **d_out, float *ms_kept, float *ms_worst);
int sfe_dsp_malloc_pair_screened(size_t in_bytes, size_t out_bytes, int tries, void **d_in, void **d_out, float *ms_kept, float *ms_worst);
/* kind = 1: the address lies in a range malloc_pair mapped from chunks; 0: anything else (plain allocations included) */
int sfe_dsp_mem_kind(const void *dptr, int *kind);
int sfe_dsp_diag_last_pool(void **va, size_t *n_chunks);
#ifdef __cplusplus
}
#endif
#endif
