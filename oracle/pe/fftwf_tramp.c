/* fftwf_tramp.c -- the six libfftw3f entry points blkconv.cxx uses, forwarded into the
 * reference's own FFTW 3.3.5 binary.
 *
 * TEST INFRASTRUCTURE ONLY (see peload.c).  `nm -u blkconv.o` lists exactly
 * fftwf_malloc / fftwf_free / fftwf_plan_dft_r2c_1d / fftwf_plan_dft_c2r_1d / fftwf_execute /
 * fftwf_destroy_plan (call sites: /root/reference/libdsp/blkconv.cxx:41,44,60,68,69,72,73,89,103,
 * 116-120; contract: contrib/fftw-3.3.5-dll64/fftw3.h).  Each function here converts the System V
 * call blkconv.o makes into the Microsoft x64 call the DLL export expects -- the compiler does it,
 * the callee pointer is declared ms_abi -- and nothing else: no argument is looked at, no value
 * is computed.  The DLL is mapped on first use from SFE_FFTW_DLL or the path the reference ships
 * it at. */
#define _GNU_SOURCE
#include "peload.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>

#define MS __attribute__((ms_abi))
#ifndef SFE_FFTW_DLL_DEFAULT
#define SFE_FFTW_DLL_DEFAULT "/root/reference/contrib/fftw-3.3.5-dll64/libfftw3f-3.dll"
#endif

static void *(MS *p_malloc)(size_t);
static void(MS *p_free)(void *);
static void *(MS *p_plan_r2c)(int, float *, void *, unsigned);
static void *(MS *p_plan_c2r)(int, void *, float *, unsigned);
static void(MS *p_execute)(const void *);
static void(MS *p_destroy)(void *);
static char *p_version;                          /* the exported string fftwf_version */
static pe_image *g_im;
static pthread_once_t g_once = PTHREAD_ONCE_INIT;

static void load_once(void)
{
    const char *path = getenv("SFE_FFTW_DLL");
    if (!path || !*path) path = SFE_FFTW_DLL_DEFAULT;
    g_im = pe_load(path, win_stub_resolve);
    if (!g_im || !pe_run_entry(g_im)) {
        fprintf(stderr, "oracle/pe: %s\n", pe_last_error());
        abort();
    }
    p_malloc = pe_export(g_im, "fftwf_malloc");
    p_free = pe_export(g_im, "fftwf_free");
    p_plan_r2c = pe_export(g_im, "fftwf_plan_dft_r2c_1d");
    p_plan_c2r = pe_export(g_im, "fftwf_plan_dft_c2r_1d");
    p_execute = pe_export(g_im, "fftwf_execute");
    p_destroy = pe_export(g_im, "fftwf_destroy_plan");
    p_version = pe_export(g_im, "fftwf_version");
    if (!p_malloc || !p_free || !p_plan_r2c || !p_plan_c2r || !p_execute || !p_destroy) {
        fprintf(stderr, "oracle/pe: %s does not export the six fftwf_ functions\n", path);
        abort();
    }
}

static void enter(void)
{
    pthread_once(&g_once, load_once);
    pe_enter_thread();
}

void *fftwf_malloc(size_t n) { enter(); return p_malloc(n); }
void fftwf_free(void *p) { enter(); p_free(p); }
void *fftwf_plan_dft_r2c_1d(int n, float *in, void *out, unsigned flags) { enter(); return p_plan_r2c(n, in, out, flags); }
void *fftwf_plan_dft_c2r_1d(int n, void *in, float *out, unsigned flags) { enter(); return p_plan_c2r(n, in, out, flags); }
void fftwf_execute(const void *plan) { enter(); p_execute(plan); }
void fftwf_destroy_plan(void *plan) { enter(); p_destroy(plan); }

/* for the test that names what was loaded: "fftw-3.3.5-sse2-avx" or the like */
const char *sfe_pe_fftwf_version(void) { enter(); return p_version ? p_version : ""; }
