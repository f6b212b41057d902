/* win_stubs.c -- the 29 KERNEL32 + 31 msvcrt functions libfftw3f-3.dll imports, over libc/pthread.
 *
 * TEST INFRASTRUCTURE ONLY (see peload.c).  Every function here is operating-system or C-runtime
 * plumbing: locks, handles, clocks, stdio, the heap, string and memory moves, a sort.  NOTHING HERE
 * COMPUTES IN FLOATING POINT (tests/test_pe_loader.py disassembles the object and checks): all
 * the arithmetic of the transforms -- twiddle generation included, the DLL carries MinGW's own
 * sin/cos -- executes inside the DLL's code.
 *
 * The DLL calls these with the Microsoft x64 convention, hence ms_abi on every one. */
#define _GNU_SOURCE
#include "peload.h"

#include <errno.h>
#include <pthread.h>
#include <semaphore.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

#define MS __attribute__((ms_abi))
typedef uint32_t DWORD;
typedef int32_t BOOL;
typedef void *HANDLE;

static void unsupported(const char *what)
{
    fprintf(stderr, "oracle/pe: the DLL called %s, which this checker does not provide\n", what);
    abort();
}

/* ---- KERNEL32: handles (mutexes, semaphores) --------------------------------------------------*/
enum { H_MUTEX = 0x4d, H_SEM = 0x53 };
struct handle { int kind; pthread_mutex_t m; sem_t s; };

static void recursive_mutex_init(pthread_mutex_t *m)
{
    pthread_mutexattr_t a;
    pthread_mutexattr_init(&a);
    pthread_mutexattr_settype(&a, PTHREAD_MUTEX_RECURSIVE);
    pthread_mutex_init(m, &a);
    pthread_mutexattr_destroy(&a);
}

static MS HANDLE k_CreateMutexA(void *attr, BOOL owned, const char *name)
{
    (void)attr; (void)name;
    struct handle *h = calloc(1, sizeof *h);
    if (!h) return NULL;
    h->kind = H_MUTEX;
    recursive_mutex_init(&h->m);
    if (owned) pthread_mutex_lock(&h->m);
    return h;
}
static MS HANDLE k_CreateSemaphoreA(void *attr, int32_t initial, int32_t maximum, const char *name)
{
    (void)attr; (void)maximum; (void)name;
    struct handle *h = calloc(1, sizeof *h);
    if (!h) return NULL;
    h->kind = H_SEM;
    sem_init(&h->s, 0, (unsigned)initial);
    return h;
}
static MS BOOL k_CloseHandle(HANDLE p)
{
    struct handle *h = p;
    if (!h || (intptr_t)p == -1) return 1;
    if (h->kind == H_MUTEX) pthread_mutex_destroy(&h->m);
    else if (h->kind == H_SEM) sem_destroy(&h->s);
    else return 0;
    h->kind = 0;
    free(h);
    return 1;
}
static MS BOOL k_ReleaseMutex(HANDLE p)
{
    struct handle *h = p;
    return h && h->kind == H_MUTEX && pthread_mutex_unlock(&h->m) == 0;
}
static MS BOOL k_ReleaseSemaphore(HANDLE p, int32_t count, int32_t *prev)
{
    struct handle *h = p;
    if (!h || h->kind != H_SEM) return 0;
    if (prev) { int v = 0; sem_getvalue(&h->s, &v); *prev = v; }
    while (count-- > 0) sem_post(&h->s);
    return 1;
}
static MS DWORD k_WaitForSingleObject(HANDLE p, DWORD ms)
{
    struct handle *h = p;
    (void)ms;                                   /* the DLL waits with INFINITE only */
    if (!h) return 0xffffffffu;
    if (h->kind == H_MUTEX) return pthread_mutex_lock(&h->m) == 0 ? 0 : 0xffffffffu;
    if (h->kind == H_SEM) { while (sem_wait(&h->s) != 0 && errno == EINTR) {} return 0; }
    return 0xffffffffu;
}

/* ---- KERNEL32: critical sections (40-byte caller-owned structs; ours lives behind slot 0) ------*/
static MS void k_InitializeCriticalSection(void **cs)
{
    pthread_mutex_t *m = malloc(sizeof *m);
    recursive_mutex_init(m);
    cs[0] = m;
}
static MS void k_DeleteCriticalSection(void **cs)
{
    if (cs[0]) { pthread_mutex_destroy(cs[0]); free(cs[0]); cs[0] = NULL; }
}
static MS void k_EnterCriticalSection(void **cs) { pthread_mutex_lock(cs[0]); }
static MS void k_LeaveCriticalSection(void **cs) { pthread_mutex_unlock(cs[0]); }

/* ---- KERNEL32: identity, clocks -----------------------------------------------------------------*/
static MS HANDLE k_GetCurrentProcess(void) { return (HANDLE)(intptr_t)-1; }
static MS DWORD k_GetCurrentProcessId(void) { return (DWORD)getpid(); }
static MS DWORD k_GetCurrentThreadId(void) { return (DWORD)syscall(SYS_gettid); }
static MS DWORD k_GetLastError(void) { return 0; }
static MS void k_GetSystemTimeAsFileTime(uint64_t *ft)
{
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    /* 100 ns ticks since 1601-01-01 */
    uint64_t v = ((uint64_t)ts.tv_sec + 11644473600ull) * 10000000ull + (uint64_t)ts.tv_nsec / 100u;
    memcpy(ft, &v, 8);
}
static uint64_t mono_ns(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}
static MS DWORD k_GetTickCount(void) { return (DWORD)(mono_ns() / 1000000u); }
static MS BOOL k_QueryPerformanceCounter(int64_t *v) { *v = (int64_t)mono_ns(); return 1; }
static MS DWORD k_GetTimeZoneInformation(void *tzi) { memset(tzi, 0, 172); return 0; }
static MS void k_Sleep(DWORD ms) { usleep((useconds_t)ms * 1000u); }
static MS BOOL k_TerminateProcess(HANDLE h, uint32_t code) { (void)h; _exit((int)code); }

/* ---- KERNEL32: structured exception handling, virtual memory, TLS --------------------------------
 * No exception is ever raised through the DLL here, so the unwind tables are accepted and ignored. */
static MS BOOL k_RtlAddFunctionTable(void *t, DWORD n, uint64_t base) { (void)t; (void)n; (void)base; return 1; }
static MS void k_RtlCaptureContext(void *ctx) { (void)ctx; }
static MS void *k_RtlLookupFunctionEntry(uint64_t pc, uint64_t *base, void *hist) { (void)pc; (void)hist; if (base) *base = 0; return NULL; }
static MS void *k_RtlVirtualUnwind(DWORD t, uint64_t b, uint64_t pc, void *fe, void *ctx, void **hd, uint64_t *ef, void *cp)
{
    (void)t; (void)b; (void)pc; (void)fe; (void)ctx; (void)hd; (void)ef; (void)cp;
    return NULL;
}
static MS void *k_SetUnhandledExceptionFilter(void *f) { (void)f; return NULL; }
static MS int32_t k_UnhandledExceptionFilter(void *info) { (void)info; return 0; }
static MS void *k_TlsGetValue(DWORD idx) { (void)idx; return NULL; }
static MS BOOL k_VirtualProtect(void *addr, size_t size, DWORD prot, DWORD *old)
{
    (void)addr; (void)size; (void)prot;         /* the image is mapped read-write-execute as a whole */
    if (old) *old = 0x40;
    return 1;
}
static MS size_t k_VirtualQuery(const void *addr, void *info, size_t len)
{
    /* MEMORY_BASIC_INFORMATION (48 bytes): BaseAddress, AllocationBase, AllocationProtect,
     * RegionSize, State, Protect, Type -- asked only about the image's own pages. */
    void *ibase; unsigned long isize;
    uint64_t out[6] = {0, 0, 0, 0, 0, 0};
    if (len < 48) return 0;
    pe_image_range(&ibase, &isize);
    uint64_t a = (uint64_t)addr & ~0xfffull, b = (uint64_t)ibase;
    if (a < b || a >= b + isize) return 0;
    out[0] = a;
    out[1] = b;
    out[2] = 0x40;                              /* AllocationProtect = PAGE_EXECUTE_READWRITE */
    out[3] = b + isize - a;                     /* RegionSize */
    out[4] = 0x1000u | ((uint64_t)0x40 << 32);  /* State = MEM_COMMIT, Protect */
    out[5] = 0x1000000u;                        /* Type = MEM_IMAGE */
    memcpy(info, out, 48);
    return 48;
}

/* ---- msvcrt: start-up and exit tables ------------------------------------------------------------*/
typedef void(MS * voidfn)(void);
static MS void *c___dllonexit(void *fn, void ***begin, void ***end) { (void)begin; (void)end; return fn; }   /* never unloaded */
static MS void *c__onexit(void *fn) { return fn; }
static MS void c___setusermatherr(void *fn) { (void)fn; }
static MS void c__amsg_exit(int code) { fprintf(stderr, "oracle/pe: runtime error %d in the DLL's start-up code\n", code); abort(); }
static MS void c__initterm(voidfn *a, voidfn *b) { for (; a < b; a++) if (*a) (*a)(); }
static pthread_mutex_t g_crt_lock;
static pthread_once_t g_crt_lock_once = PTHREAD_ONCE_INIT;
static void crt_lock_init(void) { recursive_mutex_init(&g_crt_lock); }
static MS void c__lock(int n) { (void)n; pthread_once(&g_crt_lock_once, crt_lock_init); pthread_mutex_lock(&g_crt_lock); }
static MS void c__unlock(int n) { (void)n; pthread_mutex_unlock(&g_crt_lock); }
static MS uintptr_t c__beginthreadex(void *sec, unsigned stack, void *fn, void *arg, unsigned flags, unsigned *tid)
{
    (void)sec; (void)stack; (void)fn; (void)arg; (void)flags; (void)tid;
    unsupported("_beginthreadex (FFTW's threaded planner; the reference never asks for it)");
    return 0;
}
static MS void c__endthreadex(unsigned code) { (void)code; unsupported("_endthreadex"); }
static int g_errno;
static MS int *c__errno(void) { return &g_errno; }
static MS void c_abort(void) { abort(); }
static MS void *c_signal(int sig, void *handler) { (void)sig; (void)handler; return NULL; }

/* ---- msvcrt: heap, memory, strings -----------------------------------------------------------------*/
static MS void *c_malloc(size_t n) { return malloc(n); }
static MS void *c_calloc(size_t n, size_t m) { return calloc(n, m); }
static MS void c_free(void *p) { free(p); }
static MS void *c_memcpy(void *d, const void *s, size_t n) { return memcpy(d, s, n); }
static MS void *c_memmove(void *d, const void *s, size_t n) { return memmove(d, s, n); }
static MS void *c_memset(void *d, int c, size_t n) { return memset(d, c, n); }
static MS int c_strcmp(const char *a, const char *b) { return strcmp(a, b); }
static MS int c_strncmp(const char *a, const char *b, size_t n) { return strncmp(a, b, n); }
static MS size_t c_strlen(const char *s) { return strlen(s); }

/* qsort: the comparison callback lives in the DLL (ms_abi), so libc's qsort cannot take it.
 * Insertion sort -- FFTW sorts tensor dimensions, a handful of elements. */
static MS void c_qsort(void *base, size_t n, size_t size, int(MS *cmp)(const void *, const void *))
{
    char *a = base, *tmp = malloc(size ? size : 1);
    for (size_t i = 1; tmp && i < n; i++) {
        size_t j = i;
        memcpy(tmp, a + i * size, size);
        while (j > 0 && cmp(a + (j - 1) * size, tmp) > 0) { memcpy(a + j * size, a + (j - 1) * size, size); j--; }
        memcpy(a + j * size, tmp, size);
    }
    free(tmp);
}

/* ---- msvcrt: stdio -----------------------------------------------------------------------------------
 * __iob_func() returns msvcrt's {stdin, stdout, stderr} array, 48 bytes per FILE on Win64; the DLL
 * forms &iob[1], &iob[2] itself.  Anything else is a FILE* our own fopen handed out. */
static unsigned char g_iob[3 * 48];
static MS void *c___iob_func(void) { return g_iob; }
static FILE *to_file(void *f)
{
    unsigned char *p = f;
    if (p >= g_iob && p < g_iob + sizeof g_iob) {
        size_t i = (size_t)(p - g_iob) / 48;
        return i == 0 ? stdin : i == 1 ? stdout : stderr;
    }
    return f;
}
static MS void *c_fopen(const char *path, const char *mode) { return fopen(path, mode); }
static MS int c_fclose(void *f) { return fclose(to_file(f)); }
static MS int c_ferror(void *f) { return ferror(to_file(f)); }
static MS int c_fflush(void *f) { return f ? fflush(to_file(f)) : fflush(NULL); }
static MS size_t c_fread(void *p, size_t s, size_t n, void *f) { return fread(p, s, n, to_file(f)); }
static MS size_t c_fwrite(const void *p, size_t s, size_t n, void *f) { return fwrite(p, s, n, to_file(f)); }

/* A Microsoft va_list is a pointer to consecutive 8-byte argument slots.  A System V va_list whose
 * register areas are marked exhausted reads its arguments from `overflow_arg_area` the same way,
 * 8 bytes each (integers, pointers and doubles alike), so libc can format the DLL's diagnostics. */
static int ms_vfprintf(FILE *f, const char *fmt, char *ms_ap)
{
    va_list ap;
    ap[0].gp_offset = 48;
    ap[0].fp_offset = 304;
    ap[0].overflow_arg_area = ms_ap;
    ap[0].reg_save_area = NULL;
    return vfprintf(f, fmt, ap);
}
static MS int c_vfprintf(void *f, const char *fmt, char *ms_ap) { return ms_vfprintf(to_file(f), fmt, ms_ap); }
static MS int c_fprintf(void *f, const char *fmt, ...)
{
    __builtin_ms_va_list ap;
    __builtin_ms_va_start(ap, fmt);
    int r = ms_vfprintf(to_file(f), fmt, (char *)ap);
    __builtin_ms_va_end(ap);
    return r;
}

/* ---- the table ---------------------------------------------------------------------------------------*/
#define K(n) { "KERNEL32.dll", #n, (void *)k_##n }
#define C(n) { "msvcrt.dll", #n, (void *)c_##n }
static const struct { const char *dll, *name; void *fn; } g_stubs[] = {
    K(CloseHandle), K(CreateMutexA), K(CreateSemaphoreA), K(DeleteCriticalSection), K(EnterCriticalSection),
    K(GetCurrentProcess), K(GetCurrentProcessId), K(GetCurrentThreadId), K(GetLastError),
    K(GetSystemTimeAsFileTime), K(GetTickCount), K(GetTimeZoneInformation), K(InitializeCriticalSection),
    K(LeaveCriticalSection), K(QueryPerformanceCounter), K(ReleaseMutex), K(ReleaseSemaphore),
    K(RtlAddFunctionTable), K(RtlCaptureContext), K(RtlLookupFunctionEntry), K(RtlVirtualUnwind),
    K(SetUnhandledExceptionFilter), K(Sleep), K(TerminateProcess), K(TlsGetValue), K(UnhandledExceptionFilter),
    K(VirtualProtect), K(VirtualQuery), K(WaitForSingleObject),
    C(__dllonexit), C(__iob_func), C(__setusermatherr), C(_amsg_exit), C(_beginthreadex), C(_endthreadex),
    C(_errno), C(_initterm), C(_lock), C(_onexit), C(_unlock), C(abort), C(calloc), C(fclose), C(ferror),
    C(fflush), C(fopen), C(fprintf), C(fread), C(free), C(fwrite), C(malloc), C(memcpy), C(memmove),
    C(memset), C(qsort), C(signal), C(strcmp), C(strlen), C(strncmp), C(vfprintf),
};

void *win_stub_resolve(const char *dll, const char *name)
{
    for (size_t i = 0; i < sizeof g_stubs / sizeof g_stubs[0]; i++)
        if (strcasecmp(dll, g_stubs[i].dll) == 0 && strcmp(name, g_stubs[i].name) == 0) return g_stubs[i].fn;
    return NULL;
}
