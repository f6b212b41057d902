/* peload.c -- map a Win64 DLL into this Linux process and hand out its exports.
 *
 * TEST INFRASTRUCTURE ONLY (oracle/).  Nothing under simplefe_amd/, include/ or bench.py's
 * timed region links, loads or calls this.  It exists for ONE purpose: the reference vendors the
 * real FFTW 3.3.5 single-precision library only as a Win64 binary
 * (/root/reference/contrib/fftw-3.3.5-dll64/libfftw3f-3.dll -- the library
 * libdsp/CMakeLists.txt:8-10 links on Win32, called at libdsp/blkconv.cxx:41-73,89,103), the
 * DLL is x86-64 code and so is this container, so the UNMODIFIED blkconv.cxx can call the
 * reference's own FFTW here, on the CPU, and golden vectors can be taken from it
 * (tests/golden/make_golden_fftw.py).  The DLL is read where it lies; no byte of it enters the
 * repository.  Build container only: /root/reference does not exist on the GPU box.
 *
 * What a loader has to do for this DLL (objdump -p): map 11 sections, apply the DIR64 base
 * relocations, bind 29 KERNEL32 + 31 msvcrt imports (win_stubs.c -- locks, time, stdio, malloc;
 * no arithmetic), give the thread a TEB the MinGW start-up code can look at (%gs:0x30 is read
 * once, in _CRT_INIT, for the thread's stack base as a lock-owner id), run the TLS callbacks and
 * the entry point with DLL_PROCESS_ATTACH, resolve exports by name.
 *
 * All integer work: this file and win_stubs.c contain no floating-point code. */
#define _GNU_SOURCE
#include "peload.h"

#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <unistd.h>

#define MS __attribute__((ms_abi))

static uint16_t rd16(const uint8_t *p) { uint16_t v; memcpy(&v, p, 2); return v; }
static uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint64_t rd64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }

struct pe_image {
    uint8_t *base;          /* where the image is mapped */
    uint64_t size;          /* SizeOfImage */
    uint64_t pref_base;     /* ImageBase in the file */
    uint32_t dir_rva[16], dir_size[16];
    uint32_t entry_rva;
    char err[256];
};

static char g_err[256];
const char *pe_last_error(void) { return g_err; }

#define FAIL(...) do { snprintf(g_err, sizeof g_err, __VA_ARGS__); goto fail; } while (0)

/* ---- the thread's fake TEB ------------------------------------------------------------------
 * x86-64 Linux leaves the user-mode GS base unused (glibc's TLS is on FS), Win64 code finds its
 * Thread Environment Block there.  Fields the MinGW runtime may read: +0x08 StackBase,
 * +0x10 StackLimit, +0x30 Self, +0x40/+0x48 ClientId, +0x58 ThreadLocalStoragePointer. */
#ifndef ARCH_SET_GS
#define ARCH_SET_GS 0x1001
#endif
static __thread uint64_t t_teb[0x400];          /* 8 KiB, zeroed */
static __thread void *t_tls_slots[64];
static __thread int t_teb_ready;
static void *g_tls_template;                     /* the image's .tls raw data */
static size_t g_tls_template_size, g_tls_block_size;

void pe_enter_thread(void)
{
    if (t_teb_ready) return;
    uint64_t sp = (uint64_t)__builtin_frame_address(0);
    t_teb[0x08 / 8] = (sp + 0xffff) & ~0xffffull;            /* StackBase: unique per thread */
    t_teb[0x10 / 8] = t_teb[0x08 / 8] - (8u << 20);
    t_teb[0x30 / 8] = (uint64_t)t_teb;
    t_teb[0x40 / 8] = (uint64_t)getpid();
    t_teb[0x48 / 8] = (uint64_t)syscall(SYS_gettid);
    t_teb[0x58 / 8] = (uint64_t)t_tls_slots;
    if (g_tls_block_size) {
        void *blk = calloc(1, g_tls_block_size);
        if (blk && g_tls_template_size) memcpy(blk, g_tls_template, g_tls_template_size);
        t_tls_slots[0] = blk;                                  /* the image's _tls_index is 0 */
    }
    syscall(SYS_arch_prctl, ARCH_SET_GS, (unsigned long)t_teb);
    t_teb_ready = 1;
}

/* ---- mapping ---------------------------------------------------------------------------------*/
static uint8_t *g_img_base;
static uint64_t g_img_size;
void pe_image_range(void **base, unsigned long *size) { *base = g_img_base; *size = g_img_size; }

/* ---- SHA-256 of the file about to be mapped (an opaque binary from the reference tree is executed in this process; it
 * is mapped only if it is byte for byte the library the golden vectors were made with, and only when the caller opted in).  FIPS 180-4,
 * integer arithmetic only. */
static const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
    0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
    0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
    0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
    0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
    0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
static uint32_t ror32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
static void sha256_block(uint32_t h[8], const uint8_t *b)
{
    uint32_t w[64], a[8];
    for (int i = 0; i < 16; i++) w[i] = (uint32_t)b[4 * i] << 24 | (uint32_t)b[4 * i + 1] << 16 | (uint32_t)b[4 * i + 2] << 8 | b[4 * i + 3];
    for (int i = 16; i < 64; i++)
        w[i] = w[i - 16] + (ror32(w[i - 15], 7) ^ ror32(w[i - 15], 18) ^ (w[i - 15] >> 3)) + w[i - 7] +
               (ror32(w[i - 2], 17) ^ ror32(w[i - 2], 19) ^ (w[i - 2] >> 10));
    memcpy(a, h, sizeof a);
    for (int i = 0; i < 64; i++) {
        uint32_t t1 = a[7] + (ror32(a[4], 6) ^ ror32(a[4], 11) ^ ror32(a[4], 25)) + ((a[4] & a[5]) ^ (~a[4] & a[6])) + K256[i] + w[i];
        uint32_t t2 = (ror32(a[0], 2) ^ ror32(a[0], 13) ^ ror32(a[0], 22)) + ((a[0] & a[1]) ^ (a[0] & a[2]) ^ (a[1] & a[2]));
        memmove(a + 1, a, 7 * sizeof a[0]);
        a[4] += t1;
        a[0] = t1 + t2;
    }
    for (int i = 0; i < 8; i++) h[i] += a[i];
}
static void sha256_hex(const uint8_t *p, uint64_t n, char out[65])
{
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    uint64_t i = 0;
    for (; i + 64 <= n; i += 64) sha256_block(h, p + i);
    uint8_t tail[128] = {0};
    uint64_t r = n - i, bits = n * 8;
    memcpy(tail, p + i, r);
    tail[r] = 0x80;
    unsigned len = r + 9 <= 64 ? 64 : 128;
    for (int k = 0; k < 8; k++) tail[len - 1 - k] = (uint8_t)(bits >> (8 * k));
    sha256_block(h, tail);
    if (len == 128) sha256_block(h, tail + 64);
    for (int k = 0; k < 8; k++) snprintf(out + 8 * k, 9, "%08x", h[k]);
}
/* FFTW 3.3.5 single precision, the Win64 build the reference vendors (contrib/fftw-3.3.5-dll64/libfftw3f-3.dll) */
#define SFE_PINNED_DLL_SHA256 "42ca18fff35dd12890e04478bc990005b3969cb744f6843976bd436ccd7f0a4c"

pe_image *pe_load(const char *path, pe_resolver resolve)
{
    pe_image *im = NULL;
    uint8_t *file = MAP_FAILED;
    struct stat st;
    /* explicit opt-in: nothing maps and runs the binary unless the caller's environment says so */
    const char *optin = getenv("SFE_ORACLE_RUN_FFTW_DLL");
    if (!optin || strcmp(optin, "1") != 0) {
        snprintf(g_err, sizeof g_err, "refusing to map %s: set SFE_ORACLE_RUN_FFTW_DLL=1 to opt in (it executes a binary from the reference tree)", path);
        return NULL;
    }
    int fd = open(path, O_RDONLY);
    if (fd < 0) { snprintf(g_err, sizeof g_err, "cannot open %s", path); return NULL; }
    if (fstat(fd, &st) != 0 || st.st_size < 0x200) FAIL("%s: too small for a PE image", path);
    file = mmap(NULL, st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (file == MAP_FAILED) FAIL("mmap of %s failed", path);
    {
        char hex[65];
        sha256_hex(file, (uint64_t)st.st_size, hex);
        if (strcmp(hex, SFE_PINNED_DLL_SHA256) != 0) FAIL("%s: sha256 %s is not the pinned library (%s)", path, hex, SFE_PINNED_DLL_SHA256);
    }

    if (rd16(file) != 0x5a4d) FAIL("%s: no MZ header", path);
    uint32_t nt = rd32(file + 0x3c);
    if ((uint64_t)nt + 0x108 > (uint64_t)st.st_size || rd32(file + nt) != 0x00004550) FAIL("%s: no PE signature", path);
    const uint8_t *fh = file + nt + 4, *oh = fh + 20;
    if (rd16(fh) != 0x8664) FAIL("%s: machine 0x%x is not x86-64", path, rd16(fh));
    if (rd16(oh) != 0x20b) FAIL("%s: not PE32+", path);
    unsigned n_sec = rd16(fh + 2), opt_size = rd16(fh + 16);

    im = calloc(1, sizeof *im);
    if (!im) FAIL("out of memory");
    im->entry_rva = rd32(oh + 16);
    im->pref_base = rd64(oh + 24);
    im->size = rd32(oh + 56);
    uint32_t hdr_size = rd32(oh + 60);
    unsigned n_dirs = rd32(oh + 108);
    for (unsigned i = 0; i < 16 && i < n_dirs; i++) {
        im->dir_rva[i] = rd32(oh + 112 + 8 * i);
        im->dir_size[i] = rd32(oh + 116 + 8 * i);
    }

    /* one RWX region for the whole image: protections per section buy nothing in a checker */
    im->base = mmap(NULL, im->size, PROT_READ | PROT_WRITE | PROT_EXEC, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (im->base == MAP_FAILED) { im->base = NULL; FAIL("cannot map %lu bytes for the image", (unsigned long)im->size); }
    memcpy(im->base, file, hdr_size);           /* the runtime walks its own headers (__ImageBase) */
    const uint8_t *sh = oh + opt_size;
    for (unsigned s = 0; s < n_sec; s++, sh += 40) {
        uint32_t vsize = rd32(sh + 8), va = rd32(sh + 12), rsize = rd32(sh + 16), rptr = rd32(sh + 20);
        uint32_t n = rsize < vsize || vsize == 0 ? rsize : vsize;
        if ((uint64_t)va + n > im->size || (uint64_t)rptr + n > (uint64_t)st.st_size) FAIL("section %u out of range", s);
        memcpy(im->base + va, file + rptr, n);                  /* the rest of the section stays zero */
    }

    /* base relocations: every DIR64 slot holds a preferred-base VA */
    uint64_t delta = (uint64_t)im->base - im->pref_base;
    for (uint32_t off = 0; off + 8 <= im->dir_size[5];) {
        const uint8_t *blk = im->base + im->dir_rva[5] + off;
        uint32_t page = rd32(blk), bsz = rd32(blk + 4);
        if (bsz < 8) break;
        for (uint32_t e = 8; e + 2 <= bsz; e += 2) {
            uint16_t ent = rd16(blk + e);
            unsigned type = ent >> 12;
            if (type == 0) continue;
            if (type != 10) FAIL("relocation type %u not handled", type);
            uint8_t *slot = im->base + page + (ent & 0xfff);
            uint64_t v = rd64(slot) + delta;
            memcpy(slot, &v, 8);
        }
        off += bsz;
    }

    /* imports */
    for (const uint8_t *d = im->base + im->dir_rva[1]; im->dir_rva[1] && rd32(d + 12); d += 20) {
        const char *dll = (const char *)im->base + rd32(d + 12);
        uint32_t ilt = rd32(d) ? rd32(d) : rd32(d + 16), iat = rd32(d + 16);
        for (unsigned i = 0;; i++) {
            uint64_t ent = rd64(im->base + ilt + 8 * i);
            if (!ent) break;
            if (ent >> 63) FAIL("%s: import by ordinal %lu not handled", dll, (unsigned long)(ent & 0xffff));
            const char *name = (const char *)im->base + (uint32_t)ent + 2;
            void *fn = resolve(dll, name);
            if (!fn) FAIL("no stub for %s!%s", dll, name);
            memcpy(im->base + iat + 8 * i, &fn, 8);
        }
    }

    /* TLS directory (VAs, already relocated) */
    if (im->dir_rva[9]) {
        const uint8_t *t = im->base + im->dir_rva[9];
        uint64_t start = rd64(t), end = rd64(t + 8), pidx = rd64(t + 16);
        uint32_t zero_fill = rd32(t + 32);
        g_tls_template = (void *)start;
        g_tls_template_size = end - start;
        g_tls_block_size = g_tls_template_size + zero_fill + 64;
        uint32_t idx0 = 0;
        memcpy((void *)pidx, &idx0, 4);
    }
    g_img_base = im->base;
    g_img_size = im->size;
    munmap(file, st.st_size);
    close(fd);
    return im;
fail:
    if (file != MAP_FAILED) munmap(file, st.st_size);
    close(fd);
    if (im) { if (im->base) munmap(im->base, im->size); free(im); }
    return NULL;
}

int pe_run_entry(pe_image *im)
{
    pe_enter_thread();
    typedef void(MS * tls_cb_t)(void *, uint32_t, void *);
    typedef int(MS * entry_t)(void *, uint32_t, void *);
    if (im->dir_rva[9]) {
        uint64_t cbs = rd64(im->base + im->dir_rva[9] + 24);
        for (tls_cb_t *cb = (tls_cb_t *)cbs; cbs && *cb; cb++) (*cb)(im->base, 1 /* DLL_PROCESS_ATTACH */, NULL);
    }
    if (!im->entry_rva) return 1;
    int ok = ((entry_t)(im->base + im->entry_rva))(im->base, 1, NULL);
    if (!ok) snprintf(g_err, sizeof g_err, "the DLL's entry point returned FALSE");
    return ok;
}

void *pe_export(pe_image *im, const char *name)
{
    if (!im->dir_rva[0]) return NULL;
    const uint8_t *e = im->base + im->dir_rva[0];
    uint32_t n_names = rd32(e + 24), funcs = rd32(e + 28), names = rd32(e + 32), ords = rd32(e + 36);
    for (uint32_t i = 0; i < n_names; i++) {
        const char *nm = (const char *)im->base + rd32(im->base + names + 4 * i);
        if (strcmp(nm, name) != 0) continue;
        uint32_t rva = rd32(im->base + funcs + 4 * rd16(im->base + ords + 2 * i));
        if (rva >= im->dir_rva[0] && rva < im->dir_rva[0] + im->dir_size[0]) return NULL;   /* forwarder */
        return im->base + rva;
    }
    return NULL;
}
