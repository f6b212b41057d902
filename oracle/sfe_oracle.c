/*
 * sfe_oracle.c -- CPU oracle (plain C99) for simpleFE's libdsp hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see sfe_oracle.h for the rules and the pinning status).
 * Every function names the reference file:line it restates.  Build with
 *   gcc -O2 -ffp-contract=off
 * so that the float32 operation order below is the order executed (an FMA-contracted
 * build differs from the reference in the last bit).
 */
#define _GNU_SOURCE
#include "sfe_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* =============================================================================
 * float32 FFT used by the blkconv restatement.
 *
 * The reference calls FFTW 3.3.5 single precision (libdsp/blkconv.cxx:60-73,89,103:
 * r2c forward, c2r backward unnormalised, both in place).  FFTW's source is not part
 * of the reference tree, so what is restated here is its published contract
 * (contrib/fftw-3.3.5-dll64/fftw3.h: forward sign -1, n/2+1 output bins, c2r
 * unnormalised) with an ordinary float32 radix-2 transform: same precision class
 * (error O(eps*log2 n)), not the same rounding.
 * ========================================================================== */
typedef struct {
    int    n;       /* complex length, power of two, or 0 when the slow path is used */
    float *wr, *wi; /* n/2 twiddles e^{-2 pi i k/n}, rounded from double */
    int   *rev;
} cfft_t;

static int is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

static void cfft_init(cfft_t *f, int n)
{
    f->n = n;
    f->wr = (float *)malloc(sizeof(float) * (size_t)(n / 2 + 1));
    f->wi = (float *)malloc(sizeof(float) * (size_t)(n / 2 + 1));
    f->rev = (int *)malloc(sizeof(int) * (size_t)n);
    for (int k = 0; k < n / 2; k++) {
        double a = -2.0 * M_PI * (double)k / (double)n;
        f->wr[k] = (float)cos(a);
        f->wi[k] = (float)sin(a);
    }
    int bits = 0;
    while ((1 << bits) < n) bits++;
    for (int i = 0; i < n; i++) {
        int r = 0;
        for (int b = 0; b < bits; b++)
            if (i & (1 << b)) r |= 1 << (bits - 1 - b);
        f->rev[i] = r;
    }
}

static void cfft_free(cfft_t *f)
{
    free(f->wr);
    free(f->wi);
    free(f->rev);
}

/* in-place complex FFT on separate re/im arrays; dir = -1 forward, +1 backward */
static void cfft_run(const cfft_t *f, float *re, float *im, int dir)
{
    const int n = f->n;
    for (int i = 0; i < n; i++) {
        int j = f->rev[i];
        if (j > i) {
            float t = re[i]; re[i] = re[j]; re[j] = t;
            t = im[i]; im[i] = im[j]; im[j] = t;
        }
    }
    for (int len = 2; len <= n; len <<= 1) {
        const int half = len >> 1, step = n / len;
        for (int base = 0; base < n; base += len) {
            for (int k = 0; k < half; k++) {
                const float wr = f->wr[k * step];
                const float wi = dir < 0 ? f->wi[k * step] : -f->wi[k * step];
                const int a = base + k, b = a + half;
                const float xr = re[b] * wr - im[b] * wi;
                const float xi = re[b] * wi + im[b] * wr;
                re[b] = re[a] - xr; im[b] = im[a] - xi;
                re[a] = re[a] + xr; im[a] = im[a] + xi;
            }
        }
    }
}

typedef struct {
    int    n;      /* real length */
    cfft_t half;   /* n/2-point complex FFT (n a power of two >= 4) */
    float *zr, *zi;
    float *ur, *ui; /* e^{-2 pi i k/n}, k <= n/4 .. used for the real split */
    int    slow;   /* non power-of-two: O(n^2) double-precision DFT */
} rfft_t;

static void rfft_init(rfft_t *p, int n)
{
    p->n = n;
    p->slow = !(is_pow2(n) && n >= 4);
    if (p->slow) return;
    cfft_init(&p->half, n / 2);
    p->zr = (float *)malloc(sizeof(float) * (size_t)(n / 2));
    p->zi = (float *)malloc(sizeof(float) * (size_t)(n / 2));
    p->ur = (float *)malloc(sizeof(float) * (size_t)(n / 2 + 1));
    p->ui = (float *)malloc(sizeof(float) * (size_t)(n / 2 + 1));
    for (int k = 0; k <= n / 2; k++) {
        double a = -2.0 * M_PI * (double)k / (double)n;
        p->ur[k] = (float)cos(a);
        p->ui[k] = (float)sin(a);
    }
}

static void rfft_free(rfft_t *p)
{
    if (p->slow) return;
    cfft_free(&p->half);
    free(p->zr); free(p->zi); free(p->ur); free(p->ui);
}

/* r2c, fftw3.h contract: X[k] = sum_j x[j] e^{-2 pi i jk/n}, k = 0..n/2, written as
 * interleaved (re,im) pairs.  `buf` holds n reals on entry and (n/2+1) pairs on exit
 * (in place, as blkconv.cxx:72 plans it). */
static void rfft_forward(rfft_t *p, float *buf)
{
    const int n = p->n, h = n / 2;
    if (p->slow) {
        double *x = (double *)malloc(sizeof(double) * (size_t)n);
        for (int j = 0; j < n; j++) x[j] = buf[j];
        for (int k = 0; k <= h; k++) {
            double sr = 0.0, si = 0.0;
            for (int j = 0; j < n; j++) {
                double a = -2.0 * M_PI * (double)(((long)j * k) % n) / (double)n;
                sr += x[j] * cos(a);
                si += x[j] * sin(a);
            }
            buf[2 * k] = (float)sr;
            buf[2 * k + 1] = (float)si;
        }
        free(x);
        return;
    }
    for (int j = 0; j < h; j++) { p->zr[j] = buf[2 * j]; p->zi[j] = buf[2 * j + 1]; }
    cfft_run(&p->half, p->zr, p->zi, -1);
    /* split: X[k] = E[k] + w^k O[k], E/O from Z[k], conj Z[h-k] */
    for (int k = 0; k <= h; k++) {
        const int k1 = k % h, k2 = (h - k) % h;
        const float ar = p->zr[k1], ai = p->zi[k1];
        const float br = p->zr[k2], bi = -p->zi[k2];
        const float er = 0.5f * (ar + br), ei = 0.5f * (ai + bi);
        const float dr = 0.5f * (ar - br), di = 0.5f * (ai - bi);
        /* O = -i * d */
        const float or_ = di, oi = -dr;
        const float wr = p->ur[k], wi = p->ui[k];
        buf[2 * k]     = er + (or_ * wr - oi * wi);
        buf[2 * k + 1] = ei + (or_ * wi + oi * wr);
    }
}

/* c2r, unnormalised (fftw3.h contract): x[j] = sum over the Hermitian-extended spectrum.
 * `buf` holds n/2+1 pairs on entry and n reals on exit (blkconv.cxx:73). */
static void rfft_backward(rfft_t *p, float *buf)
{
    const int n = p->n, h = n / 2;
    if (p->slow) {
        double *xr = (double *)malloc(sizeof(double) * (size_t)(h + 1));
        double *xi = (double *)malloc(sizeof(double) * (size_t)(h + 1));
        for (int k = 0; k <= h; k++) { xr[k] = buf[2 * k]; xi[k] = buf[2 * k + 1]; }
        for (int j = 0; j < n; j++) {
            double s = xr[0];
            for (int k = 1; k <= h; k++) {
                double a = 2.0 * M_PI * (double)(((long)j * k) % n) / (double)n;
                double w = (2 * k == n) ? 1.0 : 2.0;
                s += w * (xr[k] * cos(a) - xi[k] * sin(a));
            }
            buf[j] = (float)s;
        }
        free(xr); free(xi);
        return;
    }
    /* rebuild Z[k] = E[k] + i O[k] with E = (X[k]+conj X[h-k]), O = (X[k]-conj X[h-k]) w^-k
     * (factor 2 absorbed: unnormalised c2r of length n == 2 * half-length inverse) */
    for (int k = 0; k < h; k++) {
        const float ar = buf[2 * k], ai = buf[2 * k + 1];
        const float br = buf[2 * (h - k)], bi = -buf[2 * (h - k) + 1];
        const float er = ar + br, ei = ai + bi;
        const float dr = ar - br, di = ai - bi;
        const float wr = p->ur[k], wi = -p->ui[k];   /* w^-k */
        const float or_ = dr * wr - di * wi;
        const float oi = dr * wi + di * wr;
        /* Z = E + i*O */
        p->zr[k] = er - oi;
        p->zi[k] = ei + or_;
    }
    cfft_run(&p->half, p->zr, p->zi, +1);
    for (int j = 0; j < h; j++) { buf[2 * j] = p->zr[j]; buf[2 * j + 1] = p->zi[j]; }
}

/* =============================================================================
 * blkconv -- overlap-ADD FFT block convolver, real float32.
 * ========================================================================== */
struct orc_blkconv {
    rfft_t fft;
    float *spec_taps; /* (fft_len/2+1) complex  -- m_fft_taps, blkconv.cxx:41 */
    float *buf;       /* (fft_len/2+1)*2 floats -- m_data_buf, blkconv.cxx:44 */
    float *tail;      /* n_taps-1               -- m_overlap,  blkconv.cxx:52-55 */
    float  scale;     /* 1/fft_len              -- blkconv.cxx:50 */
    int    fft_len, blk, ovl;
};

/* libdsp/blkconv.cxx:34-75 */
orc_blkconv *orc_blkconv_create(const float *taps, int n_taps, int fft_len)
{
    orc_blkconv *c = (orc_blkconv *)calloc(1, sizeof(*c));
    const int bins = fft_len / 2 + 1;
    c->fft_len = fft_len;
    c->blk = fft_len + 1 - n_taps;          /* :47 */
    c->ovl = n_taps - 1;                    /* :48 */
    c->scale = 1.0f / (float)fft_len;       /* :50 */
    c->spec_taps = (float *)calloc((size_t)bins * 2, sizeof(float));
    c->buf = (float *)calloc((size_t)bins * 2, sizeof(float));
    c->tail = (float *)calloc((size_t)(c->ovl > 0 ? c->ovl : 1), sizeof(float));
    rfft_init(&c->fft, fft_len);
    /* H = rfft(zero-padded taps), :58-69 */
    for (int i = 0; i < fft_len; i++) c->buf[i] = i < n_taps ? taps[i] : 0.0f;
    rfft_forward(&c->fft, c->buf);
    memcpy(c->spec_taps, c->buf, sizeof(float) * (size_t)bins * 2);
    memset(c->buf, 0, sizeof(float) * (size_t)bins * 2);
    return c;
}

int orc_blkconv_blksize(const orc_blkconv *c) { return c->blk; }
float *orc_blkconv_buf(orc_blkconv *c) { return c->buf; }

/* libdsp/blkconv.cxx:77-110 */
void orc_blkconv_process(orc_blkconv *c)
{
    float *b = c->buf;
    const int bins = c->fft_len / 2 + 1;
    for (int i = c->blk; i < c->fft_len; i++) b[i] = 0.0f;           /* :85-87 */
    rfft_forward(&c->fft, b);                                        /* :89 */
    for (int i = 0; i < bins; i++) {                                 /* :92-101 */
        const float re = b[2 * i], im = b[2 * i + 1];
        const float cr = c->spec_taps[2 * i], ci = c->spec_taps[2 * i + 1];
        b[2 * i]     = c->scale * (re * cr - im * ci);
        b[2 * i + 1] = c->scale * (re * ci + im * cr);
    }
    rfft_backward(&c->fft, b);                                       /* :103 */
    for (int i = 0; i < c->ovl; i++) {                               /* :105-109 */
        b[i] = b[i] + c->tail[i];
        c->tail[i] = b[c->blk + i];
    }
}

void orc_blkconv_destroy(orc_blkconv *c)
{
    if (!c) return;
    rfft_free(&c->fft);
    free(c->spec_taps); free(c->buf); free(c->tail);
    free(c);
}

/* caller pattern of libdsp/test/test_blkconv.cxx:9-31 and examples/bpsk/bpsk.cxx:126-164:
 * fill [0,blk), process(), read [0,blk). */
void orc_blkconv_stream(orc_blkconv *c, const float *x, float *y, long n)
{
    const int blk = c->blk;
    for (long off = 0; off < n; off += blk) {
        const long m = (n - off) < blk ? (n - off) : blk;
        memcpy(c->buf, x + off, sizeof(float) * (size_t)m);
        for (long i = m; i < blk; i++) c->buf[i] = 0.0f;
        orc_blkconv_process(c);
        memcpy(y + off, c->buf, sizeof(float) * (size_t)m);
    }
}

/* =============================================================================
 * The resampling time law shared by resample and decimate
 * (libdsp/resample.cxx:119-150 == libdsp/decimate.cxx:96-127): outputs are taken at
 * upsampled-grid instants t (float32), pos = floor(t), mu = t - pos, linear
 * interpolation between polyphase samples pos and pos+1, t += rate*U in float32,
 * with a "leftover" output when pos+1 falls into the next chunk.
 * The sample source differs between the two classes and is passed as a callback.
 * ========================================================================== */
typedef float (*sample_fn)(void *self, int phase, int n);

typedef struct {
    int   pos;        /* m_pos */
    float mu;         /* m_mu */
    float last;       /* m_last_remain */
    int   leftover;   /* m_is_leftover */
} timelaw_t;

static int timelaw_run(timelaw_t *s, int U, int n_in, float *out, int out_len, float rate,
                       sample_fn get, void *self)
{
    int n_out = 0;
    float t = (float)s->pos + s->mu;                 /* resample.cxx:89, decimate.cxx:73 */
    const float step = rate * (float)U;              /* rate * m_n_phase, float*int -> float */

    if (s->leftover) {                               /* resample.cxx:119-123 */
        out[n_out++] = s->last * (1.0f - s->mu) + s->mu * get(self, 0, 0);
        s->leftover = 0;
        t += step;
    }
    for (;;) {                                       /* resample.cxx:125-148 */
        s->pos = (int)floorf(t);
        s->mu = t - (float)s->pos;
        const int pos1 = s->pos + 1;
        /* C '/' and '%' truncate toward zero, exactly as the reference's ints do */
        const int ph0 = s->pos % U, ph1 = pos1 % U;
        const int n0 = s->pos / U, n1 = pos1 / U;
        if (n0 >= n_in || n_out >= out_len) break;
        if (n1 >= n_in) {
            s->leftover = 1;
            s->last = get(self, ph0, n0);
            break;
        }
        const float a = get(self, ph0, n0);
        const float b = get(self, ph1, n1);
        out[n_out++] = a * (1.0f - s->mu) + s->mu * b;
        t += step;
    }
    s->pos -= n_in * U;                              /* resample.cxx:150 */
    return n_out;
}

/* =============================================================================
 * resample -- all-phases-then-pick polyphase interpolating resampler.
 * ========================================================================== */
struct orc_resample {
    int     U, plen, blksize;
    float **ptaps;   /* [U][plen]      m_phase_taps, resample.cxx:55-64 */
    float **pout;    /* [U][blksize]   m_out */
    float  *hist;    /* [max(blksize,plen)] newest first, m_history */
    timelaw_t tl;
};

/* libdsp/resample.cxx:37-69 */
orc_resample *orc_resample_create(const float *taps, int n_taps, int upsample, int blksize)
{
    orc_resample *r = (orc_resample *)calloc(1, sizeof(*r));
    r->U = upsample;
    r->blksize = blksize;
    r->plen = (n_taps + upsample - 1) / upsample;                    /* :43 */
    r->ptaps = (float **)calloc((size_t)upsample, sizeof(float *));
    r->pout = (float **)calloc((size_t)upsample, sizeof(float *));
    for (int j = 0; j < upsample; j++) {
        r->ptaps[j] = (float *)calloc((size_t)r->plen, sizeof(float));
        r->pout[j] = (float *)calloc((size_t)blksize, sizeof(float));
        for (int i = 0; i < r->plen; i++) {
            const int n = i * upsample + j;
            r->ptaps[j][i] = n < n_taps ? taps[n] : 0.0f;
        }
    }
    /* the reference allocates m_history[blksize] and needs plen-1 <= blksize
     * (SURVEY section 5 "latent preconditions"); the oracle allocates enough for either */
    const int hl = (blksize > r->plen ? blksize : r->plen) + 1;
    r->hist = (float *)calloc((size_t)hl, sizeof(float));
    return r;
}

static float resample_pick(void *self, int phase, int n)
{
    orc_resample *r = (orc_resample *)self;
    return r->pout[phase][n];
}

/* libdsp/resample.cxx:85-153 */
int orc_resample_process(orc_resample *r, const float *in, int n_in, float *out, int out_len,
                         float rate)
{
    if (n_in > r->blksize || rate < 1.0 / r->U) {                    /* :91-94 */
        printf("input parameter is wrong, rate <= 1/upsample, n_in <= blksize\n");
        return 0;
    }
    if (out_len < floorf(n_in * 1.0f / rate)) {                      /* :95-98 */
        printf("output buffer is not large enough");
        return 0;
    }
    for (int i = 0; i < n_in; i++) {                                 /* :100-114 */
        for (int j = 0; j < r->U; j++) {
            float acc = r->ptaps[j][0] * in[i];
            for (int n = 1; n < r->plen; n++) acc += r->ptaps[j][n] * r->hist[n - 1];
            r->pout[j][i] = acc;
        }
        for (int n = r->plen - 2; n > 0; n--) r->hist[n] = r->hist[n - 1];
        r->hist[0] = in[i];
    }
    return timelaw_run(&r->tl, r->U, n_in, out, out_len, rate, resample_pick, r);
}

/* Test helper (tests/test_gpu_fullsize.py: parity windows deep inside a 2^28-sample stream at a general rate): the time law of
 * `n_calls` consecutive process() calls of `n_in` samples each, WITHOUT their samples -- the same timelaw_run (resample.cxx:119-150)
 * picking zeros -- so that (m_pos, m_mu, m_is_leftover) arrive where the reference's would after those calls; the float32
 * recurrence has no closed form, it has to be walked.  Returns the number of outputs those calls emit.  The history and
 * m_last_remain are NOT those of the stream: the caller runs one real call before the window it checks. */
static float zero_pick(void *self, int phase, int n) { (void)self; (void)phase; (void)n; return 0.0f; }
long orc_resample_skip_calls(orc_resample *r, long n_calls, int n_in, float rate)
{
    if (n_in > r->blksize || rate < 1.0 / r->U || n_in < 1) return -1;
    const int cap = (int)((double)n_in * r->U) + 8;                  /* step >= 1: at most n_in U outputs per call */
    float *scratch = (float *)malloc(sizeof(float) * (size_t)cap);
    long total = 0;
    for (long c = 0; c < n_calls; c++) total += timelaw_run(&r->tl, r->U, n_in, scratch, cap, rate, zero_pick, NULL);
    free(scratch);
    return total;
}

/* test helpers: the time state alone (m_pos, m_mu, m_is_leftover), so that one walked object can hand its place to fresh ones */
void orc_resample_get_time(const orc_resample *r, int *pos, float *mu, int *leftover)
{
    *pos = r->tl.pos;
    *mu = r->tl.mu;
    *leftover = r->tl.leftover;
}
void orc_resample_set_time(orc_resample *r, int pos, float mu, int leftover)
{
    r->tl.pos = pos;
    r->tl.mu = mu;
    r->tl.leftover = leftover;
}

void orc_resample_destroy(orc_resample *r)
{
    if (!r) return;
    for (int j = 0; j < r->U; j++) { free(r->ptaps[j]); free(r->pout[j]); }
    free(r->ptaps); free(r->pout); free(r->hist);
    free(r);
}

/* =============================================================================
 * decimate -- same law, the two polyphase dot products per output on demand.
 * ========================================================================== */
struct orc_decimate {
    int    U, n_taps, blksize, len;
    float *taps;   /* odd-ized copy, decimate.cxx:42-51 */
    float *hist;   /* [n_taps + blksize], decimate.cxx:53-57 */
    float *cur;    /* m_in */
    timelaw_t tl;
};

/* libdsp/decimate.cxx:37-59 */
orc_decimate *orc_decimate_create(const float *taps, int n_taps, int upsample, int blksize)
{
    orc_decimate *d = (orc_decimate *)calloc(1, sizeof(*d));
    d->U = upsample;
    d->blksize = blksize;
    d->n_taps = (n_taps % 2 == 0) ? n_taps + 1 : n_taps;             /* :42-44 */
    d->taps = (float *)calloc((size_t)d->n_taps, sizeof(float));
    memcpy(d->taps, taps, sizeof(float) * (size_t)n_taps);           /* :46-51 */
    d->len = d->n_taps + blksize;
    d->hist = (float *)calloc((size_t)d->len, sizeof(float));
    return d;
}

/* libdsp/decimate.cxx:132-140 */
static float decimate_dot(void *self, int phase, int n)
{
    orc_decimate *d = (orc_decimate *)self;
    float acc = 0.0f;
    for (int m = phase, j = 0; m < d->n_taps; m += d->U, j++) acc += d->taps[m] * d->cur[n - j];
    return acc;
}

/* libdsp/decimate.cxx:69-129 */
int orc_decimate_process(orc_decimate *d, const float *in, int n_in, float *out, int out_len,
                         float rate)
{
    if (rate < 1.0) {                                                /* :75-78 */
        printf("rate should be larger than 1.0\n");
        return 0;
    }
    if (n_in > d->blksize) {                                         /* :79-82 */
        printf("number of samples should be less than blksize\n");
        return 0;
    }
    if (out_len < floorf(n_in * 1.0f / rate)) {                      /* :84-87 */
        printf("output buffer is not large enough");
        return 0;
    }
    memmove(d->hist, d->hist + n_in, sizeof(float) * (size_t)(d->len - n_in)); /* :89 */
    memcpy(d->hist + (d->len - n_in), in, sizeof(float) * (size_t)n_in);       /* :90 */
    d->cur = d->hist + (d->len - n_in);                                       /* :91 */
    return timelaw_run(&d->tl, d->U, n_in, out, out_len, rate, decimate_dot, d);
}

void orc_decimate_destroy(orc_decimate *d)
{
    if (!d) return;
    free(d->taps); free(d->hist);
    free(d);
}

/* =============================================================================
 * wire-format converters ("next" row N2)
 * ========================================================================== */
/* gr-simplefe/lib/source_c_impl.cc:121-132 : src_len bytes -> src_len/2 complex;
 * returns bytes written */
int orc_rx_u8_to_cf32(float *dst, const unsigned char *src, int src_len)
{
    const float qinv = 1.0f / 127.0f;
    int j = 0;
    for (int i = 0; i < src_len; i += 2, j++) {
        dst[2 * j]     = (float)(src[i] - 128) * qinv;
        dst[2 * j + 1] = (float)(src[i + 1] - 128) * qinv;
    }
    return j * 2 * (int)sizeof(float);
}

/* gr-simplefe/lib/source_f_impl.cc:120-129 */
int orc_rx_u8_to_f32(float *dst, const unsigned char *src, int src_len)
{
    const float qinv = 1.0f / 127.0f;
    for (int i = 0; i < src_len; i++) dst[i] = (float)(src[i] - 128) * qinv;
    return src_len * (int)sizeof(float);
}

/* gr-simplefe/lib/sink_f_impl.cc:117-143 (sink_c_impl.cc:118-144 is the same on
 * interleaved re,im): 4 floats -> 5 bytes; returns bytes written */
int orc_tx_f32_to_10bit(unsigned char *dst, const float *src, int src_len)
{
    int j = 0;
    for (int i = 0; i + 3 < src_len; i += 4) {
        unsigned short u[4];
        for (int k = 0; k < 4; k++)
            u[k] = (unsigned short)((((short)(src[i + k] * 511)) + 512) & 0x3FF);
        dst[j++] = (unsigned char)((u[0] >> 8) | ((u[1] >> 8) << 2) | ((u[2] >> 8) << 4) |
                                   ((u[3] >> 8) << 6));
        dst[j++] = (unsigned char)(u[0] & 0xFF);
        dst[j++] = (unsigned char)(u[1] & 0xFF);
        dst[j++] = (unsigned char)(u[2] & 0xFF);
        dst[j++] = (unsigned char)(u[3] & 0xFF);
    }
    return j;
}

/* =============================================================================
 * All-host-cores CPU baseline (bench.py cpu_baseline.all_cores; SURVEY.md 8(d) last row).
 * The reference classes are single-threaded, one object per stream (no threads anywhere in
 * libdsp/); the only way to use more cores on ONE stream is to cut it into spans, give each
 * span its own object and feed every object `ovl` samples of the stream in front of its span
 * so its carried state (blkconv.cxx:105-109 overlap / the resamplers' history) is the
 * stream's.  Outputs of the lead-in are dropped.  Timing/validation helper, not a product path.
 * ========================================================================== */
#include <pthread.h>
#include <sched.h>

/* Pin the calling thread to the idx-th CPU this process may run on: short-lived worker threads
 * are otherwise left time-slicing on the core that created them (measured here: 8 threads of
 * 70 ms each took 700 ms unpinned, 75 ms pinned). */
static void pin_to_nth_cpu(int idx)
{
    cpu_set_t allowed, one;
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return;
    const int n = CPU_COUNT(&allowed);
    if (n <= 1) return;
    int want = idx % n, seen = 0;
    for (int c = 0; c < CPU_SETSIZE; c++)
        if (CPU_ISSET(c, &allowed)) {
            if (seen == want) {
                CPU_ZERO(&one);
                CPU_SET(c, &one);
                (void)pthread_setaffinity_np(pthread_self(), sizeof(one), &one);
                return;
            }
            seen++;
        }
}

typedef struct {
    int kind;                  /* 0 blkconv, 1 resample, 2 decimate */
    const float *taps;
    int n_taps, p1, p2;        /* blkconv: fft_len, -; rs: upsample, blksize */
    float rate;
    const float *x;
    float *y;                  /* blkconv: y[n] (may be NULL: outputs discarded) */
    long lo, s, e;             /* feed x[lo..e), keep outputs of x[s..e) */
    long n_out;
    int  cpu;                  /* >= 0: pin to that index of the allowed set */
} mt_job;

static void *mt_run(void *arg)
{
    mt_job *j = (mt_job *)arg;
    const long span = j->e - j->lo;
    if (span <= 0) return NULL;
    if (j->cpu >= 0) pin_to_nth_cpu(j->cpu);
    if (j->kind == 0) {
        orc_blkconv *c = orc_blkconv_create(j->taps, j->n_taps, j->p1);
        const int blk = orc_blkconv_blksize(c);
        float *buf = orc_blkconv_buf(c);
        /* block by block through the object's own buffer, as a caller of the class would
         * (no span-sized scratch: fresh pages from N threads at once serialise in the kernel) */
        for (long off = j->lo; off < j->e; off += blk) {
            const long m = (j->e - off) < blk ? (j->e - off) : blk;
            memcpy(buf, j->x + off, sizeof(float) * (size_t)m);
            for (long i = m; i < blk; i++) buf[i] = 0.0f;
            orc_blkconv_process(c);
            if (j->y && off + m > j->s) {
                const long a = off < j->s ? j->s - off : 0;
                memcpy(j->y + off + a, buf + a, sizeof(float) * (size_t)(m - a));
            }
        }
        j->n_out = j->e - j->s;
        orc_blkconv_destroy(c);
    } else {
        const int B = j->p2;
        const int cap = (int)((float)B / j->rate) + 4;
        float *out = (float *)malloc(sizeof(float) * (size_t)cap);
        void *h = j->kind == 1 ? (void *)orc_resample_create(j->taps, j->n_taps, j->p1, B)
                               : (void *)orc_decimate_create(j->taps, j->n_taps, j->p1, B);
        long k = 0;
        for (long off = j->lo; off < j->e; off += B) {
            const int m = (int)((j->e - off) < B ? (j->e - off) : B);
            k += j->kind == 1 ? orc_resample_process((orc_resample *)h, j->x + off, m, out, cap, j->rate)
                              : orc_decimate_process((orc_decimate *)h, j->x + off, m, out, cap, j->rate);
        }
        j->n_out = k;
        if (j->kind == 1) orc_resample_destroy((orc_resample *)h);
        else orc_decimate_destroy((orc_decimate *)h);
        free(out);
    }
    return NULL;
}

static long mt_dispatch(mt_job proto, long n, long ovl, long quantum, int n_threads)
{
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    pthread_t th[256];
    mt_job jobs[256];
    long per = (n + n_threads - 1) / n_threads;
    per = (per + quantum - 1) / quantum * quantum;    /* cuts on whole phase periods */
    int used = 0;
    for (int t = 0; t < n_threads; t++) {
        long s = (long)t * per, e = s + per < n ? s + per : n;
        if (s >= n) break;
        jobs[t] = proto;
        jobs[t].s = s;
        jobs[t].e = e;
        jobs[t].lo = s - ovl > 0 ? (s - ovl) / quantum * quantum : 0;
        jobs[t].n_out = 0;
        jobs[t].cpu = t > 0 ? t : -1;      /* job 0 runs on the caller's thread, left where it is */
        used++;
    }
    for (int t = 1; t < used; t++) pthread_create(&th[t], NULL, mt_run, &jobs[t]);
    mt_run(&jobs[0]);
    long total = jobs[0].n_out;
    for (int t = 1; t < used; t++) {
        pthread_join(th[t], NULL);
        total += jobs[t].n_out;
    }
    return total;
}

long orc_blkconv_stream_mt(const float *taps, int n_taps, int fft_len, const float *x, float *y, long n,
                           int n_threads)
{
    mt_job p;
    memset(&p, 0, sizeof(p));
    p.kind = 0;
    p.taps = taps;
    p.n_taps = n_taps;
    p.p1 = fft_len;
    p.x = x;
    p.y = y;
    return mt_dispatch(p, n, n_taps - 1, 1, n_threads);
}

long orc_rs_stream_mt(int decimate_class, const float *taps, int n_taps, int upsample, int blksize, float rate,
                      const float *x, long n, long quantum, int n_threads)
{
    mt_job p;
    memset(&p, 0, sizeof(p));
    p.kind = decimate_class ? 2 : 1;
    p.taps = taps;
    p.n_taps = n_taps;
    p.p1 = upsample;
    p.p2 = blksize;
    p.rate = rate;
    p.x = x;
    return mt_dispatch(p, n, (n_taps + upsample - 1) / upsample + 1, quantum < 1 ? 1 : quantum, n_threads);
}
