// ring_buffer<T> (include/ringbuf.h) under the three scenarios the reference asserts in
// gr-simplefe/lib/qa_simplefe.cc:103-166 (its only asserting tests): simple write/read,
// wrap-around with a complex->float converting read, and wrap-around with a u8->complex
// converting read.  Pure host; built and run by tests/test_host_logic.py.
#include <complex>
#include <cstdio>
#include <cmath>

// -DSFE_REF_RINGBUF -I<reference>/libdsp: the same scenarios on the reference's own header (run in
// the authoring container only), showing the two classes behave alike where the reference asserts.
#ifdef SFE_REF_RINGBUF
#include "ringbuf.h"
#else
#include "../../include/ringbuf.h"
#endif

typedef std::complex<float> cf;
static int fails = 0;
#define CHECK(c) do { if (!(c)) { printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); fails++; } } while (0)

static int cplx_to_floats(void *dst, void *src, int n)
{
    cf *s = static_cast<cf *>(src);
    float *o = static_cast<float *>(dst);
    for (int i = 0; i < n; i++) { o[2 * i] = s[i].real(); o[2 * i + 1] = s[i].imag(); }
    return n * 2 * (int)sizeof(float);
}
static int bytes_for_cplx(int dst_bytes) { return dst_bytes / (int)sizeof(cf); }

static int u8_to_cplx(void *dst, void *src, int n)
{
    unsigned char *s = static_cast<unsigned char *>(src);
    cf *o = static_cast<cf *>(dst);
    int k = 0;
    for (int i = 0; i < n; i += 2) o[k++] = cf(s[i] * 1.0f, s[i + 1] * 1.0f);
    return k * (int)sizeof(cf);
}
static int two_per_item(int dst_items) { return dst_items * 2; }

int main()
{
    const int N = 16;
    cf w[2 * N];
    for (int i = 0; i < 2 * N; i++) w[i] = cf(i * 1.0f, i * 2.0f);
    unsigned char wb[4 * N];
    for (int i = 0; i < 4 * N; i++) wb[i] = (unsigned char)i;

    {   // qa_simplefe.cc:103-116
        ring_buffer<cf> rb(N);
        float r[4 * N] = {0};
        CHECK(rb.write(w, 16) == 16);
        CHECK(rb.get_space() == 0 && rb.get_count() == 16);
        CHECK(rb.write(w, 1) == 0);                       // all-or-nothing
        CHECK(rb.read(r, 16 * 2 * sizeof(float), cplx_to_floats, bytes_for_cplx) == 16);
        for (int i = 0; i < 32; i += 2) {
            CHECK(std::fabs(r[i] - (i / 2) * 1.0f) < 1e-6f);
            CHECK(std::fabs(r[i + 1] - (i / 2) * 2.0f) < 1e-6f);
        }
        CHECK(rb.read(r, 8, cplx_to_floats, bytes_for_cplx) == 0);   // nothing queued
    }
    {   // qa_simplefe.cc:118-140: write 12, read 6, write 10 (wraps), read 16 (wraps)
        ring_buffer<cf> rb(N);
        float r[4 * N] = {0};
        rb.write(w, 12);
        CHECK(rb.read(r, 6 * 2 * sizeof(float), cplx_to_floats, bytes_for_cplx) == 6);
        CHECK(rb.write(w, 10) == 10);
        CHECK(rb.read(r + 12, 16 * 2 * sizeof(float), cplx_to_floats, bytes_for_cplx) == 16);
        for (int i = 0; i < (16 + 6) * 2; i += 2) {
            float x = (float)((i % 24) / 2);
            CHECK(std::fabs(r[i] - x) < 1e-6f);
            CHECK(std::fabs(r[i + 1] - 2.0f * x) < 1e-6f);
        }
        CHECK(rb.get_count() == 0);
    }
    {   // qa_simplefe.cc:143-164: bytes in, complex out
        ring_buffer<unsigned char> rb(2 * N);
        cf r[2 * N];
        rb.write(wb, 24);
        CHECK(rb.read(r, 6, u8_to_cplx, two_per_item) == 12);
        CHECK(rb.write(wb, 20) == 20);
        CHECK(rb.read(r + 6, 16, u8_to_cplx, two_per_item) == 32);
        for (int i = 0, j = 0; i < 16 + 6; i++, j += 2) {
            CHECK((int)r[i].real() == j % 24);
            CHECK((int)r[i].imag() == (j + 1) % 24);
        }
    }
    {   // default ctor + alloc_buffer, null callbacks
        ring_buffer<float> rb;
        CHECK(rb.get_space() == 0);
        rb.alloc_buffer(8);
        float v[8] = {1, 2, 3, 4, 5, 6, 7, 8}, o[8];
        CHECK(rb.write(v, 8) == 8);
        CHECK(rb.read(o, 8, 0, 0) == 0);
    }
    printf(fails ? "ringbuf: %d failures\n" : "ringbuf ok\n", fails);
    return fails ? 1 : 0;
}
