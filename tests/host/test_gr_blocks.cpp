// Drives include/gr_sfe/blocks.h the way the GNU Radio scheduler drives a block: repeated
// work()/general_work() calls with scheduler-sized item counts.  Needs a GPU to run.
//   test_gr_blocks <fir|decimate|resample> <taps.f32> <x.cf32> <y.cf32> [decim] [interp]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/gr_sfe/blocks.h"

static std::vector<float> slurp(const char *p)
{
    FILE *f = fopen(p, "rb");
    if (!f) { perror(p); exit(2); }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<float> v((size_t)n / 4);
    if (fread(v.data(), 4, v.size(), f) != v.size()) exit(2);
    fclose(f);
    return v;
}

int main(int argc, char **argv)
{
    if (argc < 5) return 2;
    std::vector<float> taps = slurp(argv[2]), x = slurp(argv[3]);
    const int n = (int)(x.size() / 2);
    std::vector<float> y;
    const int sizes[] = {4096, 1000, 8191, 37, 16384};     // what a scheduler hands out
    int si = 0;
    if (!strcmp(argv[1], "fir")) {
        gr::sfe::fir_ccf::sptr b = gr::sfe::fir_ccf::make(taps);
        y.resize(x.size());
        for (int off = 0; off < n;) {
            int m = sizes[si++ % 5];
            if (m > n - off) m = n - off;
            gr_vector_const_void_star in(1, x.data() + 2 * (size_t)off);
            gr_vector_void_star out(1, y.data() + 2 * (size_t)off);
            if (b->work(m, in, out) != m) return 1;
            off += m;
        }
    } else if (!strcmp(argv[1], "decimate")) {
        const unsigned D = (unsigned)atoi(argv[5]);
        gr::sfe::decimate_ccf::sptr b = gr::sfe::decimate_ccf::make(taps, D, 4096);
        y.resize(x.size() / D + 64);
        int produced = 0;
        for (int off = 0; off + (int)D <= n;) {
            int m = sizes[si++ % 5] / (int)D;                // output items this call
            if (m < 1) m = 1;
            if ((long long)m * D > n - off) m = (n - off) / (int)D;
            gr_vector_const_void_star in(1, x.data() + 2 * (size_t)off);
            gr_vector_void_star out(1, y.data() + 2 * (size_t)produced);
            int r = b->work(m, in, out);
            produced += r;
            off += m * (int)D;
        }
        y.resize(2 * (size_t)produced);
    } else {
        const unsigned D = (unsigned)atoi(argv[5]), I = (unsigned)atoi(argv[6]);
        gr::sfe::rational_resampler_ccf::sptr b = gr::sfe::rational_resampler_ccf::make(I, D, taps, 4096);
        y.resize((size_t)(2.0 * n * I / D) + 1024);
        int produced = 0;
        for (int off = 0; off < n;) {
            int room = sizes[si++ % 5];
            gr_vector_int nin(1, n - off);
            gr_vector_const_void_star in(1, x.data() + 2 * (size_t)off);
            gr_vector_void_star out(1, y.data() + 2 * (size_t)produced);
            int r = b->general_work(room, nin, in, out);
            produced += r;
            if (b->consumed() == 0 && r == 0) { if (room < 8) continue; }
            off += b->consumed();
        }
        y.resize(2 * (size_t)produced);
    }
    FILE *f = fopen(argv[4], "wb");
    fwrite(y.data(), 4, y.size(), f);
    fclose(f);
    printf("%zu\n", y.size() / 2);
    return 0;
}
