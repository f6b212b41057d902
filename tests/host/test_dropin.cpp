// The drop-in C++ classes (include/blkconv.h, resample.h, decimate.h) driven the way the
// reference's callers drive them.  Needs a GPU to RUN; tests/test_host_logic.py only checks
// that it compiles and links on the CPU box, tests/test_gpu_dropin.py runs it.
//
//   test_dropin blkconv                     the scenario of libdsp/test/test_blkconv.cxx:5-33
//   test_dropin rs <resample|decimate> <taps.f32> <x.f32> <U> <B> <out_len> <rate> <y.f32>
//                                           the driver loop of libdsp/test/test_decimate.py:22-25;
//                                           writes outputs raw, prints per-call n_out
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/blkconv.h"
#include "../../include/decimate.h"
#include "../../include/resample.h"

static std::vector<float> slurp(const char *path)
{
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<float> v((size_t)n / sizeof(float));
    if (fread(v.data(), sizeof(float), v.size(), f) != v.size()) exit(2);
    fclose(f);
    return v;
}

template <class C>
static int drive(std::vector<float> &taps, std::vector<float> &x, int U, int B, int out_len, float rate,
                 const char *ypath)
{
    C obj(taps.data(), (int)taps.size(), U, B);
    std::vector<float> out((size_t)out_len + 1), y;
    for (size_t off = 0; off < x.size(); off += (size_t)B) {
        int n_in = (int)((x.size() - off) < (size_t)B ? (x.size() - off) : (size_t)B);
        int n = obj.process(x.data() + off, n_in, out.data(), out_len, rate);
        printf("%d\n", n);
        y.insert(y.end(), out.begin(), out.begin() + n);
    }
    FILE *f = fopen(ypath, "wb");
    fwrite(y.data(), sizeof(float), y.size(), f);
    fclose(f);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc >= 2 && !strcmp(argv[1], "blkconv")) {
        float taps[] = {1.0f, 1.0f, 1.0f, 1.0f, 1.0f};
        blkconv conv(taps, 5, 32);
        float *buf = conv.get_process_buf();
        int len = conv.get_blksize();
        printf("blksize = %d\n", len);
        for (int i = 0; i < len; i++) buf[i] = 1.0f;
        conv.process();
        for (int i = 0; i < len; i++) printf("%.9g\n", buf[i]);
        for (int i = 0; i < len; i++) buf[i] = 0.0f;
        conv.process();
        for (int i = 0; i < len; i++) printf("%.9g\n", buf[i]);
        return 0;
    }
    if (argc == 10 && !strcmp(argv[1], "rs")) {
        std::vector<float> taps = slurp(argv[3]), x = slurp(argv[4]);
        int U = atoi(argv[5]), B = atoi(argv[6]), out_len = atoi(argv[7]);
        float rate = (float)atof(argv[8]);
        if (!strcmp(argv[2], "resample")) return drive<resample>(taps, x, U, B, out_len, rate, argv[9]);
        return drive<decimate>(taps, x, U, B, out_len, rate, argv[9]);
    }
    fprintf(stderr, "usage: see header comment\n");
    return 2;
}
