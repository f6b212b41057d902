// Drives include/gr_sfe/blocks.h the way the GNU Radio scheduler drives a block: repeated
// work()/general_work() calls with scheduler-sized item counts.  Needs a GPU to run.
//   test_gr_blocks <fir|fir_sync|fir_f|decimate|resample|decimate_f|resample_f> <taps.f32> <x> <y> [decim] [interp]
//   test_gr_blocks bank <taps.f32> <x> <y> <n_channels> <blocks>   fir_bank_ccf_sync: one port per channel, channel blocks on device 0
//   test_gr_blocks rate <taps.f32> <x> <y>     4096-item calls through fir_ccf (batched) and fir_ccf_sync;
//                                              prints "<items/s batched> <items/s sync>"; y = batched output
// (_f: float items, otherwise gr_complex items)
#include <chrono>
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

static const int sizes[] = {4096, 1000, 8191, 37, 16384};     // what a scheduler hands out

// a pipe-backed rate-changing block driven the way the scheduler would: whatever input is left (at most
// one call size), room for one call size of outputs; after the input ends, input-less calls until a
// call produces nothing
template <int W, class Sptr>
static std::vector<float> run_rate_block(Sptr b, const std::vector<float> &x, size_t out_estimate)
{
    const int n = (int)(x.size() / W);
    std::vector<float> y(W * (out_estimate + 4096));
    int off = 0, produced = 0, si = 0, idle = 0;
    while (idle < 2) {
        const int room = sizes[si++ % 5];
        gr_vector_int req(1, 0);
        b->forecast(room, req);
        const int avail = n - off < room ? n - off : room;
        if (avail < req[0]) break;                        // nothing pending and no input left
        gr_vector_int nin(1, avail);
        gr_vector_const_void_star in(1, x.data() + W * (size_t)off);
        gr_vector_void_star out(1, y.data() + W * (size_t)produced);
        const int r = b->general_work(room, nin, in, out);
        off += b->consumed();
        produced += r;
        idle = (r == 0 && b->consumed() == 0 && off == n) ? idle + 1 : 0;
    }
    y.resize(W * (size_t)produced);
    return y;
}

template <class Block, int W>
static std::vector<float> run_decimate(const std::vector<float> &taps, const std::vector<float> &x, unsigned D)
{
    return run_rate_block<W>(Block::make(taps, D, 8192), x, x.size() / W / D);
}

template <class Block, int W>
static std::vector<float> run_resample(const std::vector<float> &taps, const std::vector<float> &x, unsigned D, unsigned I)
{
    return run_rate_block<W>(Block::make(I, D, taps, 8192), x, (size_t)((double)(x.size() / W) * I / D));
}

// general_work the way the scheduler calls it: whatever input is left (at most `call` items), room
// for `call` outputs; after the input ends, calls with no input until everything has come out
template <class Block, int W>
static bool run_fir_batched_into(typename Block::sptr b, const std::vector<float> &x, std::vector<float> &y, const int *call_sizes, int n_sizes)
{
    const int n = (int)(x.size() / W);
    int off = 0, produced = 0, si = 0;
    while (produced < n) {
        const int room = call_sizes[si++ % n_sizes];
        gr_vector_int req(1, 0);
        b->forecast(room, req);
        int avail = n - off < room ? n - off : room;
        if (avail < req[0]) return false;      // the block asks for input that will never come
        gr_vector_int nin(1, avail);
        gr_vector_const_void_star in(1, x.data() + W * (size_t)off);
        gr_vector_void_star out(1, y.data() + W * (size_t)produced);
        int r = b->general_work(room < n - produced ? room : n - produced, nin, in, out);
        off += b->consumed();
        produced += r;
    }
    return true;
}

template <class Block, int W>
static std::vector<float> run_fir_batched(typename Block::sptr b, const std::vector<float> &x, const int *call_sizes, int n_sizes)
{
    std::vector<float> y(x.size());
    if (!run_fir_batched_into<Block, W>(b, x, y, call_sizes, n_sizes)) y.clear();
    return y;
}

static std::vector<unsigned char> slurp_bytes(const char *p)
{
    FILE *f = fopen(p, "rb");
    if (!f) { perror(p); exit(2); }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<unsigned char> v((size_t)n);
    if (fread(v.data(), 1, v.size(), f) != v.size()) exit(2);
    fclose(f);
    return v;
}

// a wire-format block the way the scheduler drives it: byte streams in and / or out; in_sz / out_sz = item sizes,
// out_mult = the block's output multiple (5 for the 10-bit format); input-less calls after the input ends, until
// a call produces nothing
template <class Sptr>
static std::vector<unsigned char> run_wire(Sptr b, const unsigned char *x, size_t n_items, int in_sz, int out_sz, size_t out_items_max)
{
    std::vector<unsigned char> y((out_items_max + 65536) * (size_t)out_sz);
    size_t off = 0, produced = 0;
    int si = 0, idle = 0;
    const int mult = b->output_multiple();
    while (idle < 2) {
        int room = sizes[si++ % 5];
        room = room / mult * mult;
        if (room == 0) room = mult;
        gr_vector_int req(1, 0);
        b->forecast(room, req);
        const int avail = (int)(n_items - off < (size_t)sizes[si % 5] ? n_items - off : (size_t)sizes[si % 5]);
        if (avail < req[0]) break;
        gr_vector_int nin(1, avail);
        gr_vector_const_void_star in(1, x + off * (size_t)in_sz);
        gr_vector_void_star out(1, y.data() + produced * (size_t)out_sz);
        const int r = b->general_work(room, nin, in, out);
        if (r % mult) exit(4);
        off += (size_t)b->consumed();
        produced += (size_t)r;
        idle = (r == 0 && b->consumed() == 0 && off == n_items) ? idle + 1 : 0;
    }
    y.resize(produced * (size_t)out_sz);
    return y;
}

int main(int argc, char **argv)
{
    if (argc < 5) return 2;
    if (!strcmp(argv[1], "rx_fir") || !strcmp(argv[1], "fir_tx") || !strcmp(argv[1], "rx_fir_tx")) {
        // wire-format blocks: x is a byte file (u8 (I,Q) pairs) for rx_*, a float file (interleaved cf32) for fir_tx
        std::vector<float> taps = slurp(argv[2]);
        std::vector<unsigned char> xb = slurp_bytes(argv[3]), yb;
        if (!strcmp(argv[1], "rx_fir"))
            yb = run_wire(gr::sfe::rx_fir_bc::make(taps, 8192), xb.data(), xb.size() / 2, 2, 8, xb.size() / 2);
        else if (!strcmp(argv[1], "fir_tx"))
            yb = run_wire(gr::sfe::fir_tx_cb::make(taps, 8192), xb.data(), xb.size() / 8, 8, 1, xb.size() / 8 / 2 * 5);
        else
            yb = run_wire(gr::sfe::rx_fir_tx_bb::make(taps, 8192), xb.data(), xb.size() / 2, 2, 1, xb.size() / 2 / 2 * 5);
        FILE *f = fopen(argv[4], "wb");
        fwrite(yb.data(), 1, yb.size(), f);
        fclose(f);
        printf("%zu\n", yb.size());
        return 0;
    }
    if (!strcmp(argv[1], "bank")) {
        // test_gr_blocks bank <taps> <x: n_channels rows of n gr_complex> <y> <n_channels> <blocks>: every block on device 0
        std::vector<float> taps = slurp(argv[2]), x = slurp(argv[3]);
        const int nch = atoi(argv[5]), nblk = atoi(argv[6]);
        const int n = (int)(x.size() / 2 / nch);
        std::vector<float> y(x.size());
        gr::sfe::fir_bank_ccf_sync::sptr b = gr::sfe::fir_bank_ccf_sync::make(taps, nch, std::vector<int>(nblk, 0), 5000);
        int si = 0;
        for (int off = 0; off < n;) {
            int m = sizes[si++ % 5];
            if (m > n - off) m = n - off;
            gr_vector_const_void_star in(nch);
            gr_vector_void_star out(nch);
            for (int c = 0; c < nch; c++) {
                in[c] = x.data() + 2 * ((size_t)c * n + off);
                out[c] = y.data() + 2 * ((size_t)c * n + off);
            }
            if (b->work(m, in, out) != m) return 1;
            off += m;
        }
        FILE *f = fopen(argv[4], "wb");
        fwrite(y.data(), 4, y.size(), f);
        fclose(f);
        printf("%d\n", n);
        return 0;
    }
    std::vector<float> taps = slurp(argv[2]), x = slurp(argv[3]);
    const int n = (int)(x.size() / 2);
    std::vector<float> y;
    int si = 0;
    if (!strcmp(argv[1], "fir")) {
        y = run_fir_batched<gr::sfe::fir_ccf, 2>(gr::sfe::fir_ccf::make(taps, 8192), x, sizes, 5);
    } else if (!strcmp(argv[1], "fir_f")) {
        y = run_fir_batched<gr::sfe::fir_fff, 1>(gr::sfe::fir_fff::make(taps, 4096), x, sizes, 5);
    } else if (!strcmp(argv[1], "rate")) {
        static const int k4096[] = {4096};
        gr::sfe::fir_ccf::sptr warm = gr::sfe::fir_ccf::make(taps);
        run_fir_batched<gr::sfe::fir_ccf, 2>(warm, x, k4096, 1);
        gr::sfe::fir_ccf::sptr bb = gr::sfe::fir_ccf::make(taps);     // pinned batches are allocated here, not in the timed part
        y.assign(x.size(), 0.0f);                                        // output pages touched before timing, as for the sync run
        auto t0 = std::chrono::steady_clock::now();
        if (!run_fir_batched_into<gr::sfe::fir_ccf, 2>(bb, x, y, k4096, 1)) return 3;
        const double tb = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        gr::sfe::fir_ccf_sync::sptr bs = gr::sfe::fir_ccf_sync::make(taps);
        std::vector<float> ys(x.size(), 0.0f);
        t0 = std::chrono::steady_clock::now();
        for (int off = 0; off < n;) {
            int m = 4096 < n - off ? 4096 : n - off;
            gr_vector_const_void_star in(1, x.data() + 2 * (size_t)off);
            gr_vector_void_star out(1, ys.data() + 2 * (size_t)off);
            if (bs->work(m, in, out) != m) return 1;
            off += m;
        }
        const double ts = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("%.6g %.6g\n", n / tb, n / ts);
    } else if (!strcmp(argv[1], "fir_sync")) {
        gr::sfe::fir_ccf_sync::sptr b = gr::sfe::fir_ccf_sync::make(taps);
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
        y = run_decimate<gr::sfe::decimate_ccf, 2>(taps, x, (unsigned)atoi(argv[5]));
    } else if (!strcmp(argv[1], "decimate_f")) {
        y = run_decimate<gr::sfe::decimate_fff, 1>(taps, x, (unsigned)atoi(argv[5]));
    } else if (!strcmp(argv[1], "resample_f")) {
        y = run_resample<gr::sfe::rational_resampler_fff, 1>(taps, x, (unsigned)atoi(argv[5]), (unsigned)atoi(argv[6]));
    } else {
        y = run_resample<gr::sfe::rational_resampler_ccf, 2>(taps, x, (unsigned)atoi(argv[5]), (unsigned)atoi(argv[6]));
    }
    if (y.empty()) return 3;
    const size_t W = (strlen(argv[1]) > 2 && !strcmp(argv[1] + strlen(argv[1]) - 2, "_f")) ? 1 : 2;
    FILE *f = fopen(argv[4], "wb");
    fwrite(y.data(), 4, y.size(), f);
    fclose(f);
    printf("%zu\n", y.size() / W);
    return 0;
}
