// bpsk_gpu.cpp -- the reference's BPSK transmitter pipeline (examples/bpsk/bpsk.cxx:104-174) on
// the GPU path, end to end, with its threading model and WITHOUT the USB device:
//
//   process thread : bits -> impulse train (10 samples/symbol) -> blkconv RRC pulse shaping
//                    (drop-in class, include/blkconv.h: the filtering runs on the MI355X)
//                    -> ring_buffer<float> (include/ringbuf.h) under mutex + condvar
//   consumer thread: stands in for libsimpleFE's libusb event thread (tx_callback, bpsk.cxx:104-119):
//                    pulls `length` bytes through the ring's converting read --
//                    4 floats -> 5 bytes, 10-bit offset binary (bpsk.cxx:76-101) -- and appends
//                    them to a file instead of a USB transfer.
//
// Deterministic (an LCG replaces rand()) so tests/test_gpu_dropin.py can rebuild the same
// stream with the oracle.   usage: bpsk_gpu <n_blocks> <out.bin> [fft_len] [n_taps taps.f32]
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/blkconv.h"
#include "../../include/ringbuf.h"

#define SAMPLES_PER_SYMBOL 10
#define SCALING_FACTOR (.85f / 1.35f)

static pthread_cond_t buf_cond = PTHREAD_COND_INITIALIZER;
static pthread_mutex_t buf_mutex = PTHREAD_MUTEX_INITIALIZER;
static volatile int producer_done = 0;
static int n_blocks = 8, fft_len = 2048;
static std::vector<float> g_taps;

static int calc_num_samples(int bytes) { return bytes / 5 * 4; }

// 4 floats -> 5 bytes (the reference's convert_samples_to_bytes, bpsk.cxx:76-101, restated)
static int samples_to_bytes(void *dst, void *src, int n)
{
    const float *in = static_cast<const float *>(src);
    unsigned char *out = static_cast<unsigned char *>(dst);
    int j = 0;
    for (int i = 0; i + 3 < n; i += 4) {
        unsigned short u[4];
        for (int k = 0; k < 4; k++) u[k] = (unsigned short)((((short)(in[i + k] * 511)) + 512) & 0x3FF);
        out[j++] = (unsigned char)((u[0] >> 8) | ((u[1] >> 8) << 2) | ((u[2] >> 8) << 4) | ((u[3] >> 8) << 6));
        for (int k = 0; k < 4; k++) out[j++] = (unsigned char)(u[k] & 0xFF);
    }
    return j;
}

static unsigned lcg_state = 12345u;
static unsigned lcg() { lcg_state = lcg_state * 1664525u + 1013904223u; return lcg_state >> 1; }

static void *process(void *data)
{
    ring_buffer<float> *buf = static_cast<ring_buffer<float> *>(data);
    blkconv pulse_filter(g_taps.data(), (int)g_taps.size(), fft_len);      // bpsk.cxx:125
    const int blk_size = pulse_filter.get_blksize();
    float *proc_buf = pulse_filter.get_process_buf();                      // pinned; cached once (bpsk.cxx:127)
    int n_phase = 0, done = 0;
    while (done < n_blocks) {
        pthread_mutex_lock(&buf_mutex);
        if (buf->get_space() >= blk_size) {
            int n_input = 0;
            if (n_phase > 0) {                                             // finish the symbol cut by the block edge
                for (int i = n_phase; i < SAMPLES_PER_SYMBOL && n_input < blk_size; i++) proc_buf[n_input++] = 0.0f;
                n_phase = 0;
            }
            while (n_input < blk_size) {
                const unsigned word = lcg();
                for (int j = 0; j < 31 && n_input < blk_size; j++) {
                    proc_buf[n_input++] = (word & (1u << j)) ? -SCALING_FACTOR : SCALING_FACTOR;
                    for (n_phase = 1; n_phase < SAMPLES_PER_SYMBOL && n_input < blk_size; n_phase++) proc_buf[n_input++] = 0.0f;
                    if (n_phase == SAMPLES_PER_SYMBOL) n_phase = 0;
                }
            }
            pulse_filter.process();                                        // GPU
            buf->write(proc_buf, blk_size);
            done++;
            pthread_cond_broadcast(&buf_cond);
        } else {
            pthread_cond_wait(&buf_cond, &buf_mutex);
        }
        pthread_mutex_unlock(&buf_mutex);
    }
    pthread_mutex_lock(&buf_mutex);
    producer_done = 1;
    pthread_cond_broadcast(&buf_cond);
    pthread_mutex_unlock(&buf_mutex);
    return NULL;
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: bpsk_gpu <n_blocks> <out.bin> [fft_len] [taps.f32]\n"); return 2; }
    n_blocks = atoi(argv[1]);
    if (argc > 3) fft_len = atoi(argv[3]);
    if (argc > 4) {
        FILE *f = fopen(argv[4], "rb");
        if (!f) { perror(argv[4]); return 2; }
        float v;
        while (fread(&v, 4, 1, f) == 1) g_taps.push_back(v);
        fclose(f);
    } else {
        for (int k = 0; k < 111; k++) g_taps.push_back(k == 55 ? 1.0f : 0.0f);
    }
    FILE *out = fopen(argv[2], "wb");
    if (!out) { perror(argv[2]); return 2; }

    ring_buffer<float> dev_buf(20000);
    pthread_t proc_thread;
    pthread_create(&proc_thread, NULL, process, &dev_buf);

    // the "USB" side: fixed-size transfers, like the ISO callback (bpsk.cxx:104-119)
    const int length = 5 * 512;
    std::vector<unsigned char> xfer(length);
    long total = 0;
    for (;;) {
        pthread_mutex_lock(&buf_mutex);
        while (dev_buf.get_count() < calc_num_samples(length) && !producer_done) pthread_cond_wait(&buf_cond, &buf_mutex);
        int got = 0;
        if (dev_buf.get_count() >= calc_num_samples(length)) {
            got = dev_buf.read(xfer.data(), length, samples_to_bytes, calc_num_samples);
            pthread_cond_broadcast(&buf_cond);
        }
        const int finished = producer_done && dev_buf.get_count() < calc_num_samples(length);
        pthread_mutex_unlock(&buf_mutex);
        if (got) { fwrite(xfer.data(), 1, length, out); total += length; }
        if (finished) break;
    }
    pthread_join(proc_thread, NULL);
    fclose(out);
    printf("%ld bytes\n", total);
    return 0;
}
