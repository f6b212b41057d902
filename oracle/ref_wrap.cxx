// ref_wrap.cxx -- extern "C" handles around the UNMODIFIED reference classes.
//
// TEST INFRASTRUCTURE ONLY.  Compiled by oracle/Makefile together with
// /root/reference/libdsp/resample.cxx and decimate.cxx, from where they lie, into
// oracle/_ref/libsferef.so (git-ignored; never copied into the repo).  Used to pin
// the C restatement (sfe_oracle.c), to generate tests/golden/*.npz, and as the
// "reference" CPU baseline in bench.py.
//
// blkconv.cxx is built separately (ref_wrap_blkconv.cxx): it needs an FFTW3-API library, and
// the one this image has (ROCm's libhipfftw) runs on a GPU box only.
#include "decimate.h"   // -I/root/reference/libdsp
#include "resample.h"

extern "C" {

void *ref_resample_create(float *taps, int n_taps, int upsample, int blksize)
{
    return new resample(taps, n_taps, upsample, blksize);
}
int ref_resample_process(void *h, float *in, int n_in, float *out, int out_len, float rate)
{
    return static_cast<resample *>(h)->process(in, n_in, out, out_len, rate);
}
void ref_resample_destroy(void *h) { delete static_cast<resample *>(h); }

void *ref_decimate_create(float *taps, int n_taps, int upsample, int blksize)
{
    return new decimate(taps, n_taps, upsample, blksize);
}
int ref_decimate_process(void *h, float *in, int n_in, float *out, int out_len, float rate)
{
    return static_cast<decimate *>(h)->process(in, n_in, out, out_len, rate);
}
void ref_decimate_destroy(void *h) { delete static_cast<decimate *>(h); }

}  // extern "C"

// All-host-cores baseline over the reference's OWN classes (bench.py cpu_baseline.all_cores):
// one stream cut into n_threads spans on whole phase periods (`quantum` input samples), one
// reference object per span, each fed phase_len+1 samples of lead-in so that its history is the
// stream's; the lead-in's outputs are dropped.  Returns the outputs produced (timing helper).
#include <algorithm>
#include <thread>
#include <vector>

#include <pthread.h>
#include <sched.h>

// short-lived workers are otherwise left time-slicing on the core that created them
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

template <class T>
static long run_span(float *taps, int n_taps, int U, int B, float rate, float *x, long lo, long e)
{
    T obj(taps, n_taps, U, B);
    std::vector<float> out((size_t)((float)B / rate) + 4);
    long k = 0;
    for (long off = lo; off < e; off += B)
        k += obj.process(x + off, (int)std::min<long>(B, e - off), out.data(), (int)out.size(), rate);
    return k;
}

extern "C" long ref_rs_stream_mt(int decimate_class, float *taps, int n_taps, int upsample, int blksize,
                                 float rate, float *x, long n, long quantum, int n_threads)
{
    n_threads = std::max(1, std::min(n_threads, 256));
    quantum = std::max<long>(1, quantum);
    long per = (n + n_threads - 1) / n_threads;
    per = (per + quantum - 1) / quantum * quantum;
    const long ovl = (n_taps + upsample - 1) / upsample + 1;
    std::vector<long> got(n_threads, 0);
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; t++) {
        const long s = (long)t * per, e = std::min(s + per, n);
        if (s >= n) break;
        const long lo = s - ovl > 0 ? (s - ovl) / quantum * quantum : 0;
        th.emplace_back([=, &got] {
            pin_to_nth_cpu(t);
            got[t] = decimate_class ? run_span<decimate>(taps, n_taps, upsample, blksize, rate, x, lo, e)
                                    : run_span<resample>(taps, n_taps, upsample, blksize, rate, x, lo, e);
        });
    }
    long total = 0;
    for (size_t t = 0; t < th.size(); t++) {
        th[t].join();
        total += got[t];
    }
    return total;
}
