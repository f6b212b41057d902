// api.hip -- the C ABI of libsfe_dsp.so (include/sfe_dsp.h), part 1: errors, devices, memory, timers, the synthetic
// stream, the wire-format converters.  Host code only; kernels live in fir_fft.hip / polyphase.hip / poly_fft.hip /
// poly_gen.hip / util.hip; the handles and what the host files share are in host.h.
#include "host.h"

namespace sfe {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char *what)
{
    set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    return e == hipErrorOutOfMemory ? SFE_ENOMEM : (e == hipErrorNoDevice ? SFE_ENODEV : SFE_EHIP);
}

int device_cu_count()
{
    constexpr int MAXDEV = 64;
    static std::atomic<int> cache[MAXDEV];          // zero-initialised; 0 = not asked yet
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return 256;
    int c = cache[dev].load(std::memory_order_relaxed);
    if (c > 0) return c;
    if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = 256;
    cache[dev].store(c, std::memory_order_relaxed);   // racing threads store the same value
    return c;
}

// [a, a + an) and [b, b + bn) share a byte
bool ranges_overlap(const void *a, size_t an, const void *b, size_t bn)
{
    const uintptr_t pa = reinterpret_cast<uintptr_t>(a), pb = reinterpret_cast<uintptr_t>(b);
    return an && bn && pa < pb + bn && pb < pa + an;
}

int use_device(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no HIP device available (%s): libsfe_dsp has no CPU fallback",
                  e != hipSuccess ? hipGetErrorString(e) : "device count 0");
        return SFE_ENODEV;
    }
    if (device < 0 || device >= n) {
        set_error("device %d out of range (0..%d)", device, n - 1);
        return SFE_EINVAL;
    }
    SFE_HIP(hipSetDevice(device));
    return SFE_OK;
}

}  // namespace sfe

using namespace sfe;

extern "C" {

const char *sfe_dsp_version(void) { return "simplefe_amd 0.1 (gfx950)"; }
const char *sfe_dsp_last_error(void) { return g_err; }

int sfe_dsp_device_count(int *count)
{
    if (!count) return SFE_EINVAL;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    *count = (e == hipSuccess) ? n : 0;
    return SFE_OK;
}

int sfe_dsp_set_device(int device) { return use_device(device); }

int sfe_dsp_get_device(int *device)
{
    if (!device) return SFE_EINVAL;
    *device = 0;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        set_error("no HIP device available: libsfe_dsp has no CPU fallback");
        return SFE_ENODEV;
    }
    SFE_HIP(hipGetDevice(device));
    return SFE_OK;
}

int sfe_dsp_sync(sfe_stream_t stream)
{
    SFE_HIP(hipStreamSynchronize((hipStream_t)stream));
    return SFE_OK;
}

int sfe_dsp_malloc(void **dptr, size_t bytes)
{
    if (!dptr) return SFE_EINVAL;
    SFE_HIP(hipMalloc(dptr, bytes ? bytes : 16));
    return SFE_OK;
}
#ifndef SFE_DIAG      // (the diagnostic library's own, diag/alloc.hip, also undoes the chunk-mapped ranges of its sfe_dsp_malloc_pair)
int sfe_dsp_free(void *dptr)
{
    if (dptr) SFE_HIP(hipFree(dptr));
    return SFE_OK;
}
#endif
int sfe_dsp_host_alloc(void **hptr, size_t bytes)
{
    if (!hptr) return SFE_EINVAL;
    SFE_HIP(hipHostMalloc(hptr, bytes ? bytes : 16));
    return SFE_OK;
}
int sfe_dsp_host_free(void *hptr)
{
    if (hptr) SFE_HIP(hipHostFree(hptr));
    return SFE_OK;
}
int sfe_dsp_memcpy_h2d(void *dptr, const void *hptr, size_t bytes, sfe_stream_t stream)
{
    SFE_HIP(hipMemcpyAsync(dptr, hptr, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return SFE_OK;
}
int sfe_dsp_memcpy_d2h(void *hptr, const void *dptr, size_t bytes, sfe_stream_t stream)
{
    SFE_HIP(hipMemcpyAsync(hptr, dptr, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return SFE_OK;
}
int sfe_dsp_memset(void *dptr, int value, size_t bytes, sfe_stream_t stream)
{
    SFE_HIP(hipMemsetAsync(dptr, value, bytes, (hipStream_t)stream));
    return SFE_OK;
}

struct Timer {
    hipEvent_t a, b;
};
int sfe_dsp_timer_create(sfe_timer_t *t)
{
    if (!t) return SFE_EINVAL;
    *t = nullptr;
    Timer *x = new (std::nothrow) Timer;
    if (!x) return SFE_ENOMEM;
    hipError_t e = hipEventCreate(&x->a);
    if (e != hipSuccess) {
        delete x;
        return hip_fail(e, "hipEventCreate");
    }
    e = hipEventCreate(&x->b);
    if (e != hipSuccess) {
        (void)hipEventDestroy(x->a);
        delete x;
        return hip_fail(e, "hipEventCreate");
    }
    *t = x;
    return SFE_OK;
}
int sfe_dsp_timer_start(sfe_timer_t t, sfe_stream_t stream)
{
    SFE_HIP(hipEventRecord(static_cast<Timer *>(t)->a, (hipStream_t)stream));
    return SFE_OK;
}
int sfe_dsp_timer_stop(sfe_timer_t t, sfe_stream_t stream)
{
    SFE_HIP(hipEventRecord(static_cast<Timer *>(t)->b, (hipStream_t)stream));
    return SFE_OK;
}
int sfe_dsp_timer_elapsed_ms(sfe_timer_t t, float *ms)
{
    Timer *x = static_cast<Timer *>(t);
    SFE_HIP(hipEventSynchronize(x->b));
    SFE_HIP(hipEventElapsedTime(ms, x->a, x->b));
    return SFE_OK;
}
int sfe_dsp_timer_destroy(sfe_timer_t t)
{
    Timer *x = static_cast<Timer *>(t);
    if (!x) return SFE_OK;
    (void)hipEventDestroy(x->a);
    (void)hipEventDestroy(x->b);
    delete x;
    return SFE_OK;
}

int sfe_dsp_synth_fill(void *dptr, uint64_t n_floats, uint32_t seed, uint32_t channel,
                       uint64_t first, sfe_stream_t stream)
{
    if (!dptr && n_floats) return SFE_EINVAL;
    return launch_synth_fill(static_cast<float *>(dptr), n_floats, seed, channel, first,
                             (hipStream_t)stream);
}

// ------------------------------------------------------------------------ converters
int sfe_dsp_rx_u8_to_f32(const void *d_bytes, void *d_floats, size_t n_bytes, sfe_stream_t stream)
{
    if (n_bytes && (!d_bytes || !d_floats)) return SFE_EINVAL;
    return launch_rx_u8_to_f32(static_cast<const uint8_t *>(d_bytes), static_cast<float *>(d_floats),
                               n_bytes, (hipStream_t)stream);
}

int sfe_dsp_tx_f32_to_10bit(const void *d_floats, void *d_bytes, size_t n_floats, sfe_stream_t stream)
{
    if (n_floats && (!d_bytes || !d_floats)) return SFE_EINVAL;
    return launch_tx_f32_to_10bit(static_cast<const float *>(d_floats), static_cast<uint8_t *>(d_bytes),
                                  n_floats, (hipStream_t)stream);
}

}  // extern "C"
