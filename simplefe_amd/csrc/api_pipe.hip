// api_pipe.hip -- the pinned host pipe over a FIR or a resample / decimate handle (sfe_dsp_*_pipe_*).  Host code only.
#include "host.h"

using namespace sfe;

// ------------------------------------------------------------ pipelined host streaming
// A GNU Radio scheduler hands a block a few thousand items per work() call
// (gr-simplefe/lib/sink_c_impl.cc:157-174, source_c_impl.cc:134-153); one synchronous H2D ->
// kernel -> D2H round trip per call is launch/sync bound (27 us per 3841 samples).  The pipe
// collects pushed items in pinned batches and keeps up to SFE_PIPE_SLOTS batches in flight on
// three streams (copy in / filter / copy out overlap, PCIe is full duplex); pull hands out finished
// items in order.  Sample alignment is untouched: item k out is the filter's output for item k in.
#include <immintrin.h>
namespace sfe {
// Host copies into / out of the pinned batches are what bounds the pipe (the GPU side of a batch
// is ~30 us, the two copies ~60): stream them past the cache -- the pinned side is touched next by
// the DMA engine, not by this core.  Falls back to memcpy without AVX2 or for small / odd pieces.
__attribute__((target("avx2"))) static void copy_stream_avx2(char *dst, const char *src, size_t n)
{
    while (n && (reinterpret_cast<uintptr_t>(dst) & 31u)) {
        *dst++ = *src++;
        n--;
    }
    for (; n >= 128; n -= 128, dst += 128, src += 128) {
        const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src));
        const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + 32));
        const __m256i c = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + 64));
        const __m256i d = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + 96));
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst), a);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + 32), b);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + 64), c);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + 96), d);
    }
    _mm_sfence();
    if (n) memcpy(dst, src, n);
}
static void copy_stream(void *dst, const void *src, size_t n)
{
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2 && n >= 4096) copy_stream_avx2(static_cast<char *>(dst), static_cast<const char *>(src), n);
    else memcpy(dst, src, n);
}

constexpr int PIPE_SLOTS = 4;
struct FirPipe {
    uint32_t magic = 0x50495031u;   // 'PIP1'
    Fir *f = nullptr;               // the filter behind the pipe ...
    void *rs = nullptr;             // ... or the resampler / decimator (sfe_rs_t), at `rate`
    float rate = 1.0f;
    int device = 0;
    size_t batch = 0, out_cap = 0, in_e = 0, out_e = 0;     // batch: input items per slot; out_cap: output items a slot can hold
    size_t quantum = 1;             // a partly filled batch sent on its way early is cut on a multiple of this (rs pipes at a
                                    // non-integer step: blksize, so the cut falls where a reference call ends; else 1)
    size_t tx_gs = 0;               // 10-bit packed output: samples per 5-byte group (2 complex / 4 real); an output ITEM is one group.
                                    // Cuts then fall on whole groups ONLY (less than a group is never sent: the converter emits whole groups)
    struct Slot {
        char *h_in = nullptr, *h_out = nullptr;
        void *d_in = nullptr, *d_out = nullptr;
        size_t n = 0;               // items submitted in this slot
        size_t n_out = 0;           // items it produces (== n behind a filter)
        hipEvent_t ev_in = nullptr, ev_k = nullptr, ev_out = nullptr;
        bool busy = false;          // submitted and not yet fully pulled
    } slot[PIPE_SLOTS];
    hipStream_t s_in = nullptr, s_k = nullptr, s_out = nullptr;
    int head = 0;                   // slot being filled
    size_t fill = 0;                // items in it
    int tail = 0;                   // oldest busy slot
    size_t out_off = 0;             // items already pulled from it
    bool tail_ready = false;        // its ev_out has been seen complete
};

static FirPipe *as_pipe(void *h)
{
    FirPipe *p = static_cast<FirPipe *>(h);
    if (p && p->magic != 0x50495031u) {
        set_error("not a live pipe handle");
        return nullptr;
    }
    return p;
}

static void pipe_free(FirPipe *p)
{
    if (!p) return;
    p->magic = 0;
    DeviceGuard g(p->device);
    for (auto &sl : p->slot) {
        if (sl.h_in) (void)hipHostFree(sl.h_in);
        if (sl.h_out) (void)hipHostFree(sl.h_out);
        if (sl.d_in) (void)hipFree(sl.d_in);
        if (sl.d_out) (void)hipFree(sl.d_out);
        if (sl.ev_in) (void)hipEventDestroy(sl.ev_in);
        if (sl.ev_k) (void)hipEventDestroy(sl.ev_k);
        if (sl.ev_out) (void)hipEventDestroy(sl.ev_out);
    }
    if (p->s_in) (void)hipStreamDestroy(p->s_in);
    if (p->s_k) (void)hipStreamDestroy(p->s_k);
    if (p->s_out) (void)hipStreamDestroy(p->s_out);
    delete p;
}

// submits the first `count` items of the batch being filled (all of it when count == fill); what is left
// moves to the front of the next slot (free: a partial submit only happens when nothing is in flight)
static int pipe_submit(FirPipe *p, size_t count)
{
    FirPipe::Slot &sl = p->slot[p->head];
    const size_t rem = p->fill - count;
    sl.n = count;
    SFE_HIP(hipMemcpyAsync(sl.d_in, sl.h_in, sl.n * p->in_e, hipMemcpyHostToDevice, p->s_in));
    SFE_HIP(hipEventRecord(sl.ev_in, p->s_in));
    SFE_HIP(hipStreamWaitEvent(p->s_k, sl.ev_in, 0));
    // a slot's device buffers are reused PIPE_SLOTS batches later: by then its copy-out has been
    // waited for (the slot was pulled), so no further ordering is needed on s_k
    int rc;
    if (p->f) {
        rc = fir_run(p->f, sl.d_in, sl.d_out, sl.n, sl.n, sl.n, p->s_k);
        sl.n_out = p->tx_gs ? sl.n / p->tx_gs : sl.n;
    } else {
        // the output count is known on the host as soon as the launch is made (closed form for
        // integer-valued steps, the replayed float32 recurrence otherwise): the copy-out is sized by it
        rc = sfe_dsp_rs_process_stream(p->rs, sl.d_in, sl.n, sl.n, sl.d_out, p->out_cap, p->out_cap, p->rate, &sl.n_out, p->s_k);
    }
    if (rc != SFE_OK) return rc;
    SFE_HIP(hipEventRecord(sl.ev_k, p->s_k));
    SFE_HIP(hipStreamWaitEvent(p->s_out, sl.ev_k, 0));
    if (sl.n_out) SFE_HIP(hipMemcpyAsync(sl.h_out, sl.d_out, sl.n_out * p->out_e, hipMemcpyDeviceToHost, p->s_out));
    SFE_HIP(hipEventRecord(sl.ev_out, p->s_out));
    sl.busy = true;
    p->head = (p->head + 1) % PIPE_SLOTS;
    if (rem) memcpy(p->slot[p->head].h_in, sl.h_in + count * p->in_e, rem * p->in_e);
    p->fill = rem;
    return SFE_OK;
}
}  // namespace sfe

extern "C" {

static int pipe_alloc(FirPipe *p, sfe_pipe_t *out)
{
    auto fail = [&](hipError_t e, const char *what) { int rc = hip_fail(e, what); pipe_free(p); return rc; };
#define TRY(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return fail(e__, #call); } while (0)
    TRY(hipStreamCreateWithFlags(&p->s_in, hipStreamNonBlocking));
    TRY(hipStreamCreateWithFlags(&p->s_k, hipStreamNonBlocking));
    TRY(hipStreamCreateWithFlags(&p->s_out, hipStreamNonBlocking));
    for (auto &sl : p->slot) {
        TRY(hipHostMalloc((void **)&sl.h_in, p->batch * p->in_e));
        TRY(hipHostMalloc((void **)&sl.h_out, p->out_cap * p->out_e));
        TRY(hipMalloc(&sl.d_in, p->batch * p->in_e));
        TRY(hipMalloc(&sl.d_out, p->out_cap * p->out_e));
        TRY(hipEventCreateWithFlags(&sl.ev_in, hipEventDisableTiming));
        TRY(hipEventCreateWithFlags(&sl.ev_k, hipEventDisableTiming));
        TRY(hipEventCreateWithFlags(&sl.ev_out, hipEventDisableTiming));
    }
#undef TRY
    *out = p;
    return SFE_OK;
}

int sfe_dsp_fir_pipe_create(sfe_fir_t fir, size_t batch_items, sfe_pipe_t *out)
{
    if (!out) return SFE_EINVAL;
    *out = nullptr;
    Fir *f = as_fir(fir);
    if (!f || f->n_channels != 1) {
        set_error("fir_pipe_create: needs a single-channel FIR handle");
        return SFE_EINVAL;
    }
    if (batch_items == 0) batch_items = (size_t)1 << 18;
    batch_items = (batch_items + 3) & ~(size_t)3;          // whole 10-bit groups per batch, whatever the output format
    if (batch_items < 256 || batch_items > ((size_t)1 << 26)) {
        set_error("fir_pipe_create: batch of %zu items out of range (256 .. 2^26)", batch_items);
        return SFE_EINVAL;
    }
    SFE_ON_DEVICE(f->device);
    FirPipe *p = new (std::nothrow) FirPipe;
    if (!p) return SFE_ENOMEM;
    p->f = f;
    p->device = f->device;
    p->batch = p->out_cap = batch_items;
    // items in: float32 samples, or the u8 wire format when the handle converts on load (SFE_FMT_U8:
    // 2 bytes per complex item, 1 per real one -- a receive chain hands the device's bytes straight in)
    p->in_e = f->in_u8 ? (f->data_complex ? 2 : 1) : (f->data_complex ? 8 : 4);
    p->out_e = f->out_complex ? 8 : 4;
    if (f->out_tx10) {
        // the transmit wire format out (sink_c_impl.cc:118-144 / sink_f_impl.cc:117-143): an output item is one 5-byte group
        p->tx_gs = f->out_complex ? 2 : 4;
        p->quantum = p->tx_gs;
        p->out_e = 5;
        p->out_cap = p->batch / p->tx_gs;
    }
    const int rc = pipe_alloc(p, out);
    if (rc == SFE_OK) f->piped++;          // the handle's formats are frozen while the pipe lives
    return rc;
}

int sfe_dsp_rs_pipe_create(sfe_rs_t rs, size_t batch_items, float rate, sfe_pipe_t *out)
{
    if (!out) return SFE_EINVAL;
    *out = nullptr;
    Rs *r = as_rs(rs);
    if (!r || r->n_channels != 1) {
        set_error("rs_pipe_create: needs a single-channel resample/decimate handle");
        return SFE_EINVAL;
    }
    if (r->mode == SFE_RS_RESAMPLE ? (rate < 1.0 / r->U) : (rate < 1.0)) {
        set_error("rs_pipe_create: rate %g not accepted by this mode", (double)rate);
        return SFE_EINVAL;
    }
    if (batch_items == 0) batch_items = (size_t)1 << 18;
    // whole reference calls per batch: for a non-integer step the result depends on where the
    // blksize-sample calls fall (resample.cxx:85-153), and they must fall where they would without the pipe
    batch_items = (batch_items + (size_t)r->blksize - 1) / (size_t)r->blksize * (size_t)r->blksize;
    if (batch_items < 256 || batch_items > ((size_t)1 << 26)) {
        set_error("rs_pipe_create: batch of %zu items out of range (256 .. 2^26)", batch_items);
        return SFE_EINVAL;
    }
    SFE_ON_DEVICE(r->device);
    FirPipe *p = new (std::nothrow) FirPipe;
    if (!p) return SFE_ENOMEM;
    p->rs = rs;
    p->rate = rate;
    p->device = r->device;
    p->batch = batch_items;
    p->out_cap = (size_t)ceil((double)batch_items / (double)rate) + 8;
    p->out_e = (size_t)r->esz();
    p->in_e = r->in_u8 ? (r->data_complex ? 2 : 1) : p->out_e;      // u8 wire-format items in (integer-valued steps)
    {
        // an integer-valued step gives the same items wherever the calls are cut; any other step does not
        const float stepf = rate * (float)r->U;
        p->quantum = (stepf >= 1.0f && stepf == floorf(stepf)) ? 1 : (size_t)r->blksize;
    }
    const int rc = pipe_alloc(p, out);
    if (rc == SFE_OK) r->piped++;
    return rc;
}

int sfe_dsp_pipe_push(sfe_pipe_t h, const void *in, size_t n_items, size_t *n_taken)
{
    FirPipe *p = as_pipe(h);
    if (!p || !n_taken || (n_items && !in)) return SFE_EINVAL;
    *n_taken = 0;
    SFE_ON_DEVICE(p->device);
    const char *src = static_cast<const char *>(in);
    while (n_items) {
        FirPipe::Slot &sl = p->slot[p->head];
        if (sl.busy) break;                                   // every slot in flight: pull first (backpressure)
        size_t m = p->batch - p->fill;
        if (m > n_items) m = n_items;
        copy_stream(sl.h_in + p->fill * p->in_e, src, m * p->in_e);
        p->fill += m;
        src += m * p->in_e;
        n_items -= m;
        *n_taken += m;
        if (p->fill == p->batch) {
            int rc = pipe_submit(p, p->fill);
            if (rc != SFE_OK) return rc;
        }
    }
    return SFE_OK;
}

// The oldest finished items, in place: *ptr / *count describe what is left of the oldest batch in flight once its
// copy-out has completed (count 0: nothing is ready).  wait: 0 = never block, 1 = block for that batch, 2 = also send a
// partly filled batch on its way when nothing else is in flight (end of stream / drain) and block for it.
static int pipe_front(FirPipe *p, int wait, const char **ptr, size_t *count)
{
    *ptr = nullptr;
    *count = 0;
    for (;;) {
        FirPipe::Slot &sl = p->slot[p->tail];
        if (!sl.busy) {
            if (wait == 2 && p->fill > 0 && p->tail == p->head) {
                // whole reference calls first: the remainder -- less than one call -- goes out only when
                // it is all there is, as the short last call a reference caller would make
                const size_t whole = p->fill / p->quantum * p->quantum;
                if (!whole && p->tx_gs) return SFE_OK;   // less than one 10-bit group: nothing the converter would emit
                int rc = pipe_submit(p, whole ? whole : p->fill);
                if (rc != SFE_OK) return rc;
                continue;
            }
            return SFE_OK;
        }
        if (!p->tail_ready) {
            if (wait) {
                SFE_HIP(hipEventSynchronize(sl.ev_out));
            } else {
                hipError_t e = hipEventQuery(sl.ev_out);
                if (e == hipErrorNotReady) return SFE_OK;
                if (e != hipSuccess) return hip_fail(e, "hipEventQuery");
            }
            p->tail_ready = true;
        }
        if (sl.n_out == p->out_off) {        // a batch that produced nothing (a decimator fed less than one step): retire it
            sl.busy = false;
            p->tail = (p->tail + 1) % PIPE_SLOTS;
            p->out_off = 0;
            p->tail_ready = false;
            continue;
        }
        *ptr = sl.h_out + p->out_off * p->out_e;
        *count = sl.n_out - p->out_off;
        return SFE_OK;
    }
}

// `m` of the items pipe_front described have been consumed
static void pipe_advance(FirPipe *p, size_t m)
{
    FirPipe::Slot &sl = p->slot[p->tail];
    p->out_off += m;
    if (p->out_off == sl.n_out) {
        sl.busy = false;
        p->tail = (p->tail + 1) % PIPE_SLOTS;
        p->out_off = 0;
        p->tail_ready = false;
    }
}

int sfe_dsp_pipe_pull(sfe_pipe_t h, void *out, size_t max_items, int wait, size_t *n_got)
{
    FirPipe *p = as_pipe(h);
    if (!p || !n_got || (max_items && !out)) return SFE_EINVAL;
    *n_got = 0;
    SFE_ON_DEVICE(p->device);
    char *dst = static_cast<char *>(out);
    int w = wait;                        // wait == 1 blocks for the oldest batch only, wait == 2 for all of them
    while (max_items) {
        const char *src;
        size_t m;
        int rc = pipe_front(p, w, &src, &m);
        if (rc != SFE_OK) return rc;
        if (!m) break;
        if (wait == 1) w = 0;
        if (m > max_items) m = max_items;
        copy_stream(dst, src, m * p->out_e);
        dst += m * p->out_e;
        max_items -= m;
        *n_got += m;
        pipe_advance(p, m);
    }
    return SFE_OK;
}

int sfe_dsp_pipe_acquire(sfe_pipe_t h, void **buf, size_t *room_items)
{
    FirPipe *p = as_pipe(h);
    if (!p || !buf || !room_items) return SFE_EINVAL;
    *buf = nullptr;
    *room_items = 0;
    FirPipe::Slot &sl = p->slot[p->head];
    if (sl.busy) return SFE_OK;          // every batch in flight: take finished items out first
    *buf = sl.h_in + p->fill * p->in_e;
    *room_items = p->batch - p->fill;
    return SFE_OK;
}

int sfe_dsp_pipe_commit(sfe_pipe_t h, size_t n_items)
{
    FirPipe *p = as_pipe(h);
    if (!p) return SFE_EINVAL;
    if (p->slot[p->head].busy ? n_items != 0 : n_items > p->batch - p->fill) {
        set_error("pipe_commit: %zu items exceed the room the last acquire reported", n_items);
        return SFE_EINVAL;
    }
    SFE_ON_DEVICE(p->device);
    p->fill += n_items;
    if (p->fill == p->batch) return pipe_submit(p, p->fill);
    return SFE_OK;
}

int sfe_dsp_pipe_peek(sfe_pipe_t h, const void **out, size_t *n_items, int wait)
{
    FirPipe *p = as_pipe(h);
    if (!p || !out || !n_items) return SFE_EINVAL;
    SFE_ON_DEVICE(p->device);
    const char *src;
    int rc = pipe_front(p, wait, &src, n_items);
    *out = src;
    return rc;
}

int sfe_dsp_pipe_release(sfe_pipe_t h, size_t n_items)
{
    FirPipe *p = as_pipe(h);
    if (!p) return SFE_EINVAL;
    FirPipe::Slot &sl = p->slot[p->tail];
    if (n_items && (!sl.busy || !p->tail_ready || n_items > sl.n_out - p->out_off)) {
        set_error("pipe_release: %zu items exceed what the last peek reported", n_items);
        return SFE_EINVAL;
    }
    if (n_items) pipe_advance(p, n_items);
    return SFE_OK;
}

int sfe_dsp_pipe_pending(sfe_pipe_t h, size_t *items)
{
    FirPipe *p = as_pipe(h);
    if (!p || !items) return SFE_EINVAL;
    size_t n = p->tx_gs ? p->fill / p->tx_gs : p->fill;         // (10-bit output: in groups, the unit pull hands out)
    for (int i = 0; i < PIPE_SLOTS; i++)
        if (p->slot[i].busy) n += p->slot[i].n_out - (i == p->tail ? p->out_off : 0);
    *items = n;
    return SFE_OK;
}

int sfe_dsp_pipe_destroy(sfe_pipe_t h)
{
    FirPipe *p = as_pipe(h);
    if (!p) return SFE_OK;
    {
        DeviceGuard g(p->device);
        (void)hipStreamSynchronize(p->s_in);
        (void)hipStreamSynchronize(p->s_k);
        (void)hipStreamSynchronize(p->s_out);
    }
    // the handle is alive: its destroy call refuses while a pipe borrows it
    if (p->f && p->f->piped > 0) p->f->piped--;
    if (p->rs && static_cast<Rs *>(p->rs)->piped > 0) static_cast<Rs *>(p->rs)->piped--;
    pipe_free(p);
    return SFE_OK;
}

}  // extern "C"
