// blocks.h -- GNU-Radio-shaped blocks over libsfe_dsp.
//
// Same shape as gr-simplefe's public blocks (gr-simplefe/include/simplefe/source_c.h:36-49 and
// lib/source_c_impl.h): a public class that inherits VIRTUALLY from the runtime's block type and
// exposes only `typedef <runtime shared_ptr> sptr` and `static sptr make(...)`; a private _impl
// class with the constructor and work().  Items are gr_complex / float
// (gr-simplefe/lib/source_c_impl.cc:44-46, sink_f_impl.cc:44-46).  gr-simplefe's blocks are the
// hardware endpoints of a flowgraph; these are the filters that sit between them:
//     simplefe::source_c -> gr::sfe::fir_ccf / decimate_ccf / rational_resampler_ccf -> simplefe::sink_c
//
// All of them batch: the scheduler's few-thousand-item calls are collected into GPU-sized pinned
// batches and up to four batches are in flight on three streams (sfe_dsp_fir_pipe_* /
// sfe_dsp_rs_pipe_*), so a call costs a memcpy, not a launch + two PCIe round trips.  They are
// gr::block's (general_work): an item may come out a few calls after the inputs it depends on went in.
// What the stream equals: the reference class fed the same samples in calls of the batch's size.  For
// the FIR blocks and for integer-valued resampling steps (decimate_xxf, rational_resampler_xxf at any
// interp/decim: the step is `decim`) that does not depend on where the calls are cut -- bit for bit
// the reference whatever the scheduler does.  A source SLOWER than the block starves the pipe: the
// scheduler then calls with no input, the partly filled batch is sent on its way (whole blksize
// multiples first, sfe_dsp.h), and batching degenerates towards one round trip per call -- correct,
// but no faster than the _sync blocks; size batch_items to the source's rate.
// fir_ccf_sync / fir_fff_sync keep the one-round-trip-per-call sync_block form.
#ifndef GR_SFE_BLOCKS_H_
#define GR_SFE_BLOCKS_H_

#include <string.h>

#include <stdexcept>
#include <string>
#include <vector>

#include "../sfe_dsp.h"
#include "gr_compat.h"

namespace gr {
namespace sfe {

inline void check(int rc, const char *where)
{
    if (rc != SFE_OK) throw std::runtime_error(std::string(where) + ": " + sfe_dsp_last_error());
}

// ------------------------------------------------------------------------------- FIR, batched
// y[n] = sum_k taps[k] x[n-k] (the blkconv law): gr_complex items (fir_ccf) or float items
// (fir_fff, libdsp's own case: examples/bpsk/bpsk.cxx:125 pulse shaping), float taps.
template <bool CPLX>
class fir_xxf : virtual public gr::block
{
public:
    typedef typename sptr_of<fir_xxf>::type sptr;
    // batch_items: items per GPU launch (0 = 262144); device: HIP device ordinal
    static sptr make(const std::vector<float> &taps, int batch_items = 0, int device = 0);
};

template <bool CPLX>
class fir_xxf_impl : public fir_xxf<CPLX>
{
public:
    fir_xxf_impl(const std::vector<float> &taps, int batch_items, int device)
        : gr::block(CPLX ? "sfe_fir_ccf" : "sfe_fir_fff", gr::io_signature::make(1, 1, CPLX ? sizeof(gr_complex) : sizeof(float)),
                    gr::io_signature::make(1, 1, CPLX ? sizeof(gr_complex) : sizeof(float))),
          d_h(0), d_p(0)
    {
        check(sfe_dsp_fir_create(taps.data(), (int)taps.size(), 0, CPLX ? 1 : 0, 1, 0, device, &d_h), "fir_xxf");
        int rc = sfe_dsp_fir_pipe_create(d_h, (size_t)(batch_items > 0 ? batch_items : 0), &d_p);
        if (rc != SFE_OK) {
            sfe_dsp_fir_destroy(d_h);
            check(rc, "fir_xxf pipe");
        }
        this->set_relative_rate(1.0);
    }
    ~fir_xxf_impl()
    {
        sfe_dsp_pipe_destroy(d_p);
        sfe_dsp_fir_destroy(d_h);
    }

    // with items in flight the block can produce without new input (drain at the end of a stream)
    void forecast(int, gr_vector_int &req)
    {
        size_t pend = 0;
        sfe_dsp_pipe_pending(d_p, &pend);
        for (size_t i = 0; i < req.size(); i++) req[i] = pend ? 0 : 1;
    }

    int general_work(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items,
                     gr_vector_void_star &output_items)
    {
        size_t taken = 0, got = 0;
        if (ninput_items[0] > 0) check(sfe_dsp_pipe_push(d_p, input_items[0], (size_t)ninput_items[0], &taken), "fir_xxf::push");
        check(sfe_dsp_pipe_pull(d_p, output_items[0], (size_t)noutput_items, 0, &got), "fir_xxf::pull");
        if (taken == 0 && got == 0) {
            // no progress possible without waiting: every batch is in flight (block for the oldest),
            // or the upstream has nothing for us right now (send the partial batch on its way)
            check(sfe_dsp_pipe_pull(d_p, output_items[0], (size_t)noutput_items, ninput_items[0] > 0 ? 1 : 2, &got), "fir_xxf::pull");
        }
        this->consume_each((int)taken);
        return (int)got;
    }

private:
    sfe_fir_t d_h;
    sfe_pipe_t d_p;
};

template <bool CPLX>
typename fir_xxf<CPLX>::sptr fir_xxf<CPLX>::make(const std::vector<float> &taps, int batch_items, int device)
{
    return typename fir_xxf<CPLX>::sptr(new fir_xxf_impl<CPLX>(taps, batch_items, device));
}
typedef fir_xxf<true> fir_ccf;
typedef fir_xxf<false> fir_fff;

// --------------------------------------------------------------- FIR, one round trip per call
template <bool CPLX>
class fir_xxf_sync : virtual public gr::sync_block
{
public:
    typedef typename sptr_of<fir_xxf_sync>::type sptr;
    static sptr make(const std::vector<float> &taps, int device = 0);
};

template <bool CPLX>
class fir_xxf_sync_impl : public fir_xxf_sync<CPLX>
{
public:
    fir_xxf_sync_impl(const std::vector<float> &taps, int device)
        : gr::sync_block(CPLX ? "sfe_fir_ccf_sync" : "sfe_fir_fff_sync",
                         gr::io_signature::make(1, 1, CPLX ? sizeof(gr_complex) : sizeof(float)),
                         gr::io_signature::make(1, 1, CPLX ? sizeof(gr_complex) : sizeof(float))), d_h(0)
    {
        check(sfe_dsp_fir_create(taps.data(), (int)taps.size(), 0, CPLX ? 1 : 0, 1, 0, device, &d_h), "fir_xxf_sync");
    }
    ~fir_xxf_sync_impl() { sfe_dsp_fir_destroy(d_h); }

    int work(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &output_items)
    {
        check(sfe_dsp_fir_process_host(d_h, input_items[0], output_items[0], (size_t)noutput_items), "fir_xxf_sync::work");
        return noutput_items;
    }

private:
    sfe_fir_t d_h;
};

template <bool CPLX>
typename fir_xxf_sync<CPLX>::sptr fir_xxf_sync<CPLX>::make(const std::vector<float> &taps, int device)
{
    return typename fir_xxf_sync<CPLX>::sptr(new fir_xxf_sync_impl<CPLX>(taps, device));
}
typedef fir_xxf_sync<true> fir_ccf_sync;
typedef fir_xxf_sync<false> fir_fff_sync;

// --------------------------------------- a bank of FIR streams over several GPUs, one round trip per call
// n_channels independent gr_complex streams through one filter, ONE INPUT AND ONE OUTPUT PORT PER CHANNEL --
// what n_channels reference objects are (libdsp/blkconv.h:35-62) -- with the channels cut into contiguous
// blocks over `devices` inside the library (sfe_dsp_fir_group_*, sfe_dsp.h): every device gets its copy-in,
// its launch and its copy-out on a stream of its own before the call waits for any of them.
// devices = {0} is the one-GPU bank; {0, 1, ..., 7} spreads 64 channels eight to a GPU.
class fir_bank_ccf_sync : virtual public gr::sync_block
{
public:
    typedef sptr_of<fir_bank_ccf_sync>::type sptr;
    // max_items: the largest work() call served in one piece (longer calls are processed in pieces)
    static sptr make(const std::vector<float> &taps, int n_channels, const std::vector<int> &devices, int max_items = 65536);
};

class fir_bank_ccf_sync_impl : public fir_bank_ccf_sync
{
public:
    fir_bank_ccf_sync_impl(const std::vector<float> &taps, int n_channels, const std::vector<int> &devices, int max_items)
        : gr::sync_block("sfe_fir_bank_ccf_sync", gr::io_signature::make(n_channels, n_channels, sizeof(gr_complex)),
                         gr::io_signature::make(n_channels, n_channels, sizeof(gr_complex))),
          d_g(0), d_max(max_items > 0 ? max_items : 65536)
    {
        // (a check() that throws part-way leaves a half-built object whose destructor never runs -- everything
        // acquired so far is released here and the caller's current device put back before the exception goes on)
        int home = 0;
        sfe_dsp_get_device(&home);
        try {
            check(sfe_dsp_fir_group_create(taps.data(), (int)taps.size(), 0, 1, 0, n_channels, devices.data(), (int)devices.size(), &d_g),
                  "fir_bank_ccf_sync");
            int n = 0;
            check(sfe_dsp_fir_group_shards(d_g, &n), "fir_bank_ccf_sync");
            d_sh.resize(n);
            d_in.assign(n, (void *)0);
            d_out.assign(n, (void *)0);
            for (int k = 0; k < n; k++) {
                part &s = d_sh[k];
                check(sfe_dsp_fir_group_shard(d_g, k, &s.device, &s.first, &s.count, 0, &s.stream), "fir_bank_ccf_sync");
                const size_t bytes = (size_t)s.count * d_max * sizeof(gr_complex);
                check(sfe_dsp_set_device(s.device), "fir_bank_ccf_sync");
                check(sfe_dsp_malloc(&d_in[k], bytes), "fir_bank_ccf_sync");
                check(sfe_dsp_malloc(&d_out[k], bytes), "fir_bank_ccf_sync");
                check(sfe_dsp_host_alloc(&s.h_in, bytes), "fir_bank_ccf_sync");
                check(sfe_dsp_host_alloc(&s.h_out, bytes), "fir_bank_ccf_sync");
            }
        } catch (...) {
            release();
            sfe_dsp_set_device(home);
            throw;
        }
        sfe_dsp_set_device(home);
    }
    ~fir_bank_ccf_sync_impl() { release(); }

    int work(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &output_items)
    {
        for (int off = 0; off < noutput_items;) {
            const size_t m = (size_t)(noutput_items - off < d_max ? noutput_items - off : d_max), row = m * sizeof(gr_complex);
            for (size_t k = 0; k < d_sh.size(); k++) {             // copy-in of every block, each on its block's stream
                part &s = d_sh[k];
                for (int c = 0; c < s.count; c++)
                    memcpy((char *)s.h_in + c * row, (const gr_complex *)input_items[s.first + c] + off, row);
                check(sfe_dsp_memcpy_h2d(d_in[k], s.h_in, s.count * row, s.stream), "fir_bank_ccf_sync::work");
            }
            check(sfe_dsp_fir_group_process_stream(d_g, (const void *const *)d_in.data(), d_out.data(), m, m, m), "fir_bank_ccf_sync::work");
            for (size_t k = 0; k < d_sh.size(); k++)
                check(sfe_dsp_memcpy_d2h(d_sh[k].h_out, d_out[k], d_sh[k].count * row, d_sh[k].stream), "fir_bank_ccf_sync::work");
            check(sfe_dsp_fir_group_sync(d_g), "fir_bank_ccf_sync::work");
            for (size_t k = 0; k < d_sh.size(); k++)
                for (int c = 0; c < d_sh[k].count; c++)
                    memcpy((gr_complex *)output_items[d_sh[k].first + c] + off, (const char *)d_sh[k].h_out + c * row, row);
            off += (int)m;
        }
        return noutput_items;
    }

private:
    struct part {
        int device, first, count;
        sfe_stream_t stream;
        void *h_in, *h_out;
        part() : device(0), first(0), count(0), stream(0), h_in(0), h_out(0) {}
    };
    // gives back whatever has been acquired (each device's memory with that device current); safe on a half-built object
    void release()
    {
        int home = 0;
        sfe_dsp_get_device(&home);
        if (d_g) sfe_dsp_fir_group_sync(d_g);
        for (size_t k = 0; k < d_sh.size(); k++) {
            sfe_dsp_set_device(d_sh[k].device);
            if (k < d_in.size() && d_in[k]) sfe_dsp_free(d_in[k]);
            if (k < d_out.size() && d_out[k]) sfe_dsp_free(d_out[k]);
            if (d_sh[k].h_in) sfe_dsp_host_free(d_sh[k].h_in);
            if (d_sh[k].h_out) sfe_dsp_host_free(d_sh[k].h_out);
        }
        d_sh.clear();
        d_in.clear();
        d_out.clear();
        if (d_g) sfe_dsp_fir_group_destroy(d_g);
        d_g = 0;
        sfe_dsp_set_device(home);
    }
    sfe_fir_group_t d_g;
    int d_max;
    std::vector<part> d_sh;
    std::vector<void *> d_in, d_out;
};

inline fir_bank_ccf_sync::sptr fir_bank_ccf_sync::make(const std::vector<float> &taps, int n_channels, const std::vector<int> &devices,
                                                       int max_items)
{
    return fir_bank_ccf_sync::sptr(new fir_bank_ccf_sync_impl(taps, n_channels, devices, max_items));
}

// ------------------------------------------------------- decimation / rational resampling, batched
// One pipe-backed block serves both classes: `decimate` at integer rate D with upsample 1
// (libdsp/decimate.cxx:69-129) and `resample` at rate decim/interp with upsample = interp
// (libdsp/resample.cxx:85-153).  gr::block's (general_work): the scheduler's calls are collected into
// pinned batches of whole blksize-sample reference calls, four batches in flight
// (sfe_dsp_rs_pipe_create); exact mode, so the items are the reference classes' bit for bit.
template <bool CPLX>
class rs_block_impl_base
{
protected:
    rs_block_impl_base() : d_h(0), d_p(0) {}
    ~rs_block_impl_base()
    {
        if (d_p) sfe_dsp_pipe_destroy(d_p);
        if (d_h) sfe_dsp_rs_destroy(d_h);
    }
    void open(const std::vector<float> &taps, int upsample, int mode, float rate, int batch_items, int device, const char *what)
    {
        check(sfe_dsp_rs_create(taps.data(), (int)taps.size(), upsample, 4096, CPLX ? 1 : 0, 1, device, mode, &d_h), what);
        check(sfe_dsp_rs_set_exact(d_h, 1), what);
        check(sfe_dsp_rs_pipe_create(d_h, (size_t)(batch_items > 0 ? batch_items : 0), rate, &d_p), what);
    }
    // push what came in, pull what is finished; without progress wait for the oldest batch, or -- when
    // the upstream has nothing for us -- send the partial batch on its way (end of stream / slow source)
    int pump(int noutput_items, int ninput, const void *in, void *out, int *consumed, const char *what)
    {
        size_t taken = 0, got = 0;
        if (ninput > 0) check(sfe_dsp_pipe_push(d_p, in, (size_t)ninput, &taken), what);
        check(sfe_dsp_pipe_pull(d_p, out, (size_t)noutput_items, 0, &got), what);
        if (taken == 0 && got == 0) check(sfe_dsp_pipe_pull(d_p, out, (size_t)noutput_items, ninput > 0 ? 1 : 2, &got), what);
        *consumed = (int)taken;
        return (int)got;
    }
    bool pending() const
    {
        size_t n = 0;
        sfe_dsp_pipe_pending(d_p, &n);
        return n != 0;
    }
    sfe_rs_t d_h;
    sfe_pipe_t d_p;
};

// Integer decimation by D with an anti-alias FIR.  CPLX: gr_complex items (decimate_ccf) or float
// items (decimate_fff).
template <bool CPLX>
class decimate_xxf : virtual public gr::block
{
public:
    typedef typename sptr_of<decimate_xxf>::type sptr;
    static sptr make(const std::vector<float> &taps, unsigned decimation, int batch_items = 0, int device = 0);
    virtual unsigned decimation() const = 0;
};

template <bool CPLX>
class decimate_xxf_impl : public decimate_xxf<CPLX>, private rs_block_impl_base<CPLX>
{
public:
    decimate_xxf_impl(const std::vector<float> &taps, unsigned d, int batch_items, int device)
        : gr::block(CPLX ? "sfe_decimate_ccf" : "sfe_decimate_fff",
                    gr::io_signature::make(1, 1, CPLX ? sizeof(gr_complex) : sizeof(float)),
                    gr::io_signature::make(1, 1, CPLX ? sizeof(gr_complex) : sizeof(float))), d_D(d)
    {
        this->set_relative_rate(1.0 / d);
        this->open(taps, 1, SFE_RS_DECIMATE, (float)d, batch_items, device, "decimate");
    }
    unsigned decimation() const { return d_D; }

    void forecast(int noutput_items, gr_vector_int &req)
    {
        for (size_t i = 0; i < req.size(); i++) req[i] = this->pending() ? 0 : (noutput_items > 0 ? 1 : 0);
    }
    int general_work(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items,
                     gr_vector_void_star &output_items)
    {
        int consumed = 0;
        const int got = this->pump(noutput_items, ninput_items[0], input_items[0], output_items[0], &consumed, "decimate::general_work");
        this->consume_each(consumed);
        return got;
    }

private:
    unsigned d_D;
};

template <bool CPLX>
typename decimate_xxf<CPLX>::sptr decimate_xxf<CPLX>::make(const std::vector<float> &taps, unsigned decimation, int batch_items, int device)
{
    return typename decimate_xxf<CPLX>::sptr(new decimate_xxf_impl<CPLX>(taps, decimation, batch_items, device));
}
typedef decimate_xxf<true> decimate_ccf;
typedef decimate_xxf<false> decimate_fff;

// `interp` outputs per `decim` inputs through a prototype designed at the upsampled rate.
template <bool CPLX>
class rational_resampler_xxf : virtual public gr::block
{
public:
    typedef typename sptr_of<rational_resampler_xxf>::type sptr;
    static sptr make(unsigned interp, unsigned decim, const std::vector<float> &taps, int batch_items = 0, int device = 0);
};

template <bool CPLX>
class rational_resampler_xxf_impl : public rational_resampler_xxf<CPLX>, private rs_block_impl_base<CPLX>
{
public:
    rational_resampler_xxf_impl(unsigned interp, unsigned decim, const std::vector<float> &taps, int batch_items, int device)
        : gr::block(CPLX ? "sfe_rational_resampler_ccf" : "sfe_rational_resampler_fff",
                    gr::io_signature::make(1, 1, CPLX ? sizeof(gr_complex) : sizeof(float)),
                    gr::io_signature::make(1, 1, CPLX ? sizeof(gr_complex) : sizeof(float)))
    {
        this->set_relative_rate((double)interp / decim);
        this->open(taps, (int)interp, SFE_RS_RESAMPLE, (float)decim / (float)interp, batch_items, device, "rational_resampler");
    }

    void forecast(int noutput_items, gr_vector_int &req)
    {
        for (size_t i = 0; i < req.size(); i++) req[i] = this->pending() ? 0 : (noutput_items > 0 ? 1 : 0);
    }
    int general_work(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items,
                     gr_vector_void_star &output_items)
    {
        int consumed = 0;
        const int got = this->pump(noutput_items, ninput_items[0], input_items[0], output_items[0], &consumed,
                                   "rational_resampler::general_work");
        this->consume_each(consumed);
        return got;
    }
};

template <bool CPLX>
typename rational_resampler_xxf<CPLX>::sptr rational_resampler_xxf<CPLX>::make(unsigned interp, unsigned decim, const std::vector<float> &taps,
                                                                               int batch_items, int device)
{
    return typename rational_resampler_xxf<CPLX>::sptr(new rational_resampler_xxf_impl<CPLX>(interp, decim, taps, batch_items, device));
}
typedef rational_resampler_xxf<true> rational_resampler_ccf;
typedef rational_resampler_xxf<false> rational_resampler_fff;

// ---------------------------------------------------------- FIR between the wire formats, batched
// gr-simplefe's own blocks ARE the converters: source_c turns the device's u8 offset-binary (I,Q)
// bytes into gr_complex (lib/source_c_impl.cc:121-132), sink_c turns gr_complex into the 10-bit
// packed transmit format, 2 samples in 5 bytes (lib/sink_c_impl.cc:118-144).  These blocks take /
// hand out those byte streams directly and run the conversion inside the FIR kernel's load and store
// (SURVEY.md 8(f) N2 at the block level): a receive chain feeds the device's bytes straight in
// (2 instead of 8 bytes per sample over PCIe and from HBM), a transmit chain gets the bytes the USB
// writer sends.  Items:  IN_U8  -> one item = one (I,Q) byte pair (item size 2; real: 1 byte)
//                        OUT_TX10 -> unsigned char items, in whole 5-byte groups (output multiple 5)
//   rx_fir_bc  = u8 pairs in, gr_complex out        fir_tx_cb = gr_complex in, 10-bit bytes out
//   rx_fir_tx_bb = wire to wire (the source_c -> FIR -> sink_c flowgraph as one block)
template <bool CPLX, bool IN_U8, bool OUT_TX10>
class fir_wire : virtual public gr::block
{
public:
    typedef typename sptr_of<fir_wire>::type sptr;
    static sptr make(const std::vector<float> &taps, int batch_items = 0, int device = 0);
};

template <bool CPLX, bool IN_U8, bool OUT_TX10>
class fir_wire_impl : public fir_wire<CPLX, IN_U8, OUT_TX10>
{
    enum { GS = CPLX ? 2 : 4,                                                   // samples per 10-bit group
           IN_SZ = IN_U8 ? (CPLX ? 2 : 1) : (CPLX ? (int)sizeof(gr_complex) : (int)sizeof(float)),
           OUT_SZ = OUT_TX10 ? 1 : (CPLX ? (int)sizeof(gr_complex) : (int)sizeof(float)) };

public:
    fir_wire_impl(const std::vector<float> &taps, int batch_items, int device)
        : gr::block("sfe_fir_wire", gr::io_signature::make(1, 1, IN_SZ), gr::io_signature::make(1, 1, OUT_SZ)), d_h(0), d_p(0)
    {
        check(sfe_dsp_fir_create(taps.data(), (int)taps.size(), 0, CPLX ? 1 : 0, 1, 0, device, &d_h), "fir_wire");
        int rc = IN_U8 ? sfe_dsp_fir_set_input_format(d_h, SFE_FMT_U8) : SFE_OK;
        if (rc == SFE_OK && OUT_TX10) rc = sfe_dsp_fir_set_output_format(d_h, SFE_FMT_TX10);
        if (rc == SFE_OK) rc = sfe_dsp_fir_pipe_create(d_h, (size_t)(batch_items > 0 ? batch_items : 0), &d_p);
        if (rc != SFE_OK) {
            sfe_dsp_fir_destroy(d_h);
            check(rc, "fir_wire setup");
        }
        if (OUT_TX10) {
            this->set_output_multiple(5);
            this->set_relative_rate(5.0 / GS);
        } else
            this->set_relative_rate(1.0);
    }
    ~fir_wire_impl()
    {
        sfe_dsp_pipe_destroy(d_p);
        sfe_dsp_fir_destroy(d_h);
    }
    void forecast(int, gr_vector_int &req)
    {
        size_t pend = 0;
        sfe_dsp_pipe_pending(d_p, &pend);
        for (size_t i = 0; i < req.size(); i++) req[i] = pend ? 0 : 1;
    }
    int general_work(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items,
                     gr_vector_void_star &output_items)
    {
        const size_t room = OUT_TX10 ? (size_t)noutput_items / 5 : (size_t)noutput_items;      // pipe items: 5-byte groups / samples
        size_t taken = 0, got = 0;
        if (ninput_items[0] > 0) check(sfe_dsp_pipe_push(d_p, input_items[0], (size_t)ninput_items[0], &taken), "fir_wire::push");
        check(sfe_dsp_pipe_pull(d_p, output_items[0], room, 0, &got), "fir_wire::pull");
        if (taken == 0 && got == 0) check(sfe_dsp_pipe_pull(d_p, output_items[0], room, ninput_items[0] > 0 ? 1 : 2, &got), "fir_wire::pull");
        this->consume_each((int)taken);
        return (int)(OUT_TX10 ? got * 5 : got);
    }

private:
    sfe_fir_t d_h;
    sfe_pipe_t d_p;
};

template <bool CPLX, bool IN_U8, bool OUT_TX10>
typename fir_wire<CPLX, IN_U8, OUT_TX10>::sptr fir_wire<CPLX, IN_U8, OUT_TX10>::make(const std::vector<float> &taps, int batch_items, int device)
{
    return typename fir_wire<CPLX, IN_U8, OUT_TX10>::sptr(new fir_wire_impl<CPLX, IN_U8, OUT_TX10>(taps, batch_items, device));
}
typedef fir_wire<true, true, false> rx_fir_bc;        // source_c's bytes in, gr_complex out
typedef fir_wire<true, false, true> fir_tx_cb;        // gr_complex in, sink_c's bytes out
typedef fir_wire<true, true, true> rx_fir_tx_bb;      // wire to wire
typedef fir_wire<false, true, false> rx_fir_bf;       // source_f's bytes in, float out
typedef fir_wire<false, false, true> fir_tx_fb;       // float in, sink_f's bytes out

}  // namespace sfe
}  // namespace gr

#endif
