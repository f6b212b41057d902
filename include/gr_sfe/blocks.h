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
// fir_ccf / fir_fff batch: the scheduler's few-thousand-item calls are collected into GPU-sized
// pinned batches and up to four batches are in flight on three streams (sfe_dsp_fir_pipe_*), so a
// call costs a memcpy, not a launch + two PCIe round trips.  They are gr::block's (general_work):
// item k out is the filter's output for item k in, but it may come out a few calls later.
// fir_ccf_sync / fir_fff_sync keep the one-round-trip-per-call sync_block form.
#ifndef GR_SFE_BLOCKS_H_
#define GR_SFE_BLOCKS_H_

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

// ------------------------------------------------------------------------------- decimation
// Integer decimation by D with an anti-alias FIR: the `decimate` class at rate D, upsample 1
// (libdsp/decimate.cxx:69-129).  A sync_decimator: work() consumes D*noutput_items inputs.
// CPLX: gr_complex items (decimate_ccf) or float items (decimate_fff).
template <bool CPLX>
class decimate_xxf : virtual public gr::sync_decimator
{
public:
    typedef typename sptr_of<decimate_xxf>::type sptr;
    static sptr make(const std::vector<float> &taps, unsigned decimation, int max_items = 1 << 16, int device = 0);
};

template <bool CPLX>
class decimate_xxf_impl : public decimate_xxf<CPLX>
{
public:
    decimate_xxf_impl(const std::vector<float> &taps, unsigned d, int max_items, int device)
        : gr::sync_decimator(CPLX ? "sfe_decimate_ccf" : "sfe_decimate_fff",
                             gr::io_signature::make(1, 1, CPLX ? sizeof(gr_complex) : sizeof(float)),
                             gr::io_signature::make(1, 1, CPLX ? sizeof(gr_complex) : sizeof(float)), d),
          d_h(0), d_blk(max_items), d_D(d)
    {
        check(sfe_dsp_rs_create(taps.data(), (int)taps.size(), 1, max_items, CPLX ? 1 : 0, 1, device, SFE_RS_DECIMATE, &d_h), "decimate");
    }
    ~decimate_xxf_impl() { sfe_dsp_rs_destroy(d_h); }

    int work(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &output_items)
    {
        const size_t W = CPLX ? 2 : 1;          // floats per item
        const float *in = static_cast<const float *>(input_items[0]);
        float *out = static_cast<float *>(output_items[0]);
        int produced = 0, n_in = noutput_items * (int)d_D;
        // the class call takes at most blksize inputs (decimate.cxx:79-82): feed it in pieces
        for (int off = 0; off < n_in; off += d_blk) {
            const int m = n_in - off < d_blk ? n_in - off : d_blk;
            int n_out = 0;
            check(sfe_dsp_rs_process(d_h, in + W * (size_t)off, m, out + W * (size_t)produced, noutput_items - produced + 1,
                                     (float)d_D, &n_out), "decimate::work");
            produced += n_out;
        }
        return produced;
    }

private:
    sfe_rs_t d_h;
    int d_blk;
    unsigned d_D;
};

template <bool CPLX>
typename decimate_xxf<CPLX>::sptr decimate_xxf<CPLX>::make(const std::vector<float> &taps, unsigned decimation, int max_items, int device)
{
    return typename decimate_xxf<CPLX>::sptr(new decimate_xxf_impl<CPLX>(taps, decimation, max_items, device));
}
typedef decimate_xxf<true> decimate_ccf;
typedef decimate_xxf<false> decimate_fff;

// ------------------------------------------------------------------------ rational resampler
// `interp` outputs per `decim` inputs through a prototype designed at the upsampled rate -- the
// `resample` class with upsample = interp, rate = decim/interp (libdsp/resample.cxx:85-153).
// A general block: general_work() consumes what it is given.
template <bool CPLX>
class rational_resampler_xxf : virtual public gr::block
{
public:
    typedef typename sptr_of<rational_resampler_xxf>::type sptr;
    static sptr make(unsigned interp, unsigned decim, const std::vector<float> &taps, int max_items = 1 << 16, int device = 0);
};

template <bool CPLX>
class rational_resampler_xxf_impl : public rational_resampler_xxf<CPLX>
{
public:
    rational_resampler_xxf_impl(unsigned interp, unsigned decim, const std::vector<float> &taps, int max_items, int device)
        : gr::block(CPLX ? "sfe_rational_resampler_ccf" : "sfe_rational_resampler_fff",
                    gr::io_signature::make(1, 1, CPLX ? sizeof(gr_complex) : sizeof(float)),
                    gr::io_signature::make(1, 1, CPLX ? sizeof(gr_complex) : sizeof(float))),
          d_h(0), d_interp(interp), d_decim(decim), d_blk(max_items), d_rate((float)decim / (float)interp)
    {
        this->set_relative_rate((double)interp / decim);
        check(sfe_dsp_rs_create(taps.data(), (int)taps.size(), (int)interp, max_items, CPLX ? 1 : 0, 1, device, SFE_RS_RESAMPLE, &d_h),
              "rational_resampler");
    }
    ~rational_resampler_xxf_impl() { sfe_dsp_rs_destroy(d_h); }

    void forecast(int noutput_items, gr_vector_int &req)
    {
        for (size_t i = 0; i < req.size(); i++) req[i] = (int)((long long)noutput_items * d_decim / d_interp) + 1;
    }

    int general_work(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items,
                     gr_vector_void_star &output_items)
    {
        const float *in = static_cast<const float *>(input_items[0]);
        float *out = static_cast<float *>(output_items[0]);
        // take as many inputs as surely fit the output buffer
        long long can = ((long long)(noutput_items - 1) * d_decim) / d_interp;
        int n_in = ninput_items[0] < can ? ninput_items[0] : (int)can;
        if (n_in > d_blk) n_in = d_blk;
        if (n_in <= 0) { this->consume_each(0); return 0; }
        int n_out = 0;
        check(sfe_dsp_rs_process(d_h, in, n_in, out, noutput_items, d_rate, &n_out), "rational_resampler::general_work");
        this->consume_each(n_in);
        return n_out;
    }

private:
    sfe_rs_t d_h;
    unsigned d_interp, d_decim;
    int d_blk;
    float d_rate;
};

template <bool CPLX>
typename rational_resampler_xxf<CPLX>::sptr rational_resampler_xxf<CPLX>::make(unsigned interp, unsigned decim, const std::vector<float> &taps,
                                                                               int max_items, int device)
{
    return typename rational_resampler_xxf<CPLX>::sptr(new rational_resampler_xxf_impl<CPLX>(interp, decim, taps, max_items, device));
}
typedef rational_resampler_xxf<true> rational_resampler_ccf;
typedef rational_resampler_xxf<false> rational_resampler_fff;

}  // namespace sfe
}  // namespace gr

#endif
