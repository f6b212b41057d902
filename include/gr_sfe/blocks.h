// blocks.h -- GNU-Radio-shaped blocks over libsfe_dsp.
//
// Same shape as gr-simplefe's public blocks: a `make(...)` factory returning an sptr
// (gr-simplefe/include/simplefe/source_c.h:36-49) and a work() that moves gr_complex / float items
// (gr-simplefe/lib/source_c_impl.cc:44-46, sink_f_impl.cc:44-46).  gr-simplefe's blocks are the
// hardware endpoints of a flowgraph; these are the filters that sit between them:
//     simplefe::source_c -> gr::sfe::fir_ccf / decimate_ccf / rational_resampler_ccf -> simplefe::sink_c
// Each work() call is one synchronous host round trip (H2D, kernel, D2H); the scheduler's
// buffer sizes (a few thousand items) make that PCIe/launch bound -- see DESIGN.md section 5.
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

// y[n] = sum_k taps[k] x[n-k]: complex in, complex out, float taps (the blkconv law).
class fir_ccf : public gr::sync_block
{
public:
    typedef std::shared_ptr<fir_ccf> sptr;
    static sptr make(const std::vector<float> &taps, int device = 0) { return sptr(new fir_ccf(taps, device)); }
    ~fir_ccf() { sfe_dsp_fir_destroy(d_h); }

    int work(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &output_items)
    {
        check(sfe_dsp_fir_process_host(d_h, input_items[0], output_items[0], (size_t)noutput_items), "fir_ccf::work");
        return noutput_items;
    }

private:
    fir_ccf(const std::vector<float> &taps, int device)
        : gr::sync_block("sfe_fir_ccf", gr::io_signature::make(1, 1, sizeof(gr_complex)),
                         gr::io_signature::make(1, 1, sizeof(gr_complex))), d_h(0)
    {
        check(sfe_dsp_fir_create(taps.data(), (int)taps.size(), 0, 1, 1, 0, device, &d_h), "fir_ccf");
    }
    sfe_fir_t d_h;
};

// float in, float out (libdsp's own case: examples/bpsk/bpsk.cxx:125 pulse shaping)
class fir_fff : public gr::sync_block
{
public:
    typedef std::shared_ptr<fir_fff> sptr;
    static sptr make(const std::vector<float> &taps, int device = 0) { return sptr(new fir_fff(taps, device)); }
    ~fir_fff() { sfe_dsp_fir_destroy(d_h); }

    int work(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &output_items)
    {
        check(sfe_dsp_fir_process_host(d_h, input_items[0], output_items[0], (size_t)noutput_items), "fir_fff::work");
        return noutput_items;
    }

private:
    fir_fff(const std::vector<float> &taps, int device)
        : gr::sync_block("sfe_fir_fff", gr::io_signature::make(1, 1, sizeof(float)), gr::io_signature::make(1, 1, sizeof(float))),
          d_h(0)
    {
        check(sfe_dsp_fir_create(taps.data(), (int)taps.size(), 0, 0, 1, 0, device, &d_h), "fir_fff");
    }
    sfe_fir_t d_h;
};

// Integer decimation by D with an anti-alias FIR: the `decimate` class at rate D, upsample 1
// (libdsp/decimate.cxx:69-129).  A sync_decimator: work() consumes D*noutput_items inputs.
// CPLX: gr_complex items (decimate_ccf) or float items (decimate_fff).
template <bool CPLX>
class decimate_xxf : public gr::sync_decimator
{
public:
    typedef std::shared_ptr<decimate_xxf> sptr;
    static sptr make(const std::vector<float> &taps, unsigned decimation, int max_items = 1 << 16, int device = 0)
    {
        return sptr(new decimate_xxf(taps, decimation, max_items, device));
    }
    ~decimate_xxf() { sfe_dsp_rs_destroy(d_h); }

    int work(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &output_items)
    {
        constexpr size_t W = CPLX ? 2 : 1;          // floats per item
        const float *in = static_cast<const float *>(input_items[0]);
        float *out = static_cast<float *>(output_items[0]);
        int produced = 0, n_in = noutput_items * (int)decimation();
        // the class call takes at most blksize inputs (decimate.cxx:79-82): feed it in pieces
        for (int off = 0; off < n_in; off += d_blk) {
            const int m = n_in - off < d_blk ? n_in - off : d_blk;
            int n_out = 0;
            check(sfe_dsp_rs_process(d_h, in + W * (size_t)off, m, out + W * (size_t)produced, noutput_items - produced + 1,
                                     (float)decimation(), &n_out), "decimate::work");
            produced += n_out;
        }
        return produced;
    }

private:
    decimate_xxf(const std::vector<float> &taps, unsigned d, int max_items, int device)
        : gr::sync_decimator(CPLX ? "sfe_decimate_ccf" : "sfe_decimate_fff",
                             gr::io_signature::make(1, 1, CPLX ? sizeof(gr_complex) : sizeof(float)),
                             gr::io_signature::make(1, 1, CPLX ? sizeof(gr_complex) : sizeof(float)), d), d_h(0), d_blk(max_items)
    {
        check(sfe_dsp_rs_create(taps.data(), (int)taps.size(), 1, max_items, CPLX ? 1 : 0, 1, device, SFE_RS_DECIMATE, &d_h), "decimate");
    }
    sfe_rs_t d_h;
    int d_blk;
};
typedef decimate_xxf<true> decimate_ccf;
typedef decimate_xxf<false> decimate_fff;

// Rational resampler: `interp` outputs per `decim` inputs through a prototype designed at the
// upsampled rate -- the `resample` class with upsample = interp, rate = decim/interp
// (libdsp/resample.cxx:85-153).  A general block: general_work() consumes what it is given.
template <bool CPLX>
class rational_resampler_xxf : public gr::block
{
public:
    typedef std::shared_ptr<rational_resampler_xxf> sptr;
    static sptr make(unsigned interp, unsigned decim, const std::vector<float> &taps, int max_items = 1 << 16, int device = 0)
    {
        return sptr(new rational_resampler_xxf(interp, decim, taps, max_items, device));
    }
    ~rational_resampler_xxf() { sfe_dsp_rs_destroy(d_h); }

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
        if (n_in <= 0) { consume_each(0); return 0; }
        int n_out = 0;
        check(sfe_dsp_rs_process(d_h, in, n_in, out, noutput_items, d_rate, &n_out), "rational_resampler::general_work");
        consume_each(n_in);
        return n_out;
    }

private:
    rational_resampler_xxf(unsigned interp, unsigned decim, const std::vector<float> &taps, int max_items, int device)
        : gr::block(CPLX ? "sfe_rational_resampler_ccf" : "sfe_rational_resampler_fff",
                    gr::io_signature::make(1, 1, CPLX ? sizeof(gr_complex) : sizeof(float)),
                    gr::io_signature::make(1, 1, CPLX ? sizeof(gr_complex) : sizeof(float))),
          d_h(0), d_interp(interp), d_decim(decim), d_blk(max_items), d_rate((float)decim / (float)interp)
    {
        set_relative_rate((double)interp / decim);
        check(sfe_dsp_rs_create(taps.data(), (int)taps.size(), (int)interp, max_items, CPLX ? 1 : 0, 1, device, SFE_RS_RESAMPLE, &d_h),
              "rational_resampler");
    }
    sfe_rs_t d_h;
    unsigned d_interp, d_decim;
    int d_blk;
    float d_rate;
};
typedef rational_resampler_xxf<true> rational_resampler_ccf;
typedef rational_resampler_xxf<false> rational_resampler_fff;

}  // namespace sfe
}  // namespace gr

#endif
