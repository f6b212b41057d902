// gr_compat.h -- the slice of the GNU Radio runtime API that the adapters in this directory
// are written against.  With GNU Radio installed the real headers are used; without it (this
// image has none) a minimal stand-in with the same names, signatures and class shapes lets the
// adapters compile and be driven by a test harness exactly as the scheduler would drive them:
//   gr::sync_block::work(int noutput_items, gr_vector_const_void_star&, gr_vector_void_star&)
//   gr::block::general_work(int, gr_vector_int&, gr_vector_const_void_star&, gr_vector_void_star&)
// are the entry points gr-simplefe's own blocks implement
// (gr-simplefe/lib/source_c_impl.h:51-53, sink_c_impl.cc:157-174).
//
// Shared-pointer type: GNU Radio 3.7/3.8 blocks are held in boost::shared_ptr, 3.9+ in
// std::shared_ptr (gr-simplefe targets 3.7: include/simplefe/source_c.h:39).  tb->connect() only
// accepts the runtime's own kind, so the adapters' sptr is DERIVED from gr::basic_block_sptr
// (sptr_of<T> below) instead of being spelled out.
#ifndef GR_SFE_COMPAT_H_
#define GR_SFE_COMPAT_H_

#if defined(__has_include)
#if __has_include(<gnuradio/sync_block.h>)
#define GR_SFE_HAVE_GNURADIO 1
#endif
#endif

#ifdef GR_SFE_HAVE_GNURADIO
#include <gnuradio/io_signature.h>
#include <gnuradio/sync_block.h>
#include <gnuradio/sync_decimator.h>
#include <gnuradio/block.h>
#else
#include <complex>
#include <memory>
#include <string>
#include <vector>

// the stand-in's shared pointer template (tests/host/test_gr_sptr.cpp swaps in a boost-like one to
// show that the adapters follow whatever the runtime uses)
#ifndef GR_SFE_STANDIN_SPTR
#define GR_SFE_STANDIN_SPTR std::shared_ptr
#endif

typedef std::complex<float> gr_complex;
typedef std::vector<const void *> gr_vector_const_void_star;
typedef std::vector<void *> gr_vector_void_star;
typedef std::vector<int> gr_vector_int;

namespace gr {
class io_signature
{
public:
    typedef GR_SFE_STANDIN_SPTR<io_signature> sptr;
    static sptr make(int min_streams, int max_streams, int sizeof_stream_item)
    {
        return sptr(new io_signature(min_streams, max_streams, sizeof_stream_item));
    }
    int min_streams() const { return d_min; }
    int max_streams() const { return d_max; }
    int sizeof_stream_item(int) const { return d_size; }

private:
    io_signature(int a, int b, int c) : d_min(a), d_max(b), d_size(c) {}
    int d_min, d_max, d_size;
};

// As in the runtime, every level has a protected default constructor "to allow pure virtual
// interface sub-classes" (gnuradio-runtime/include/gnuradio/sync_block.h): the public block
// classes inherit VIRTUALLY from sync_block / block and only the private _impl class names the
// constructor with arguments -- the shape of gr-simplefe's blocks (source_c.h:36, source_c_impl.h).
class basic_block
{
public:
    virtual ~basic_block() {}
    const std::string &name() const { return d_name; }
    io_signature::sptr input_signature() const { return d_in; }
    io_signature::sptr output_signature() const { return d_out; }

protected:
    basic_block() {}
    basic_block(const std::string &n, io_signature::sptr i, io_signature::sptr o) : d_name(n), d_in(i), d_out(o) {}
    std::string d_name;
    io_signature::sptr d_in, d_out;
};
typedef GR_SFE_STANDIN_SPTR<basic_block> basic_block_sptr;

class block : public basic_block
{
public:
    virtual void forecast(int noutput_items, gr_vector_int &ninput_items_required)
    {
        for (size_t i = 0; i < ninput_items_required.size(); i++) ninput_items_required[i] = noutput_items;
    }
    virtual int general_work(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items,
                             gr_vector_void_star &output_items) = 0;
    virtual bool start() { return true; }
    virtual bool stop() { return true; }
    void consume_each(int n) { d_consumed = n; }
    int consumed() const { return d_consumed; }     // stand-in only: lets a harness see consume_each()
    void set_output_multiple(int m) { d_multiple = m; }
    int output_multiple() const { return d_multiple; }
    void set_relative_rate(double r) { d_rate = r; }
    double relative_rate() const { return d_rate; }

protected:
    block() : d_consumed(0), d_multiple(1), d_rate(1.0) {}
    block(const std::string &n, io_signature::sptr i, io_signature::sptr o)
        : basic_block(n, i, o), d_consumed(0), d_multiple(1), d_rate(1.0) {}
    int d_consumed, d_multiple;
    double d_rate;
};

class sync_block : public block
{
public:
    virtual int work(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &output_items) = 0;
    int general_work(int noutput_items, gr_vector_int &, gr_vector_const_void_star &in, gr_vector_void_star &out)
    {
        int r = work(noutput_items, in, out);
        if (r > 0) consume_each(r);
        return r;
    }

protected:
    sync_block() {}
    sync_block(const std::string &n, io_signature::sptr i, io_signature::sptr o) : block(n, i, o) {}
};

class sync_decimator : public sync_block
{
public:
    unsigned decimation() const { return d_decim; }

protected:
    sync_decimator() : d_decim(1) {}
    sync_decimator(const std::string &n, io_signature::sptr i, io_signature::sptr o, unsigned d) : sync_block(n, i, o), d_decim(d)
    {
        set_relative_rate(1.0 / d);
    }
    unsigned d_decim;
};
}  // namespace gr
#endif  // GR_SFE_HAVE_GNURADIO

namespace gr {
namespace sfe {
// the runtime's shared pointer template, rebound to T: boost::shared_ptr<T> under GNU Radio 3.7/3.8,
// std::shared_ptr<T> under 3.9+ and under the stand-in
template <class P, class T> struct rebind_sptr;
template <template <class> class SP, class U, class T> struct rebind_sptr<SP<U>, T> { typedef SP<T> type; };
template <class T> struct sptr_of { typedef typename rebind_sptr<gr::basic_block_sptr, T>::type type; };
}  // namespace sfe
}  // namespace gr
#endif
