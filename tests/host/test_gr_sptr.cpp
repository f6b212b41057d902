// Compile-time check of the adapters' class shape against a runtime whose shared pointer is NOT
// std::shared_ptr -- GNU Radio 3.7/3.8 hold blocks in boost::shared_ptr
// (gr-simplefe/include/simplefe/source_c.h:39) and top_block::connect() takes
// gr::basic_block_sptr, so a block whose sptr is spelled std::shared_ptr would not connect.
// A boost-like template stands in for boost (not in this image).  No GPU, nothing is constructed.
#include <memory>
#include <type_traits>

namespace fakeboost {
template <class T>
class shared_ptr
{
public:
    shared_ptr() {}
    template <class U> explicit shared_ptr(U *p) : d(p) {}
    template <class U, class = typename std::enable_if<std::is_convertible<U *, T *>::value>::type>
    shared_ptr(const shared_ptr<U> &o) : d(o.d) {}
    T *operator->() const { return d.get(); }
    T *get() const { return d.get(); }
    std::shared_ptr<T> d;
};
}  // namespace fakeboost
#define GR_SFE_STANDIN_SPTR fakeboost::shared_ptr
#include "../../include/gr_sfe/blocks.h"

using namespace gr::sfe;
static_assert(std::is_same<fir_ccf::sptr, fakeboost::shared_ptr<fir_ccf>>::value, "sptr follows the runtime's template");
static_assert(std::is_same<decimate_ccf::sptr, fakeboost::shared_ptr<decimate_ccf>>::value, "");
static_assert(std::is_same<rational_resampler_fff::sptr, fakeboost::shared_ptr<rational_resampler_fff>>::value, "");
// what tb->connect(src, 0, blk, 0) needs: the block's sptr converts to basic_block_sptr
static_assert(std::is_convertible<fir_ccf::sptr, gr::basic_block_sptr>::value, "");
static_assert(std::is_convertible<fir_fff_sync::sptr, gr::basic_block_sptr>::value, "");
static_assert(std::is_convertible<decimate_fff::sptr, gr::basic_block_sptr>::value, "");
static_assert(std::is_convertible<rational_resampler_ccf::sptr, gr::basic_block_sptr>::value, "");
// virtual inheritance from the runtime's block types, abstract public classes, private impl
static_assert(std::is_base_of<gr::block, fir_ccf>::value && std::is_abstract<fir_ccf>::value, "");
static_assert(std::is_base_of<gr::sync_block, fir_ccf_sync>::value && std::is_abstract<fir_ccf_sync>::value, "");
static_assert(std::is_base_of<gr::block, decimate_ccf>::value && std::is_abstract<decimate_ccf>::value, "");
static_assert(std::is_base_of<fir_ccf, fir_xxf_impl<true>>::value && !std::is_abstract<fir_xxf_impl<true>>::value, "");

static void connect(gr::basic_block_sptr) {}
int main(int argc, char **)
{
    if (argc > 100) {                       // never runs: instantiates make() and the conversion
        std::vector<float> t(3, 1.0f);
        connect(fir_ccf::make(t));
        connect(decimate_ccf::make(t, 8));
        connect(rational_resampler_ccf::make(3, 5, t));
    }
    return 0;
}
