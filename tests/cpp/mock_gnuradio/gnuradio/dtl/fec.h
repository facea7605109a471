/* syntax-check stand-in for gr-dtl's <gnuradio/dtl/fec.h> (the reference documents the interface at
 * include/gnuradio/dtl/fec.h:25-31: decode(const float*, int* nit, unsigned char*), get_k, get_n; see ../../README.md) */
#ifndef MOCK_GNURADIO_DTL_FEC_H
#define MOCK_GNURADIO_DTL_FEC_H
#include <memory>
namespace gr {
namespace dtl {
class fec_dec
{
public:
    typedef std::shared_ptr<fec_dec> sptr;
    virtual ~fec_dec() {}
    virtual int decode(const float* in_data, int* nit, unsigned char* out_data) = 0;
    virtual int get_k() = 0;
    virtual int get_n() = 0;
};
} // namespace dtl
} // namespace gr
#endif
