/* syntax-check stand-in for <gnuradio/io_signature.h> (see ../README.md) */
#ifndef MOCK_GNURADIO_IO_SIGNATURE_H
#define MOCK_GNURADIO_IO_SIGNATURE_H
#include <cstddef>
#include <memory>
#include <vector>
namespace gr {
class io_signature
{
public:
    typedef std::shared_ptr<io_signature> sptr;
    static sptr make(int /*min_streams*/, int /*max_streams*/, int /*sizeof_stream_item*/) { return sptr(); }
    static sptr makev(int /*min_streams*/, int /*max_streams*/, const std::vector<int>& /*sizeof_stream_items*/) { return sptr(); }
};
} // namespace gr
#endif
