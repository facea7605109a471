/* stand-in for <gnuradio/io_signature.h> (see ../README.md): keeps what it is given */
#ifndef MOCK_GNURADIO_IO_SIGNATURE_H
#define MOCK_GNURADIO_IO_SIGNATURE_H
#include <cstddef>
#include <memory>
#include <stdexcept>
#include <vector>
namespace gr {
class io_signature
{
    int d_min, d_max;
    std::vector<int> d_sizes;
    io_signature(int mn, int mx, std::vector<int> sizes) : d_min(mn), d_max(mx), d_sizes(std::move(sizes))
    {
        if (mn < 0 || (mx >= 0 && mx < mn) || (mx > 0 && d_sizes.empty())) throw std::invalid_argument("gr::io_signature");
    }

public:
    typedef std::shared_ptr<io_signature> sptr;
    static sptr make(int min_streams, int max_streams, int sizeof_stream_item)
    {
        return sptr(new io_signature(min_streams, max_streams, std::vector<int>{ sizeof_stream_item }));
    }
    static sptr makev(int min_streams, int max_streams, const std::vector<int>& sizeof_stream_items)
    {
        return sptr(new io_signature(min_streams, max_streams, sizeof_stream_items));
    }
    int min_streams() const { return d_min; }
    int max_streams() const { return d_max; }
    /* the last size repeats for further streams, as in GNU Radio */
    int sizeof_stream_item(int index) const { return d_sizes[size_t(index) < d_sizes.size() ? size_t(index) : d_sizes.size() - 1]; }
    std::vector<int> sizeof_stream_items() const { return d_sizes; }
};
} // namespace gr
#endif
