/* syntax-check stand-in for <gnuradio/block.h> + what it pulls in (basic_block.h, tags.h, types.h, sptr_magic.h):
 * see ../README.md.  Signatures as in GNU Radio 3.10. */
#ifndef MOCK_GNURADIO_BLOCK_H
#define MOCK_GNURADIO_BLOCK_H
#include <gnuradio/io_signature.h>
#include <pmt/pmt.h>

#include <complex>
#include <cstdint>
#include <memory>
#include <string>
#include <utility>
#include <vector>

typedef std::complex<float> gr_complex;
typedef std::vector<int> gr_vector_int;
typedef std::vector<const void*> gr_vector_const_void_star;
typedef std::vector<void*> gr_vector_void_star;

namespace gr {

struct tag_t {
    uint64_t offset = 0;
    pmt::pmt_t key, value, srcid;
};

class basic_block
{
public:
    virtual ~basic_block() {}
    void message_port_register_out(pmt::pmt_t /*port_id*/) {}
    void message_port_register_in(pmt::pmt_t /*port_id*/) {}
    void message_port_pub(pmt::pmt_t /*port_id*/, pmt::pmt_t /*msg*/) {}
};

class block : public basic_block
{
public:
    enum work_return_t { WORK_CALLED_PRODUCE = -2, WORK_DONE = -1 };
    enum tag_propagation_policy_t { TPP_DONT = 0, TPP_ALL_TO_ALL = 1, TPP_ONE_TO_ONE = 2, TPP_CUSTOM = 3 };

    virtual void forecast(int /*noutput_items*/, gr_vector_int& /*ninput_items_required*/) {}
    virtual int general_work(int noutput_items, gr_vector_int& ninput_items, gr_vector_const_void_star& input_items,
                             gr_vector_void_star& output_items) = 0;
    virtual bool start() { return true; }
    virtual bool stop() { return true; }

    void set_output_multiple(int) {}
    void set_min_noutput_items(int) {}
    void set_tag_propagation_policy(tag_propagation_policy_t) {}
    void consume(int /*which_input*/, int /*how_many_items*/) {}
    void consume_each(int /*how_many_items*/) {}
    void produce(int /*which_output*/, int /*how_many_items*/) {}
    uint64_t nitems_read(unsigned int /*which_input*/) { return 0; }
    uint64_t nitems_written(unsigned int /*which_output*/) { return 0; }
    void add_item_tag(unsigned int /*which_output*/, uint64_t /*abs_offset*/, const pmt::pmt_t& /*key*/,
                      const pmt::pmt_t& /*value*/, const pmt::pmt_t& /*srcid*/ = pmt::pmt_t()) {}
    void add_item_tag(unsigned int /*which_output*/, const tag_t& /*tag*/) {}
    void get_tags_in_range(std::vector<tag_t>& /*v*/, unsigned int /*which_input*/, uint64_t /*abs_start*/, uint64_t /*abs_end*/) {}

protected:
    block() {}
    block(const std::string& /*name*/, gr::io_signature::sptr /*input_signature*/, gr::io_signature::sptr /*output_signature*/) {}
};

} // namespace gr

namespace gnuradio {
template <typename T, typename... Args>
std::shared_ptr<T> make_block_sptr(Args&&... args)
{
    return std::shared_ptr<T>(new T(std::forward<Args>(args)...));
}
} // namespace gnuradio
#endif
