/* Minimal FUNCTIONAL stand-in for <gnuradio/block.h> + what it pulls in (basic_block.h, tags.h, types.h, sptr_magic.h):
 * see ../README.md.  Signatures as in GNU Radio 3.10; the bodies do the bookkeeping a scheduler thread does around one
 * general_work() call -- item counters, consume / produce, stored tags, published messages -- so that a block written
 * against the real headers can be EXECUTED by gr::mock::harness (below) and everything it hands to the runtime can be
 * read back.  There is no flowgraph, no buffer allocator, no thread-per-block scheduler here. */
#ifndef MOCK_GNURADIO_BLOCK_H
#define MOCK_GNURADIO_BLOCK_H
#include <gnuradio/io_signature.h>
#include <pmt/pmt.h>

#include <algorithm>
#include <complex>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
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

namespace mock {
class harness;
}

class basic_block
{
    friend class mock::harness;
    std::vector<pmt::pmt_t> d_out_ports, d_in_ports;
    std::vector<std::pair<pmt::pmt_t, pmt::pmt_t>> d_published;      /* (port, message) in publication order */

public:
    virtual ~basic_block() {}
    void message_port_register_out(pmt::pmt_t port_id) { d_out_ports.push_back(port_id); }
    void message_port_register_in(pmt::pmt_t port_id) { d_in_ports.push_back(port_id); }
    void message_port_pub(pmt::pmt_t port_id, pmt::pmt_t msg)
    {
        if (std::find(d_out_ports.begin(), d_out_ports.end(), port_id) == d_out_ports.end())
            throw std::invalid_argument("message_port_pub: port not registered: " + pmt::symbol_to_string(port_id)); /* as GNU Radio */
        d_published.emplace_back(port_id, msg);
    }
};

class block : public basic_block
{
    friend class mock::harness;
    std::string d_name;
    io_signature::sptr d_in_sig, d_out_sig;
    int d_output_multiple = 1, d_min_noutput_items = 0, d_tpp = 1;
    std::vector<uint64_t> d_nitems_read, d_nitems_written;            /* maintained by the harness */
    std::vector<int> d_consumed, d_produced;                          /* of the call in progress */
    std::vector<std::vector<tag_t>> d_out_tags;                       /* per output port, as added */

public:
    enum work_return_t { WORK_CALLED_PRODUCE = -2, WORK_DONE = -1 };
    enum tag_propagation_policy_t { TPP_DONT = 0, TPP_ALL_TO_ALL = 1, TPP_ONE_TO_ONE = 2, TPP_CUSTOM = 3 };

    virtual void forecast(int noutput_items, gr_vector_int& ninput_items_required)
    {
        for (auto& n : ninput_items_required) n = noutput_items;      /* GNU Radio's default: 1:1 */
    }
    virtual int general_work(int noutput_items, gr_vector_int& ninput_items, gr_vector_const_void_star& input_items,
                             gr_vector_void_star& output_items) = 0;
    virtual bool start() { return true; }
    virtual bool stop() { return true; }

    const std::string& name() const { return d_name; }
    io_signature::sptr input_signature() const { return d_in_sig; }
    io_signature::sptr output_signature() const { return d_out_sig; }
    void set_output_multiple(int m) { if (m < 1) throw std::invalid_argument("set_output_multiple"); d_output_multiple = m; }
    int output_multiple() const { return d_output_multiple; }
    void set_min_noutput_items(int m) { d_min_noutput_items = m; }
    int min_noutput_items() const { return d_min_noutput_items; }
    void set_tag_propagation_policy(tag_propagation_policy_t p) { d_tpp = int(p); }
    tag_propagation_policy_t tag_propagation_policy() const { return tag_propagation_policy_t(d_tpp); }
    void consume(int which_input, int how_many_items)
    {
        if (which_input < 0 || size_t(which_input) >= d_consumed.size() || how_many_items < 0) throw std::out_of_range("consume");
        d_consumed[size_t(which_input)] += how_many_items;
    }
    void consume_each(int how_many_items)
    {
        if (how_many_items < 0) throw std::out_of_range("consume_each");
        for (auto& c : d_consumed) c += how_many_items;
    }
    void produce(int which_output, int how_many_items)
    {
        if (which_output < 0 || size_t(which_output) >= d_produced.size() || how_many_items < 0) throw std::out_of_range("produce");
        d_produced[size_t(which_output)] += how_many_items;
    }
    uint64_t nitems_read(unsigned int which_input) { return d_nitems_read.at(which_input); }
    uint64_t nitems_written(unsigned int which_output) { return d_nitems_written.at(which_output); }
    void add_item_tag(unsigned int which_output, uint64_t abs_offset, const pmt::pmt_t& key, const pmt::pmt_t& value,
                      const pmt::pmt_t& srcid = pmt::get_PMT_F())
    {
        tag_t t;
        t.offset = abs_offset; t.key = key; t.value = value; t.srcid = srcid;
        add_item_tag(which_output, t);
    }
    void add_item_tag(unsigned int which_output, const tag_t& tag)
    {
        if (!pmt::is_symbol(tag.key)) throw pmt::wrong_type("add_item_tag: key");
        /* GNU Radio (block_detail::add_item_tag): "tag offset is before the items this call can still write" is an error */
        if (tag.offset < d_nitems_written.at(which_output)) throw std::out_of_range("add_item_tag: offset behind nitems_written");
        d_out_tags.at(which_output).push_back(tag);
    }
    void get_tags_in_range(std::vector<tag_t>& v, unsigned int /*which_input*/, uint64_t /*abs_start*/, uint64_t /*abs_end*/) { v.clear(); }

protected:
    block() {}
    block(const std::string& name, gr::io_signature::sptr input_signature, gr::io_signature::sptr output_signature)
        : d_name(name), d_in_sig(input_signature), d_out_sig(output_signature)
    {
        if (!d_in_sig || !d_out_sig) throw std::invalid_argument("gr::block: null io_signature");
        d_nitems_read.assign(size_t(std::max(d_in_sig->max_streams(), 0)), 0);
        d_nitems_written.assign(size_t(std::max(d_out_sig->max_streams(), 0)), 0);
        d_consumed.assign(d_nitems_read.size(), 0);
        d_produced.assign(d_nitems_written.size(), 0);
        d_out_tags.resize(d_nitems_written.size());
    }
};

namespace mock {

/* One block under a scheduler's hands.  call() is one general_work(): the caller says how many input items are on offer
 * (taken from the stream at the block's read position) and how much room every output buffer has; the harness owns the
 * buffers (with guard words behind each output), applies consume / produce or the returned count the way
 * block_executor does, appends what was produced to per-port captures and checks the rules a scheduler relies on --
 * never more consumed than offered, never more produced than there was room for, no write behind the room given. */
class harness
{
    std::shared_ptr<block> d_b;
    std::vector<std::vector<unsigned char>> d_capture;    /* per output port: every item produced so far */
    bool d_done = false;
    static constexpr uint32_t kGuard = 0xDEADC0DEu;

public:
    explicit harness(std::shared_ptr<block> b) : d_b(std::move(b))
    {
        if (!d_b) throw std::invalid_argument("harness: null block");
        d_capture.resize(d_b->d_nitems_written.size());
    }
    block& blk() { return *d_b; }
    size_t n_outputs() const { return d_capture.size(); }
    size_t out_itemsize(size_t port) const { return size_t(d_b->d_out_sig->sizeof_stream_item(int(port))); }
    size_t in_itemsize(size_t port) const { return size_t(d_b->d_in_sig->sizeof_stream_item(int(port))); }
    bool done() const { return d_done; }
    uint64_t nitems_read(size_t port) const { return d_b->d_nitems_read.at(port); }
    uint64_t nitems_written(size_t port) const { return d_b->d_nitems_written.at(port); }
    const std::vector<unsigned char>& captured(size_t port) const { return d_capture.at(port); }
    const std::vector<tag_t>& tags(size_t port) const { return d_b->d_out_tags.at(port); }
    const std::vector<std::pair<pmt::pmt_t, pmt::pmt_t>>& messages() const { return d_b->d_published; }
    int forecast(int noutput_items)
    {
        gr_vector_int req(d_b->d_nitems_read.size(), 0);
        d_b->forecast(noutput_items, req);
        return req.empty() ? 0 : req[0];
    }
    /* -> items consumed from input 0 (single-input blocks); `in` points at the item at the read position */
    int call(const void* in, int n_in_offered, int n_out_room)
    {
        if (d_done) return 0;
        const size_t n_out = d_capture.size();
        std::vector<std::vector<unsigned char>> bufs(n_out);
        gr_vector_void_star outs(n_out);
        for (size_t p = 0; p < n_out; p++) {
            bufs[p].assign(size_t(n_out_room) * out_itemsize(p) + 64, 0xA5);
            for (size_t g = 0; g < 16; g++) std::memcpy(bufs[p].data() + size_t(n_out_room) * out_itemsize(p) + 4 * g, &kGuard, 4);
            outs[p] = bufs[p].data();
        }
        gr_vector_int nin(d_b->d_nitems_read.size(), n_in_offered);
        gr_vector_const_void_star ins(d_b->d_nitems_read.size(), in);
        std::fill(d_b->d_consumed.begin(), d_b->d_consumed.end(), 0);
        std::fill(d_b->d_produced.begin(), d_b->d_produced.end(), 0);
        const int rc = d_b->general_work(n_out_room, nin, ins, outs);
        if (rc == block::WORK_DONE) { d_done = true; return 0; }
        std::vector<int> produced(n_out, 0);
        if (rc == block::WORK_CALLED_PRODUCE) produced = d_b->d_produced;
        else if (rc >= 0) std::fill(produced.begin(), produced.end(), rc);
        else throw std::logic_error("general_work returned " + std::to_string(rc));
        for (size_t p = 0; p < n_out; p++) {
            if (produced[p] < 0 || produced[p] > n_out_room)
                throw std::logic_error("port " + std::to_string(p) + ": produced " + std::to_string(produced[p]) + " items, room for " + std::to_string(n_out_room));
            for (size_t g = 0; g < 16; g++) {
                uint32_t w;
                std::memcpy(&w, bufs[p].data() + size_t(n_out_room) * out_itemsize(p) + 4 * g, 4);
                if (w != kGuard) throw std::logic_error("port " + std::to_string(p) + ": output buffer overrun");
            }
            d_capture[p].insert(d_capture[p].end(), bufs[p].begin(), bufs[p].begin() + long(size_t(produced[p]) * out_itemsize(p)));
            d_b->d_nitems_written[p] += uint64_t(produced[p]);
        }
        int consumed0 = 0;
        for (size_t i = 0; i < d_b->d_consumed.size(); i++) {
            if (d_b->d_consumed[i] > n_in_offered)
                throw std::logic_error("input " + std::to_string(i) + ": consumed " + std::to_string(d_b->d_consumed[i]) + " of " + std::to_string(n_in_offered));
            d_b->d_nitems_read[i] += uint64_t(d_b->d_consumed[i]);
            if (i == 0) consumed0 = d_b->d_consumed[i];
        }
        /* every tag must sit on an item that has been, or can still be, written (a scheduler prunes nothing here: the test
         * reads all of them back) */
        return consumed0;
    }
};

} // namespace mock
} // namespace gr

namespace gnuradio {
template <typename T, typename... Args>
std::shared_ptr<T> make_block_sptr(Args&&... args)
{
    return std::shared_ptr<T>(new T(std::forward<Args>(args)...));
}
} // namespace gnuradio
#endif
