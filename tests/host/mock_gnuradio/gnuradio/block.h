#ifndef MOCK_GR_BLOCK_H
#define MOCK_GR_BLOCK_H
// names and signatures of gr::basic_block / gr::block that gnss-sdr's blocks use.  No scheduler lives here: a test either drives general_work by hand
// (published / deliver() below) or puts the blocks into tests/host/mini_flowgraph.h, a thread-per-block scheduler over these classes -- messages published on a
// port that top_block::msg_connect has wired are then queued at the subscriber and handled on the SUBSCRIBER's thread, as GNU Radio's scheduler does
#include "gnuradio/gr_complex.h"
#include "gnuradio/io_signature.h"
#include "gnuradio/thread/thread.h"
#include "gnuradio/types.h"
#include "pmt/pmt.h"
#include <atomic>
#include <cstdint>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

namespace gr
{
struct tag_t
{
    uint64_t offset{0};
    pmt::pmt_t key;
    pmt::pmt_t value;
    pmt::pmt_t srcid;
};

class basic_block : public std::enable_shared_from_this<basic_block>
{
public:
    virtual ~basic_block() = default;
    const std::string& name() const { return d_name; }
    long unique_id() const { return d_id; }
    io_signature::sptr input_signature() const { return d_in; }
    io_signature::sptr output_signature() const { return d_out; }
    void message_port_register_out(pmt::pmt_t port) { d_out_ports.push_back(pmt::symbol_to_string(port)); }
    void message_port_pub(pmt::pmt_t port, pmt::pmt_t msg)
    {
        const std::string name = pmt::symbol_to_string(port);
        std::vector<std::pair<std::weak_ptr<basic_block>, std::string>> to;
        {
            std::lock_guard<std::mutex> lk(d_msg_mu);
            published.emplace_back(name, msg);
            auto it = d_subscribers.find(name);
            if (it != d_subscribers.end()) to = it->second;
        }
        for (auto& sub : to)
            if (auto dst = sub.first.lock()) dst->post(sub.second, msg);
    }
    void message_port_register_in(pmt::pmt_t port) { d_in_ports.push_back(pmt::symbol_to_string(port)); }
    template <typename F>
    void set_msg_handler(pmt::pmt_t port, F handler) { d_handlers[pmt::symbol_to_string(port)] = handler; }
    // test harness: deliver a message to an input port as the scheduler would -- at once, on the caller's thread
    void deliver(const std::string& port, pmt::pmt_t msg)
    {
        auto it = d_handlers.find(port);
        if (it != d_handlers.end()) it->second(msg);
    }
    // ---- the scheduler's side of message passing (mini_flowgraph.h): a message for this block is queued, the block's own thread handles it
    void post(const std::string& port, pmt::pmt_t msg)
    {
        std::function<void()> wake;
        {
            std::lock_guard<std::mutex> lk(d_msg_mu);
            d_inbox.emplace_back(port, std::move(msg));
            wake = d_wake;
        }
        if (wake) wake();
    }
    bool has_pending_messages()
    {
        std::lock_guard<std::mutex> lk(d_msg_mu);
        return !d_inbox.empty();
    }
    int handle_pending_messages()  // on the block's thread, between two general_work calls
    {
        int n = 0;
        for (;;)
            {
                std::pair<std::string, pmt::pmt_t> m;
                {
                    std::lock_guard<std::mutex> lk(d_msg_mu);
                    if (d_inbox.empty()) return n;
                    m = std::move(d_inbox.front());
                    d_inbox.pop_front();
                }
                deliver(m.first, m.second);
                n++;
            }
    }
    void set_wake(std::function<void()> f)
    {
        std::lock_guard<std::mutex> lk(d_msg_mu);
        d_wake = std::move(f);
    }
    void subscribe(const std::string& port, const std::shared_ptr<basic_block>& dst, const std::string& dst_port)
    {
        std::lock_guard<std::mutex> lk(d_msg_mu);
        d_subscribers[port].emplace_back(dst, dst_port);
    }
    void unsubscribe(const std::string& port, const std::shared_ptr<basic_block>& dst, const std::string& dst_port)
    {
        std::lock_guard<std::mutex> lk(d_msg_mu);
        auto& v = d_subscribers[port];
        for (auto it = v.begin(); it != v.end();)
            it = (it->first.lock() == dst && it->second == dst_port) ? v.erase(it) : std::next(it);
    }
    std::vector<std::string> d_in_ports;
    std::map<std::string, std::function<void(pmt::pmt_t)>> d_handlers;
    // test harness view of what the block sent
    std::vector<std::pair<std::string, pmt::pmt_t>> published;
    std::vector<std::string> d_out_ports;
    std::mutex d_msg_mu;
    std::deque<std::pair<std::string, pmt::pmt_t>> d_inbox;
    std::map<std::string, std::vector<std::pair<std::weak_ptr<basic_block>, std::string>>> d_subscribers;
    std::function<void()> d_wake;

protected:
    basic_block(const std::string& name, io_signature::sptr in, io_signature::sptr out) : d_name(name), d_in(std::move(in)), d_out(std::move(out))
    {
        static std::atomic<long> next_id{0};  // blocks are built from several test threads
        d_id = next_id++;
    }
    std::string d_name;
    io_signature::sptr d_in, d_out;
    long d_id;
};
typedef std::shared_ptr<basic_block> basic_block_sptr;

class block : public basic_block
{
public:
    // (not pure in GNU Radio either: a message-only block such as channel_msg_receiver_cc never overrides it)
    virtual int general_work(int /*noutput_items*/, gr_vector_int& /*ninput_items*/, gr_vector_const_void_star& /*input_items*/, gr_vector_void_star& /*output_items*/) { return -1; }
    virtual void forecast(int, gr_vector_int&) {}
    virtual bool start() { return true; }
    virtual bool stop() { return true; }
    enum tag_propagation_policy_t { TPP_DONT = 0, TPP_ALL_TO_ALL = 1, TPP_ONE_TO_ONE = 2, TPP_CUSTOM = 3 };
    void consume_each(int how_many_items) { consumed_last = how_many_items; consumed_total += how_many_items; }
    void set_relative_rate(double) {}
    void set_alignment(int) {}
    void set_history(unsigned h) { d_history = h; }
    unsigned history() const { return d_history; }
    unsigned d_history{1};
    void set_relative_rate(uint64_t, uint64_t) {}
    void set_tag_propagation_policy(tag_propagation_policy_t) {}
    // stream position as the scheduler keeps it: the harness advances nitems_read by what the block consumed and nitems_written by what
    // general_work returned (mock_advance(), called by the harness after every general_work)
    uint64_t nitems_read(unsigned) const { return d_nitems_read; }
    uint64_t nitems_written(unsigned) const { return d_nitems_written; }
    void mock_advance(int produced)
    {
        d_nitems_read += static_cast<uint64_t>(consumed_last);
        if (produced > 0) d_nitems_written += static_cast<uint64_t>(produced);
    }
    void get_tags_in_range(std::vector<tag_t>& v, unsigned, uint64_t start, uint64_t end)
    {
        v.clear();
        for (const auto& t : input_tags)
            if (t.offset >= start && t.offset < end) v.push_back(t);
    }
    void get_tags_in_range(std::vector<tag_t>& v, unsigned which, uint64_t start, uint64_t end, const pmt::pmt_t& key)
    {
        get_tags_in_range(v, which, start, end);
        std::vector<tag_t> k;
        for (const auto& t : v)
            if (pmt::eqv(t.key, key)) k.push_back(t);
        v.swap(k);
    }
    // relative to the read pointer of the call (GNU Radio: get_tags_in_window)
    void get_tags_in_window(std::vector<tag_t>& v, unsigned which, uint64_t rel_start, uint64_t rel_end, const pmt::pmt_t& key)
    {
        get_tags_in_range(v, which, d_nitems_read + rel_start, d_nitems_read + rel_end, key);
    }
    void add_item_tag(unsigned, const tag_t& t) { output_tags.push_back(t); }
    void add_item_tag(unsigned, uint64_t offset, const pmt::pmt_t& key, const pmt::pmt_t& value, const pmt::pmt_t& srcid = pmt::pmt_t())
    {
        tag_t t;
        t.offset = offset;
        t.key = key;
        t.value = value;
        t.srcid = srcid;
        output_tags.push_back(t);
    }
    std::vector<tag_t> input_tags, output_tags;  // test harness view
    uint64_t d_nitems_read{0}, d_nitems_written{0};
    void set_max_noutput_items(int m) { d_max_noutput = m; }
    int max_noutput_items() const { return d_max_noutput; }
    int d_max_noutput{0};  // 0: not set
    // test harness view
    int consumed_last{0};
    long long consumed_total{0};

protected:
    block(const std::string& name, io_signature::sptr in, io_signature::sptr out) : basic_block(name, std::move(in), std::move(out)) {}
    thread::mutex d_setlock;
};
typedef std::shared_ptr<block> block_sptr;
}  // namespace gr
#endif
