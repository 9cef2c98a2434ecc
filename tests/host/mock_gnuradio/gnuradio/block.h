#ifndef MOCK_GR_BLOCK_H
#define MOCK_GR_BLOCK_H
// names and signatures of gr::basic_block / gr::block that gnss-sdr's acquisition blocks use; no scheduler behind them
#include "gnuradio/gr_complex.h"
#include "gnuradio/io_signature.h"
#include "gnuradio/thread/thread.h"
#include "gnuradio/types.h"
#include "pmt/pmt.h"
#include <memory>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

namespace gr
{
class basic_block : public std::enable_shared_from_this<basic_block>
{
public:
    virtual ~basic_block() = default;
    const std::string& name() const { return d_name; }
    long unique_id() const { return d_id; }
    io_signature::sptr input_signature() const { return d_in; }
    io_signature::sptr output_signature() const { return d_out; }
    void message_port_register_out(pmt::pmt_t port) { d_out_ports.push_back(pmt::symbol_to_string(port)); }
    void message_port_pub(pmt::pmt_t port, pmt::pmt_t msg) { published.emplace_back(pmt::symbol_to_string(port), std::move(msg)); }
    // test harness view of what the block sent
    std::vector<std::pair<std::string, pmt::pmt_t>> published;
    std::vector<std::string> d_out_ports;

protected:
    basic_block(const std::string& name, io_signature::sptr in, io_signature::sptr out) : d_name(name), d_in(std::move(in)), d_out(std::move(out))
    {
        static long next_id = 0;
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
    virtual int general_work(int noutput_items, gr_vector_int& ninput_items, gr_vector_const_void_star& input_items, gr_vector_void_star& output_items) = 0;
    virtual void forecast(int, gr_vector_int&) {}
    virtual bool start() { return true; }
    virtual bool stop() { return true; }
    void consume_each(int how_many_items) { consumed_last = how_many_items; consumed_total += how_many_items; }
    void set_relative_rate(double) {}
    void set_max_noutput_items(int) {}
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
