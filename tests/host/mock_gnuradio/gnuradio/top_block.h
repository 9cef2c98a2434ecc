#ifndef MOCK_GR_TOP_BLOCK_H
#define MOCK_GR_TOP_BLOCK_H
#include "gnuradio/block.h"
#include <memory>
#include <mutex>
#include <vector>
namespace gr
{
// connect / msg_connect record what was wired (tests/host/mini_flowgraph.h builds its threads and buffers from it); msg_connect also subscribes the
// destination to the source's port, so that message_port_pub reaches it
class top_block
{
public:
    struct edge
    {
        basic_block_sptr src;
        int src_port;
        basic_block_sptr dst;
        int dst_port;
    };
    struct msg_edge
    {
        basic_block_sptr src;
        std::string src_port;
        basic_block_sptr dst;
        std::string dst_port;
    };
    void connect(basic_block_sptr src, int src_port, basic_block_sptr dst, int dst_port)
    {
        std::lock_guard<std::mutex> lk(d_mu);
        d_edges.push_back({std::move(src), src_port, std::move(dst), dst_port});
    }
    void disconnect(basic_block_sptr src, int src_port, basic_block_sptr dst, int dst_port)
    {
        std::lock_guard<std::mutex> lk(d_mu);
        for (auto it = d_edges.begin(); it != d_edges.end();)
            it = (it->src == src && it->src_port == src_port && it->dst == dst && it->dst_port == dst_port) ? d_edges.erase(it) : std::next(it);
    }
    void msg_connect(basic_block_sptr src, pmt::pmt_t src_port, basic_block_sptr dst, pmt::pmt_t dst_port)
    {
        if (!src || !dst) return;
        src->subscribe(pmt::symbol_to_string(src_port), dst, pmt::symbol_to_string(dst_port));
        std::lock_guard<std::mutex> lk(d_mu);
        d_msg_edges.push_back({src, pmt::symbol_to_string(src_port), dst, pmt::symbol_to_string(dst_port)});
    }
    void msg_disconnect(basic_block_sptr src, pmt::pmt_t src_port, basic_block_sptr dst, pmt::pmt_t dst_port)
    {
        if (!src || !dst) return;
        src->unsubscribe(pmt::symbol_to_string(src_port), dst, pmt::symbol_to_string(dst_port));
        std::lock_guard<std::mutex> lk(d_mu);
        for (auto it = d_msg_edges.begin(); it != d_msg_edges.end();)
            it = (it->src == src && it->dst == dst && it->src_port == pmt::symbol_to_string(src_port) && it->dst_port == pmt::symbol_to_string(dst_port)) ? d_msg_edges.erase(it) : std::next(it);
    }
    std::vector<edge> edges()
    {
        std::lock_guard<std::mutex> lk(d_mu);
        return d_edges;
    }
    std::vector<msg_edge> msg_edges()
    {
        std::lock_guard<std::mutex> lk(d_mu);
        return d_msg_edges;
    }

private:
    std::mutex d_mu;
    std::vector<edge> d_edges;
    std::vector<msg_edge> d_msg_edges;
};
typedef std::shared_ptr<top_block> top_block_sptr;
inline top_block_sptr make_top_block(const std::string&) { return std::make_shared<top_block>(); }
}  // namespace gr
#endif
