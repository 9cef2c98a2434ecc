// TEST INFRASTRUCTURE -- a thread-per-block scheduler over tests/host/mock_gnuradio, just enough of GNU Radio's runtime to run a Channel
// (src/algorithms/channel/adapters/channel.cc) the way a receiver's flowgraph does: every block on a thread of its own, messages handled on the RECEIVING
// block's thread between two general_work calls, one source buffer that all channel blocks read (gnss_flowgraph.cc:1227-1231) with back-pressure from the
// slowest reader, forecast() deciding when a block can run.  What is wired comes from gr::top_block (connect / msg_connect as recorded by the mock).
//
// Two ways of running the threads:
//   free-running   every thread runs whenever its block can (the real thing: cross-thread calls land while other blocks are inside general_work)
//   token          one thread at a time, in a fixed round -- source, blocks in the order they were wired, the control thread's turn --: the same threads, the same
//                  cross-thread calls (an acquisition block's thread calls ChannelFsm -> trk->start_tracking()), but a reproducible interleaving, so that two
//                  receivers (HIP blocks / the reference's blocks) can be compared event for event and sample for sample
#ifndef GSH_TEST_MINI_FLOWGRAPH_H
#define GSH_TEST_MINI_FLOWGRAPH_H
#include <gnuradio/block.h>
#include <gnuradio/top_block.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

class Mini_Flowgraph
{
public:
    // the source: `n` items of `item_size` bytes at `stream`, offered to its readers as a scheduler offers a buffer: at most buffer_items ahead of the slowest reader
    class Source_Block : public gr::block
    {
    public:
        Source_Block(size_t item_size) : gr::block("mini_source", gr::io_signature::make(0, 0, 0), gr::io_signature::make(1, 1, static_cast<int>(item_size))) {}
        int general_work(int, gr_vector_int&, gr_vector_const_void_star&, gr_vector_void_star&) override { return 0; }
    };

    struct Node_Stats
    {
        std::string name;
        uint64_t calls{0}, empty_calls{0}, consumed{0}, produced{0}, messages{0};
        double longest_gap_s{0.0};  // longest wall-clock time between two calls that made progress (free-running mode)
    };

    Mini_Flowgraph(const void* stream, size_t n_items, size_t item_size, size_t buffer_items, size_t source_chunk, bool token_mode)
        : d_stream(static_cast<const char*>(stream)), d_n(n_items), d_item(item_size), d_buffer(buffer_items), d_chunk(source_chunk), d_token(token_mode)
    {
        d_top = std::make_shared<gr::top_block>();
        d_source = std::make_shared<Source_Block>(item_size);
    }
    ~Mini_Flowgraph() { stop(); }

    gr::top_block_sptr top() const { return d_top; }
    gr::basic_block_sptr source() const { return d_source; }
    // token mode: called on the control turn of every round (the receiver's control thread work); free-running mode: the caller runs its own control thread
    void set_control_turn(std::function<bool()> f) { d_control = std::move(f); }
    // called (under no lock, on the source thread / turn) whenever the source has released items up to `head`
    void set_on_release(std::function<void(uint64_t)> f) { d_on_release = std::move(f); }
    uint64_t head() const { return d_head.load(); }
    bool failed() const { return d_failed.load(); }
    const std::string& failure() const { return d_failure; }

    void start()
    {
        build();
        d_stop.store(false);
        for (size_t i = 0; i < d_nodes.size(); i++) d_threads.emplace_back([this, i] { node_thread(i); });
        d_threads.emplace_back([this] { source_thread(); });
        if (d_token && d_control) d_threads.emplace_back([this] { control_thread(); });
    }

    // blocks until the source is exhausted and nothing can run any more (every block short of input, no message queued), or `timeout_s` has passed
    bool wait_until_drained(double timeout_s)
    {
        const auto t_end = std::chrono::steady_clock::now() + std::chrono::duration<double>(timeout_s);
        // (polled with plain sleeps: a timed wait on the condition variable is pthread_cond_clockwait, which gcc 11's ThreadSanitizer does not know releases the mutex)
        while (std::chrono::steady_clock::now() < t_end)
            {
                {
                    std::lock_guard<std::mutex> lk(d_mu);
                    if (d_failed.load()) return false;
                    if (d_head.load() >= d_n && all_idle_locked() && (!d_control_pending || !d_control_pending())) return true;
                }
                std::this_thread::sleep_for(std::chrono::milliseconds(2));
            }
        return false;
    }

    void stop()
    {
        {
            std::lock_guard<std::mutex> lk(d_mu);
            d_stop.store(true);
        }
        d_cv.notify_all();
        for (auto& t : d_threads)
            if (t.joinable()) t.join();
        d_threads.clear();
        for (auto& nd : d_nodes) nd->blk->set_wake(nullptr);
    }

    // what every block thread is doing right now (a run that does not drain)
    void dump_state()
    {
        std::lock_guard<std::mutex> lk(d_mu);
        std::printf("  flowgraph: source head %llu of %zu, slowest reader %llu, buffer %zu\n", static_cast<unsigned long long>(d_head.load()), d_n,
            static_cast<unsigned long long>(slowest_reader_locked()), d_buffer);
        for (auto& nd : d_nodes)
            std::printf("    %-34s read %-10llu %s%s calls %llu (empty %llu) consumed %llu produced %llu messages %llu\n", nd->st.name.c_str(), static_cast<unsigned long long>(nd->read),
                nd->idle ? "idle " : "BUSY ", nd->blk->has_pending_messages() ? "(messages pending)" : "", static_cast<unsigned long long>(nd->st.calls),
                static_cast<unsigned long long>(nd->st.empty_calls), static_cast<unsigned long long>(nd->st.consumed), static_cast<unsigned long long>(nd->st.produced),
                static_cast<unsigned long long>(nd->st.messages));
    }

    // free-running mode: tells wait_until_drained whether the caller's control thread still has work (queue not empty / handler running)
    void set_control_pending(std::function<bool()> f) { d_control_pending = std::move(f); }

    std::vector<Node_Stats> stats()
    {
        std::lock_guard<std::mutex> lk(d_mu);
        std::vector<Node_Stats> v;
        for (auto& nd : d_nodes) v.push_back(nd->st);
        return v;
    }
    // record the read pointer of a block after every call in which it consumed something (before start())
    void enable_trace(const gr::basic_block_sptr& b)
    {
        std::lock_guard<std::mutex> lk(d_mu);
        d_traced.push_back(b);
    }
    std::vector<uint64_t> trace_of(const gr::basic_block_sptr& b)
    {
        std::lock_guard<std::mutex> lk(d_mu);
        for (auto& nd : d_nodes)
            if (nd->blk == b) return nd->trace;
        return {};
    }
    uint64_t items_read(const gr::basic_block_sptr& b)
    {
        std::lock_guard<std::mutex> lk(d_mu);
        for (auto& nd : d_nodes)
            if (nd->blk == b) return nd->read;
        return 0;
    }

private:
    struct Edge_Buffer  // what a block with an output writes and ONE downstream block reads (tracking -> telemetry decoder): everything ever written is kept
    {
        size_t item{0};
        std::vector<char> data;
        uint64_t written{0};
    };
    struct Node
    {
        std::shared_ptr<gr::block> blk;
        bool from_source{false};
        std::shared_ptr<Edge_Buffer> in, out;
        uint64_t read{0};
        bool idle{false};
        uint64_t idle_seen{0};  // free-running mode: the input position the block was short of input at (all_idle_locked)
        bool traced{false};
        std::vector<uint64_t> trace;
        Node_Stats st;
        std::chrono::steady_clock::time_point last_progress;
    };

    void fail(const std::string& why)
    {
        std::lock_guard<std::mutex> lk(d_mu);
        if (!d_failed.exchange(true)) d_failure = why;
        d_cv.notify_all();
    }

    void build()
    {
        if (!d_nodes.empty()) return;
        auto node_of = [&](const gr::basic_block_sptr& b) -> Node* {
            for (auto& nd : d_nodes)
                if (nd->blk == b) return nd.get();
            auto blk = std::dynamic_pointer_cast<gr::block>(b);
            if (!blk || b == d_source) return nullptr;
            d_nodes.push_back(std::make_unique<Node>());
            d_nodes.back()->blk = blk;
            d_nodes.back()->st.name = blk->name() + "#" + std::to_string(blk->unique_id());
            return d_nodes.back().get();
        };
        for (const auto& e : d_top->edges())
            {
                Node* dst = node_of(e.dst);
                if (dst == nullptr) continue;
                if (e.src == d_source)
                    dst->from_source = true;
                else if (Node* src = node_of(e.src))
                    {
                        if (!src->out)
                            {
                                src->out = std::make_shared<Edge_Buffer>();
                                src->out->item = static_cast<size_t>(src->blk->output_signature()->sizeof_stream_item(0));
                            }
                        dst->in = src->out;
                    }
            }
        for (const auto& m : d_top->msg_edges())
            {
                node_of(m.src);
                node_of(m.dst);
            }
        for (auto& nd : d_nodes)
            for (const auto& t : d_traced)
                if (t == nd->blk) nd->traced = true;
        for (auto& nd : d_nodes) nd->blk->set_wake([this] {
            std::lock_guard<std::mutex> lk(d_mu);
            d_cv.notify_all();
        });
    }

    bool all_idle_locked()
    {
        for (auto& nd : d_nodes)
            {
                if (!nd->idle || nd->blk->has_pending_messages()) return false;
                // (a block thread that went to sleep short of input and has not yet seen the source's latest release is NOT idle: it will run as soon as it gets the
                // mutex -- on a loaded host that can take longer than wait_until_drained's next look, and the graph would be declared drained with periods to go)
                if (!d_token && (nd->from_source ? d_head.load() : (nd->in ? nd->in->written : 0)) != nd->idle_seen) return false;
            }
        return true;
    }

    uint64_t slowest_reader_locked()
    {
        uint64_t lo = UINT64_MAX;
        for (auto& nd : d_nodes)
            if (nd->from_source) lo = std::min(lo, nd->read);
        return lo == UINT64_MAX ? d_head.load() : lo;
    }

    // ---- token mode: turn 0 = source, 1..N = nodes, N+1 = control
    size_t n_turns() const { return d_nodes.size() + 2; }
    void wait_turn(size_t me, std::unique_lock<std::mutex>& lk)
    {
        d_cv.wait(lk, [&] { return d_stop.load() || d_turn == me; });
    }
    void pass_turn_locked()
    {
        d_turn = (d_turn + 1) % n_turns();
        if (d_turn == n_turns() - 1 && !d_control) d_turn = 0;
        d_cv.notify_all();
    }

    void source_thread()
    {
        for (;;)
            {
                uint64_t released = 0;
                {
                    std::unique_lock<std::mutex> lk(d_mu);
                    if (d_token)
                        {
                            wait_turn(0, lk);
                            if (d_stop.load()) return;
                            // a round in which nothing at all could run, with the buffer full and samples left: the graph is stuck (what GNU Radio would do too)
                            const uint64_t head = d_head.load();
                            if (head < d_n && head - slowest_reader_locked() + d_chunk <= d_buffer)
                                {
                                    d_head.store(std::min<uint64_t>(d_n, head + d_chunk));
                                    released = d_head.load();
                                    d_round_progress = true;
                                }
                            if (!d_round_progress && head < d_n && ++d_stuck_rounds > 200000)
                                {
                                    if (!d_failed.exchange(true)) d_failure = "token scheduler: no block can run and the source buffer is full";
                                    d_stop.store(true);
                                    d_cv.notify_all();
                                    return;
                                }
                            if (d_round_progress) d_stuck_rounds = 0;
                            d_round_progress = false;
                            pass_turn_locked();
                        }
                    else
                        {
                            d_cv.wait(lk, [&] { return d_stop.load() || (d_head.load() < d_n && d_head.load() - slowest_reader_locked() + d_chunk <= d_buffer); });
                            if (d_stop.load()) return;
                            d_head.store(std::min<uint64_t>(d_n, d_head.load() + d_chunk));
                            released = d_head.load();
                            d_cv.notify_all();
                        }
                }
                if (released != 0 && d_on_release) d_on_release(released);
                if (!d_token && d_head.load() >= d_n) return;
            }
    }

    void control_thread()
    {
        for (;;)
            {
                {
                    std::unique_lock<std::mutex> lk(d_mu);
                    wait_turn(n_turns() - 1, lk);
                    if (d_stop.load()) return;
                }
                const bool did = d_control();
                std::lock_guard<std::mutex> lk(d_mu);
                if (did) d_round_progress = true;
                pass_turn_locked();
            }
    }

    // one scheduler iteration of a block; returns true when it made progress (handled a message, consumed or produced something)
    // input_seen: the input position (source head / items written upstream) the decision "short of input" was taken on -- what the idle wait must compare with
    bool iterate(Node& nd, bool& short_of_input, uint64_t& input_seen)
    {
        bool progress = false;
        input_seen = 0;
        const int handled = nd.blk->handle_pending_messages();
        if (handled > 0) progress = true;
        short_of_input = true;
        if (!nd.from_source && !nd.in) return progress;  // a message-only block (channel_msg_receiver_cc)
        uint64_t avail = 0, out_room = 0;
        const char* in_ptr = nullptr;
        size_t in_item = d_item;
        {
            std::lock_guard<std::mutex> lk(d_mu);
            nd.st.messages += static_cast<uint64_t>(handled);
            if (nd.from_source)
                {
                    input_seen = d_head.load();
                    avail = std::min<uint64_t>(input_seen - nd.read, d_buffer);
                    in_ptr = d_stream + nd.read * d_item;
                }
            else
                {
                    input_seen = nd.in->written;
                    avail = input_seen - nd.read;
                    in_item = nd.in->item;
                }
        }
        const bool has_out = nd.blk->output_signature()->max_streams() > 0 && nd.out;
        int noutput = 0;
        if (nd.blk->output_signature()->max_streams() > 0)
            {
                const int cap = nd.blk->max_noutput_items() > 0 ? nd.blk->max_noutput_items() : 64;
                noutput = std::min(cap, 64);
                out_room = static_cast<uint64_t>(noutput);
            }
        // forecast: how many input items the block wants for that many output items (a sink is offered whatever there is)
        gr_vector_int need(1, 1);
        if (noutput > 0)
            {
                for (; noutput >= 1; noutput--)  // as the scheduler does: fewer output items until the input on hand suffices
                    {
                        need[0] = noutput;
                        nd.blk->forecast(noutput, need);
                        if (static_cast<uint64_t>(std::max(need[0], 0)) <= avail) break;
                    }
                if (noutput < 1) return progress;
            }
        else if (avail == 0)
            return progress;
        short_of_input = false;
        std::vector<char> in_copy;
        if (!nd.from_source)  // (the edge buffer may grow while the block reads: hand it a copy of what is there)
            {
                std::lock_guard<std::mutex> lk(d_mu);
                in_copy.assign(nd.in->data.begin() + static_cast<std::ptrdiff_t>(nd.read * in_item), nd.in->data.begin() + static_cast<std::ptrdiff_t>((nd.read + avail) * in_item));
                in_ptr = in_copy.data();
            }
        const size_t out_item = static_cast<size_t>(std::max(nd.blk->output_signature()->sizeof_stream_item(0), 1));
        std::vector<char> out_buf(static_cast<size_t>(std::max<uint64_t>(out_room, 1)) * out_item);
        gr_vector_int nin{static_cast<int>(std::min<uint64_t>(avail, 0x7fffffff))};
        gr_vector_const_void_star ins{static_cast<const void*>(in_ptr)};
        gr_vector_void_star outs{static_cast<void*>(out_buf.data())};
        nd.blk->consumed_last = 0;
        const int produced = nd.blk->general_work(std::max(noutput, static_cast<int>(std::min<uint64_t>(avail, 0x7fffffff)) * (noutput == 0 ? 1 : 0)), nin, ins, outs);
        const int consumed = nd.blk->consumed_last;
        nd.blk->mock_advance(produced);
        {
            std::lock_guard<std::mutex> lk(d_mu);
            nd.st.calls++;
            if (consumed == 0 && produced <= 0) nd.st.empty_calls++;
            nd.st.consumed += static_cast<uint64_t>(std::max(consumed, 0));
            nd.st.produced += static_cast<uint64_t>(std::max(produced, 0));
            nd.read += static_cast<uint64_t>(std::max(consumed, 0));
            if (nd.traced && consumed > 0) nd.trace.push_back(nd.read);
            if (produced > 0 && has_out)
                {
                    nd.out->data.insert(nd.out->data.end(), out_buf.begin(), out_buf.begin() + static_cast<std::ptrdiff_t>(static_cast<size_t>(produced) * out_item));
                    nd.out->written += static_cast<uint64_t>(produced);
                }
            if (consumed > 0 || produced > 0)
                {
                    const auto now = std::chrono::steady_clock::now();
                    if (nd.st.calls > 1 && nd.last_progress.time_since_epoch().count() != 0)
                        nd.st.longest_gap_s = std::max(nd.st.longest_gap_s, std::chrono::duration<double>(now - nd.last_progress).count());
                    nd.last_progress = now;
                    d_cv.notify_all();
                }
        }
        if (static_cast<uint64_t>(std::max(consumed, 0)) > avail) fail(nd.st.name + " consumed more than it was offered");
        return progress || consumed > 0 || produced > 0;
    }

    void node_thread(size_t index)
    {
        Node& nd = *d_nodes[index];
        int fruitless = 0;
        for (;;)
            {
                if (d_token)
                    {
                        {
                            std::unique_lock<std::mutex> lk(d_mu);
                            wait_turn(index + 1, lk);
                            if (d_stop.load()) return;
                        }
                        // the turn lasts until the block has made progress or is short of input: a block that has its input and comes back empty-handed (the HIP
                        // tracking block while the device is still working on the period) is called again, as the scheduler would -- it just may not pass the turn on
                        bool short_of_input = false, progress = false;
                        uint64_t input_seen = 0;
                        const auto t0 = std::chrono::steady_clock::now();
                        for (;;)
                            {
                                progress = iterate(nd, short_of_input, input_seen);
                                if (progress || short_of_input || d_stop.load()) break;
                                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20))
                                    {
                                        fail(nd.st.name + " has its input and makes no progress for 20 s");
                                        break;
                                    }
                                std::this_thread::yield();
                            }
                        std::lock_guard<std::mutex> lk(d_mu);
                        nd.idle = !progress;
                        if (progress) d_round_progress = true;
                        pass_turn_locked();
                        continue;
                    }
                bool short_of_input = false;
                uint64_t seen = 0;
                const bool progress = iterate(nd, short_of_input, seen);
                if (d_stop.load()) return;
                if (progress)
                    {
                        fruitless = 0;
                        continue;
                    }
                std::unique_lock<std::mutex> lk(d_mu);
                if (!short_of_input)
                    {
                        // READY_NO_OUTPUT: GNU Radio calls the block again at once (tpb_thread_body.cc); after a burst of fruitless calls be polite to the host
                        lk.unlock();
                        if (++fruitless > 64) std::this_thread::sleep_for(std::chrono::microseconds(20));
                        else std::this_thread::yield();
                        continue;
                    }
                fruitless = 0;
                nd.idle = true;
                nd.idle_seen = seen;
                d_cv.notify_all();
                // (a block that went idle with some items on hand is woken when more arrive or a message comes, not while the same items sit there)
                d_cv.wait(lk, [&] {
                    if (d_stop.load() || nd.blk->has_pending_messages()) return true;
                    return (nd.from_source ? d_head.load() : (nd.in ? nd.in->written : 0)) != seen;
                });
                nd.idle = false;
                if (d_stop.load()) return;
            }
    }

    const char* d_stream;
    size_t d_n, d_item, d_buffer, d_chunk;
    bool d_token;
    gr::top_block_sptr d_top;
    std::shared_ptr<Source_Block> d_source;
    std::vector<std::unique_ptr<Node>> d_nodes;
    std::vector<std::thread> d_threads;
    std::mutex d_mu;
    std::condition_variable d_cv;
    std::atomic<uint64_t> d_head{0};
    std::atomic<bool> d_stop{true}, d_failed{false};
    std::string d_failure;
    size_t d_turn{0};
    bool d_round_progress{false};
    uint64_t d_stuck_rounds{0};
    std::function<bool()> d_control, d_control_pending;
    std::function<void(uint64_t)> d_on_release;
    std::vector<gr::basic_block_sptr> d_traced;
};
#endif
