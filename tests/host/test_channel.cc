// "Drops into a Channel unchanged" (north_star), demonstrated: the reference's OWN Channel, ChannelFsm and channel_msg_receiver_cc -- channel.cc, channel_fsm.cc,
// channel_msg_receiver_cc.cc compiled from /root/reference where they lie (oracle/Makefile -> oracle/_ref/libgnsssdr_ref_chan.so) -- drive the HIP adapters
// (GpsL1CaPcpsAcquisitionHip + GpsL1CaDllPllTrackingHip) through whole channel lives, on a thread-per-block scheduler (tests/host/mini_flowgraph.h over
// tests/host/mock_gnuradio).  Beside it the SAME Channel class over the reference's own adapters (GpsL1CaPcpsAcquisition + GpsL1CaDllPllTracking, same library)
// runs the same stream: the checker.
//
//   acquire -> ChannelFsm::Event_valid_acquisition() called DIRECTLY from the acquisition block's thread (acq.cc:318-326) -> trk_->start_tracking() (channel_fsm.cc:190-194)
//   -> tracking -> signal removed -> lock detectors -> "events" 3 (trk.cc:1208-1221) -> channel_msg_receiver_cc (its own thread) -> Event_failed_tracking_standby
//   -> queue message (channel_fsm.cc:209-213) -> the control thread assigns a satellite and calls Channel::start_acquisition (what GNSSFlowgraph::apply_action does)
//   -> negative acquisitions ("events" 2 -> Event_failed_acquisition_no_repeat -> queue) until the signal is back -> re-acquired -> tracked to the end.
//
//   test_channel life      token scheduling (one block thread at a time, fixed round): both receivers must produce IDENTICAL event sequences (who, what, at which
//                          source position), identical Acq_delay_samples / Acq_doppler_hz / Acq_samplestamp_samples at every hand-over, the same first tracking
//                          window and the same consumed counts call for call (tracking read pointers +-1 sample where the GPU's accumulators differ in the last bits)
//   test_channel churn     free-running threads: N channels on one stream, a third of them forced to lose lock every ~200 ms (a telemetry decoder's fault message,
//                          trk.cc:757-769) and re-acquired through the FSM while the others track: start_tracking / stop_tracking of a churner quiesce the live
//                          residency the other channels share -- none of those may lose a window, an event or a symbol; prints the stall per start / stop
//   test_channel faults    (fake engine only) injected engine failures: a push fails once / for good, gsh_trk_live_take fails for one channel, a residency never
//                          reports, a dwell fails, gsh_trk_start fails: the failing channel publishes "events" 3 (tracking) or 2 (acquisition) and goes to standby,
//                          nothing throws across general_work, the other channels keep their windows, every thread joins (SURVEY section 5 "Failure detection")
// Links against libgnss_sdr_hip.so (GPU) or, as test_channel_fake, in front of tests/host/fake_gsh_engine.cc (CPU suite, ThreadSanitizer).
#include "GPS_L1_CA.h"
#include "channel.h"
#include "channel_event.h"
#include "concurrent_queue.h"
#include "dll_pll_tracking_hip.h"
#include "gnss_sdr_hip.h"
#include "gnss_signal.h"
#include "gnss_synchro.h"
#include "gps_l1_ca_pcps_acquisition_hip.h"
#include "gps_sdr_signal_replica.h"
#include "in_memory_configuration.h"
#include "mini_flowgraph.h"
#include "ref_chan_api.h"
#include "telemetry_decoder_interface.h"
#include <algorithm>
#include <any>
#include <atomic>
#include <chrono>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

extern "C" {
// present only in the *_fake build (tests/host/fake_gsh_engine.cc)
void fake_gsh_set_reference_stream(const float* iq, uint64_t n) __attribute__((weak));
uint64_t fake_gsh_push_mismatches(void) __attribute__((weak));
int fake_gsh_concurrent_handle_entries(void) __attribute__((weak));
void fake_gsh_inject_fault(int kind, long after, long count, int channel) __attribute__((weak));
long fake_gsh_fault_hits(int kind) __attribute__((weak));
void fake_gsh_clear_faults(void) __attribute__((weak));
}

namespace
{
int fails = 0;
std::mutex g_print_mu;
#define EXPECT(cond, ...)                                            \
    do                                                               \
        {                                                            \
            if (!(cond))                                             \
                {                                                    \
                    std::lock_guard<std::mutex> lk_(g_print_mu);     \
                    std::printf("FAIL %s:%d: ", __FILE__, __LINE__); \
                    std::printf(__VA_ARGS__);                        \
                    std::printf("\n");                               \
                    fails++;                                         \
                }                                                    \
        }                                                            \
    while (0)

typedef std::map<std::string, std::string> Props;
using Clock = std::chrono::steady_clock;

// ---- the stream: GPS L1 C/A satellites in noise, each present during given stretches ----------------------------------------------------------------------------
struct Sat
{
    uint32_t prn{1};
    double fd{0.0};     // carrier Doppler [Hz]
    double delay{0.0};  // sample index at which a code period starts
    float amp{0.0F};
    std::vector<std::pair<uint64_t, uint64_t>> present;  // [from, to) in samples; empty: always
};

float amp_for_cn0(double cn0_dbhz, double fs) { return static_cast<float>(std::sqrt(std::pow(10.0, cn0_dbhz / 10.0) * 2.0 / fs)); }

std::vector<std::complex<float>> synth(const std::vector<Sat>& sats, double fs, size_t n, unsigned seed, int n_threads)
{
    std::vector<std::complex<float>> x(n);
    std::vector<std::vector<float>> codes(sats.size(), std::vector<float>(1023));
    for (size_t s = 0; s < sats.size(); s++) gps_l1_ca_code_gen_float(codes[s], static_cast<int32_t>(sats[s].prn), 0);
    constexpr size_t CH = 4096;
    const size_t n_chunks = (n + CH - 1) / CH;
    std::atomic<size_t> next_chunk{0};
    auto work = [&]() {
        for (;;)
            {
                const size_t c = next_chunk.fetch_add(1);
                if (c >= n_chunks) return;
                const size_t i0 = c * CH, i1 = std::min(n, i0 + CH);
                std::mt19937 gen(seed * 1000003U + static_cast<unsigned>(c));
                std::normal_distribution<float> g(0.0F, 1.0F);
                for (size_t i = i0; i < i1; i++) x[i] = std::complex<float>(g(gen), g(gen));
                for (size_t s = 0; s < sats.size(); s++)
                    {
                        const Sat& sp = sats[s];
                        const double rate = 1.023e6 * (1.0 + sp.fd / 1575.42e6) / fs;  // chips per sample
                        const double w = 2.0 * M_PI * sp.fd / fs;
                        const double ph0 = std::fmod(w * static_cast<double>(i0), 2.0 * M_PI);
                        std::complex<double> ph(std::cos(ph0), std::sin(ph0));
                        const std::complex<double> step(std::cos(w), std::sin(w));
                        for (size_t i = i0; i < i1; i++, ph *= step)
                            {
                                bool on = sp.present.empty();
                                for (const auto& iv : sp.present) on = on || (i >= iv.first && i < iv.second);
                                if (!on) continue;
                                const double pos = rate * (static_cast<double>(i) - sp.delay);
                                const auto k = static_cast<long long>(std::floor(pos));
                                const long long per = (k >= 0) ? k / 1023 : -((-k + 1022) / 1023);
                                const auto idx = static_cast<size_t>(k - per * 1023);
                                // navigation bits: 20 code periods each, a fixed pseudo-random pattern per satellite
                                const long long bit = (per >= 0 ? per : 0) / 20;
                                const float sym = (((static_cast<unsigned long long>(bit) * 2654435761ULL + sp.prn * 40503ULL) >> 7) & 1ULL) ? 1.0F : -1.0F;
                                x[i] += sp.amp * sym * codes[s][idx] * std::complex<float>(static_cast<float>(ph.real()), static_cast<float>(ph.imag()));
                            }
                    }
            }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < std::max(n_threads, 1); t++) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
    return x;
}

// ---- a telemetry decoder that decodes nothing: the third block of a Channel (channel.cc:98-110) ------------------------------------------------------------------
class Nav_Block : public gr::block
{
public:
    Nav_Block() : gr::block("nav_stub", gr::io_signature::make(1, 1, sizeof(Gnss_Synchro)), gr::io_signature::make(1, 1, sizeof(Gnss_Synchro)))
    {
        this->message_port_register_out(pmt::mp("telemetry_to_trk"));
    }
    int general_work(int, gr_vector_int& ninput_items, gr_vector_const_void_star& input_items, gr_vector_void_star&) override
    {
        const auto* in = reinterpret_cast<const Gnss_Synchro*>(input_items[0]);
        {
            std::lock_guard<std::mutex> lk(mu);
            for (int i = 0; i < ninput_items[0]; i++) items.push_back(in[i]);
        }
        consume_each(ninput_items[0]);
        return 0;
    }
    // a decoder that finds its frames inconsistent tells the tracking block so (telemetry fault, trk.cc:757-769).  Called by the test when the stream has reached the
    // positions at which this channel is to lose its satellite; the message travels like any other: queued at the tracking block, handled on ITS thread.
    void report_fault()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            faults_sent++;
        }
        this->message_port_pub(pmt::mp("telemetry_to_trk"), pmt::make_any(std::any(1)));
    }
    std::mutex mu;
    std::vector<Gnss_Synchro> items;
    int faults_sent{0};
};

class Nav_Stub : public TelemetryDecoderInterface
{
public:
    Nav_Stub() : blk(std::make_shared<Nav_Block>()) {}
    std::string role() override { return "TelemetryDecoder_1C"; }
    std::string implementation() override { return "Nav_Stub"; }
    size_t item_size() override { return sizeof(Gnss_Synchro); }
    void connect(gr::top_block_sptr) override {}
    void disconnect(gr::top_block_sptr) override {}
    gr::basic_block_sptr get_left_block() override { return blk; }
    gr::basic_block_sptr get_right_block() override { return blk; }
    void reset() override { resets++; }
    void set_satellite(const Gnss_Satellite&) override {}
    void set_channel(int) override {}
    std::shared_ptr<Nav_Block> blk;
    std::atomic<int> resets{0};
};

// ---- what the Channel is handed: the adapters behind thin recorders of the calls a Channel / ChannelFsm makes (the adapters themselves are untouched) ---------------
struct Handover
{
    double acq_delay_samples{0.0}, acq_doppler_hz{0.0};
    uint64_t acq_samplestamp_samples{0};
    uint32_t prn{0};
    uint64_t source_head{0};  // how far the source had got when the FSM called start_tracking
    double call_seconds{0.0};
};

class Spy_Tracking : public TrackingInterface
{
public:
    Spy_Tracking(std::shared_ptr<TrackingInterface> inner_, std::function<uint64_t()> head_) : inner(std::move(inner_)), head(std::move(head_)) {}
    std::string role() override { return inner->role(); }
    std::string implementation() override { return inner->implementation(); }
    size_t item_size() override { return inner->item_size(); }
    void connect(gr::top_block_sptr t) override { inner->connect(std::move(t)); }
    void disconnect(gr::top_block_sptr t) override { inner->disconnect(std::move(t)); }
    gr::basic_block_sptr get_left_block() override { return inner->get_left_block(); }
    gr::basic_block_sptr get_right_block() override { return inner->get_right_block(); }
    void set_channel(unsigned int c) override { inner->set_channel(c); }
    void set_gnss_synchro(Gnss_Synchro* p) override
    {
        synchro = p;
        inner->set_gnss_synchro(p);
    }
    void start_tracking() override
    {
        Handover h;
        if (synchro != nullptr)
            {
                h.acq_delay_samples = synchro->Acq_delay_samples;
                h.acq_doppler_hz = synchro->Acq_doppler_hz;
                h.acq_samplestamp_samples = synchro->Acq_samplestamp_samples;
                h.prn = synchro->PRN;
            }
        h.source_head = head();
        const auto t0 = Clock::now();
        inner->start_tracking();
        h.call_seconds = std::chrono::duration<double>(Clock::now() - t0).count();
        std::lock_guard<std::mutex> lk(mu);
        handovers.push_back(h);
    }
    void stop_tracking() override
    {
        const auto t0 = Clock::now();
        inner->stop_tracking();
        const double dt = std::chrono::duration<double>(Clock::now() - t0).count();
        std::lock_guard<std::mutex> lk(mu);
        stop_seconds.push_back(dt);
    }
    std::shared_ptr<TrackingInterface> inner;
    std::function<uint64_t()> head;
    Gnss_Synchro* synchro{nullptr};
    std::mutex mu;
    std::vector<Handover> handovers;
    std::vector<double> stop_seconds;
};

struct Event
{
    int who{0}, what{0};
    uint64_t source_head{0};
    uint32_t prn{0};  // the satellite the channel was assigned when the event was handled
};

std::string events_string(const std::vector<Event>& ev, size_t limit);

// ---- a receiver: N Channels on one source, the control thread's part of GNSSFlowgraph ----------------------------------------------------------------------------
struct Receiver
{
    std::string kind;  // "hip" | "reference"
    std::shared_ptr<InMemoryConfiguration> cfg;
    Concurrent_Queue<pmt::pmt_t> queue;
    std::vector<std::shared_ptr<Channel>> ch;
    std::vector<std::shared_ptr<Nav_Stub>> nav;
    std::vector<std::shared_ptr<Spy_Tracking>> trk;
    std::vector<std::shared_ptr<AcquisitionInterface>> acq;
    std::vector<std::vector<uint32_t>> candidates;  // per channel: the satellites the control thread hands it in turn
    std::vector<size_t> next_candidate;
    std::vector<uint32_t> assigned;
    std::unique_ptr<Mini_Flowgraph> fg;
    std::mutex mu;
    std::vector<Event> events;
    std::atomic<bool> control_busy{false};
    std::atomic<bool> stop_control{false};
    std::thread control;
    bool usable{true};

    void assign_and_start(int c)
    {
        const uint32_t prn = candidates[static_cast<size_t>(c)][next_candidate[static_cast<size_t>(c)] % candidates[static_cast<size_t>(c)].size()];
        next_candidate[static_cast<size_t>(c)]++;
        assigned[static_cast<size_t>(c)] = prn;
        ch[static_cast<size_t>(c)]->set_signal(Gnss_Signal(Gnss_Satellite("GPS", prn), "1C"));  // gnss_flowgraph.cc:1853, 1986
        ch[static_cast<size_t>(c)]->start_acquisition();
    }

    // the reaction of GNSSFlowgraph::apply_action (gnss_flowgraph.cc:1807-2010) to the three channel events, without its satellite bookkeeping:
    //   0  acquisition failed (no repeat): next satellite of the channel's list, acquire again      1  acquisition succeeded: nothing to do here
    //   2  tracking lost the satellite: the channel is in standby; same list, acquire again
    bool handle_queue()
    {
        bool any = false;
        pmt::pmt_t msg;
        control_busy.store(true);
        while (queue.try_pop(msg))
            {
                any = true;
                int who = -1, what = -1;
                try
                    {
                        const auto ev = std::any_cast<channel_event_sptr>(pmt::any_ref(msg));
                        who = ev->channel_id;
                        what = ev->event_type;
                    }
                catch (const std::bad_any_cast&)
                    {
                        EXPECT(false, "%s: a queue message that is not a channel event", kind.c_str());
                        continue;
                    }
                if (who < 0 || who >= static_cast<int>(ch.size())) continue;
                {
                    std::lock_guard<std::mutex> lk(mu);
                    events.push_back({who, what, fg ? fg->head() : 0, assigned[static_cast<size_t>(who)]});
                }
                if (what == 0 || what == 2) assign_and_start(who);
            }
        control_busy.store(false);
        return any;
    }
    void control_loop()
    {
        while (!stop_control.load())
            if (!handle_queue()) std::this_thread::sleep_for(std::chrono::microseconds(100));
    }
};

Props receiver_props(long fs, const Props& extra)
{
    Props p{{"GNSS-SDR.internal_fs_sps", std::to_string(fs)},
        {"Acquisition_1C.item_type", "gr_complex"}, {"Acquisition_1C.doppler_max", "2500"}, {"Acquisition_1C.doppler_step", "250"}, {"Acquisition_1C.pfa", "0.0001"},
        {"Acquisition_1C.coherent_integration_time_ms", "1"}, {"Acquisition_1C.max_dwells", "1"}, {"Acquisition_1C.blocking", "true"},
        {"Tracking_1C.item_type", "gr_complex"}, {"Tracking_1C.pll_bw_hz", "35.0"}, {"Tracking_1C.dll_bw_hz", "2.0"}, {"Tracking_1C.early_late_space_chips", "0.5"},
        {"Tracking_1C.pull_in_time_s", "0"}, {"Tracking_1C.cn0_min", "38"}, {"Tracking_1C.max_lock_fail", "20"},
        {"Tracking_1C.hip_device", "0"}, {"Acquisition_1C.hip_device", "0"}, {"Tracking_1C.hip_register_input_buffer", "false"}};
    for (const auto& kv : extra) p[kv.first] = kv.second;
    return p;
}

std::unique_ptr<Receiver> make_receiver(const std::string& kind, const Props& props, int n_channels, const std::vector<std::vector<uint32_t>>& candidates,
    const std::vector<std::complex<float>>& x, size_t buffer_items, size_t chunk, bool token)
{
    auto r = std::make_unique<Receiver>();
    r->kind = kind;
    r->cfg = std::make_shared<InMemoryConfiguration>();
    for (const auto& kv : props) r->cfg->set_property(kv.first, kv.second);
    r->fg = std::make_unique<Mini_Flowgraph>(x.data(), x.size(), sizeof(gr_complex), buffer_items, chunk, token);
    r->candidates = candidates;
    r->next_candidate.assign(static_cast<size_t>(n_channels), 0);
    r->assigned.assign(static_cast<size_t>(n_channels), 0);
    Mini_Flowgraph* fg = r->fg.get();
    for (int c = 0; c < n_channels; c++)
        {
            std::shared_ptr<AcquisitionInterface> acq;
            std::shared_ptr<TrackingInterface> trk;
            if (kind == "hip")
                {
                    // what GNSSBlockFactory::GetAcqBlock / GetTrkBlock do for "GPS_L1_CA_PCPS_Acquisition_HIP" / "GPS_L1_CA_DLL_PLL_Tracking_HIP" 
                    acq = std::make_shared<GpsL1CaPcpsAcquisitionHip>(r->cfg.get(), "Acquisition_1C", 1, 0);
                    trk = std::make_shared<GpsL1CaDllPllTrackingHip>(r->cfg.get(), "Tracking_1C", 1, 1);
                }
            else
                {
                    acq = refchan_make_acquisition("GPS_L1_CA_PCPS_Acquisition", r->cfg.get(), "Acquisition_1C", 1, 0);
                    trk = refchan_make_tracking("GPS_L1_CA_DLL_PLL_Tracking", r->cfg.get(), "Tracking_1C", 1, 1);
                }
            if (!acq || !trk || acq->item_size() == 0 || trk->item_size() == 0)  // gnss_block_factory.cc:1048-1052: an unusable block is not wired
                {
                    EXPECT(false, "%s receiver: channel %d: unusable acquisition / tracking block", kind.c_str(), c);
                    r->usable = false;
                    return r;
                }
            auto spy = std::make_shared<Spy_Tracking>(trk, [fg] { return fg->head(); });
            auto nav = std::make_shared<Nav_Stub>();
            // the reference's Channel, as GNSSBlockFactory::GetChannel builds it (gnss_block_factory.cc:1040-1054)
            auto channel = std::make_shared<Channel>(r->cfg.get(), static_cast<uint32_t>(c), acq, spy, nav, "Channel", "1C", &r->queue);
            r->acq.push_back(acq);
            r->trk.push_back(spy);
            r->nav.push_back(nav);
            r->ch.push_back(channel);
        }
    // GNSSFlowgraph::connect (gnss_flowgraph.cc:1205-1231): every channel's acquisition and tracking blocks read the signal conditioner's output
    for (int c = 0; c < n_channels; c++)
        {
            auto& channel = r->ch[static_cast<size_t>(c)];
            channel->connect(fg->top());
            fg->top()->connect(fg->source(), 0, channel->get_left_block_trk(), 0);
            fg->top()->connect(fg->source(), 0, channel->get_left_block_acq(), 0);
        }
    return r;
}

void start_receiver(Receiver& r, bool token)
{
    if (token)
        r.fg->set_control_turn([&r] { return r.handle_queue(); });
    r.fg->set_control_pending([&r] { return !r.queue.empty() || r.control_busy.load(); });
    for (size_t c = 0; c < r.ch.size(); c++) r.assign_and_start(static_cast<int>(c));  // GNSSFlowgraph::start_acquisition_in_all_channels (roughly)
    r.fg->start();
    if (!token) r.control = std::thread([&r] { r.control_loop(); });
}

bool finish_receiver(Receiver& r, double timeout_s)
{
    const bool drained = r.fg->wait_until_drained(timeout_s);
    EXPECT(drained, "%s receiver: the flowgraph did not drain within %.0f s%s%s", r.kind.c_str(), timeout_s, r.fg->failed() ? ": " : "", r.fg->failed() ? r.fg->failure().c_str() : "");
    if (!drained)
        {
            r.fg->dump_state();
            std::lock_guard<std::mutex> lk(r.mu);
            std::printf("  events so far: %s\n", events_string(r.events, 80).c_str());
            std::fflush(stdout);
        }
    r.stop_control.store(true);
    if (r.control.joinable()) r.control.join();
    for (auto& c : r.ch) c->stop_channel();  // Event_stop_channel -> stop_tracking / stop_acquisition (channel.cc:210-221)
    r.fg->stop();
    for (auto& c : r.ch) c->disconnect(r.fg->top());
    return drained;
}

std::string events_string(const std::vector<Event>& ev, size_t limit = 40);
std::string events_string(const std::vector<Event>& ev, size_t limit)
{
    std::string s;
    for (size_t i = 0; i < ev.size() && i < limit; i++) s += std::to_string(ev[i].who) + ":" + std::to_string(ev[i].what) + "@" + std::to_string(ev[i].source_head) + " ";
    if (ev.size() > limit) s += "...";
    return s;
}

// every read pointer `a` stops at is one `b` stops at, give or take one sample; returns the fraction that are exactly b's
double compare_positions(const char* what, const std::vector<uint64_t>& a, const std::vector<uint64_t>& b, uint64_t from, size_t* compared)
{
    size_t j = 0, calls = 0, exact = 0;
    for (const uint64_t p : a)
        {
            if (p < from) continue;
            while (j < b.size() && b[j] + 1 < p) j++;
            if (j == b.size()) break;
            calls++;
            if (b[j] == p)
                exact++;
            else if (!(b[j] + 1 == p || p + 1 == b[j]))
                {
                    EXPECT(false, "%s: read pointer %llu is not a position the reference block stops at (nearest %llu)", what, static_cast<unsigned long long>(p),
                        static_cast<unsigned long long>(b[j]));
                    break;
                }
        }
    if (compared) *compared = calls;
    return calls ? static_cast<double>(exact) / static_cast<double>(calls) : 0.0;
}

void check_fake_engine(const char* what)
{
    if (fake_gsh_push_mismatches != nullptr)
        EXPECT(fake_gsh_push_mismatches() == 0, "%s: %llu pushes put samples at the wrong absolute index", what, static_cast<unsigned long long>(fake_gsh_push_mismatches()));
    if (fake_gsh_concurrent_handle_entries != nullptr)
        EXPECT(fake_gsh_concurrent_handle_entries() == 0, "%s: %d unserialised entries into an engine handle", what, fake_gsh_concurrent_handle_entries());
}

// ================================================================ one channel's life, both receivers, token scheduling =================================================
struct Life
{
    std::vector<Event> events;
    std::vector<Handover> handovers;
    std::vector<uint64_t> trk_positions, acq_positions;
    std::vector<Gnss_Synchro> items;
    int nav_resets{0};
    bool ok{false};
};

Life run_life(const std::string& kind, const Props& props, const std::vector<std::complex<float>>& x, int vlen)
{
    Life L;
    auto r = make_receiver(kind, props, 1, {{12U, 7U}}, x, static_cast<size_t>(4 * vlen), static_cast<size_t>(vlen / 2), true);
    if (!r->usable) return L;
    r->fg->enable_trace(r->ch[0]->get_left_block_trk());
    r->fg->enable_trace(r->ch[0]->get_left_block_acq());
    start_receiver(*r, true);
    L.ok = finish_receiver(*r, 600.0);
    L.events = r->events;
    L.handovers = r->trk[0]->handovers;
    L.trk_positions = r->fg->trace_of(r->ch[0]->get_left_block_trk());
    L.acq_positions = r->fg->trace_of(r->ch[0]->get_left_block_acq());
    {
        std::lock_guard<std::mutex> lk(r->nav[0]->blk->mu);
        L.items = r->nav[0]->blk->items;
    }
    L.nav_resets = r->nav[0]->resets.load();
    return L;
}

void test_channel_life()
{
    const long fs = 4000000;
    const int vlen = 4000;
    // PRN 7 present for 1.6 s, gone for 0.5 s, back until the end (3.1 s); PRN 12, the channel's first candidate, is never there
    const auto n = static_cast<size_t>(3.1 * fs);
    Sat s7;
    s7.prn = 7;
    s7.fd = 1180.0;
    s7.delay = 1357.25;
    s7.amp = amp_for_cn0(47.0, static_cast<double>(fs));
    s7.present = {{0, static_cast<uint64_t>(1.6 * fs)}, {static_cast<uint64_t>(2.1 * fs), n}};
    const auto x = synth({s7}, static_cast<double>(fs), n, 11U, std::max(2, static_cast<int>(std::thread::hardware_concurrency())));
    if (fake_gsh_set_reference_stream != nullptr) fake_gsh_set_reference_stream(reinterpret_cast<const float*>(x.data()), x.size());
    const Props props = receiver_props(fs, {});
    const auto t0 = Clock::now();
    const Life hip = run_life("hip", props, x, vlen);
    const double t_hip = std::chrono::duration<double>(Clock::now() - t0).count();
    const Life ref = run_life("reference", props, x, vlen);
    const double t_ref = std::chrono::duration<double>(Clock::now() - t0).count() - t_hip;
    check_fake_engine("channel life");
    if (!hip.ok || !ref.ok) return;

    // ---- the life has all its stations (reference receiver: the scenario is what it is meant to be)
    auto count = [](const std::vector<Event>& e, int what) { return std::count_if(e.begin(), e.end(), [what](const Event& v) { return v.what == what; }); };
    std::printf("channel life: %zu events (reference %zu): %ld / %ld acquisitions failed, %ld / %ld succeeded, %ld / %ld losses of lock; %zu / %zu hand-overs; "
                "%zu / %zu items; HIP receiver %.1f s, reference receiver %.1f s\n",
        hip.events.size(), ref.events.size(), static_cast<long>(count(hip.events, 0)), static_cast<long>(count(ref.events, 0)), static_cast<long>(count(hip.events, 1)),
        static_cast<long>(count(ref.events, 1)), static_cast<long>(count(hip.events, 2)), static_cast<long>(count(ref.events, 2)), hip.handovers.size(), ref.handovers.size(),
        hip.items.size(), ref.items.size(), t_hip, t_ref);
    EXPECT(count(ref.events, 1) == 2 && count(ref.events, 2) == 1 && count(ref.events, 0) >= 3, "the scenario did not play as designed on the reference receiver: %s",
        events_string(ref.events).c_str());
    EXPECT(!ref.events.empty() && ref.events[0].what == 0 && ref.events[0].prn == 12, "first event: PRN 12 is not in the stream, its acquisition must fail first");

    // ---- identical event sequences, at identical source positions, for identical satellites
    EXPECT(hip.events.size() == ref.events.size(), "event count %zu vs the reference receiver's %zu\n  hip: %s\n  ref: %s", hip.events.size(), ref.events.size(),
        events_string(hip.events).c_str(), events_string(ref.events).c_str());
    size_t same = 0;
    for (size_t i = 0; i < std::min(hip.events.size(), ref.events.size()); i++)
        {
            const Event &a = hip.events[i], &b = ref.events[i];
            if (a.who == b.who && a.what == b.what && a.source_head == b.source_head && a.prn == b.prn)
                same++;
            else
                {
                    EXPECT(false, "event %zu: channel %d event %d PRN %u at source %llu vs the reference receiver's channel %d event %d PRN %u at %llu", i, a.who, a.what, a.prn,
                        static_cast<unsigned long long>(a.source_head), b.who, b.what, b.prn, static_cast<unsigned long long>(b.source_head));
                    break;
                }
        }
    // ---- identical hand-overs: what the FSM's start_tracking found in Gnss_Synchro, and when
    EXPECT(hip.handovers.size() == ref.handovers.size() && hip.handovers.size() == 2, "%zu hand-overs vs %zu (2 expected)", hip.handovers.size(), ref.handovers.size());
    for (size_t i = 0; i < std::min(hip.handovers.size(), ref.handovers.size()); i++)
        {
            const Handover &a = hip.handovers[i], &b = ref.handovers[i];
            EXPECT(a.prn == b.prn && a.acq_samplestamp_samples == b.acq_samplestamp_samples && a.acq_delay_samples == b.acq_delay_samples && a.acq_doppler_hz == b.acq_doppler_hz &&
                       a.source_head == b.source_head,
                "hand-over %zu: PRN %u stamp %llu delay %.3f Doppler %.1f at source %llu vs the reference receiver's PRN %u stamp %llu delay %.3f Doppler %.1f at %llu", i, a.prn,
                static_cast<unsigned long long>(a.acq_samplestamp_samples), a.acq_delay_samples, a.acq_doppler_hz, static_cast<unsigned long long>(a.source_head), b.prn,
                static_cast<unsigned long long>(b.acq_samplestamp_samples), b.acq_delay_samples, b.acq_doppler_hz, static_cast<unsigned long long>(b.source_head));
            std::printf("  hand-over %zu: PRN %u, Acq_samplestamp_samples %llu, Acq_delay_samples %.1f, Acq_doppler_hz %.0f, FSM -> start_tracking at source position %llu (%.0f us in the call)\n",
                i, a.prn, static_cast<unsigned long long>(a.acq_samplestamp_samples), a.acq_delay_samples, a.acq_doppler_hz, static_cast<unsigned long long>(a.source_head), a.call_seconds * 1e6);
            // the delay is the satellite's: its code periods start at delay + k * P, P = 4000 / (1 + fd / f_carrier) samples (the code Doppler moves them 6 samples in 2 s)
            const double P = 4000.0 / (1.0 + 1180.0 / 1575.42e6);
            const double code_start = std::fmod(std::fmod(1357.25 - static_cast<double>(a.acq_samplestamp_samples), P) + P, P);
            EXPECT(std::fabs(std::remainder(a.acq_delay_samples - code_start, P)) < 3.0 && std::fabs(a.acq_doppler_hz - 1180.0) <= 250.0, "hand-over %zu: delay %.1f (code start %.1f), Doppler %.0f",
                i, a.acq_delay_samples, code_start, a.acq_doppler_hz);
        }
    // ---- consumed counts: the acquisition blocks call for call, the tracking blocks' read pointers (+-1 sample where the accumulators differ in their last bits)
    EXPECT(hip.acq_positions == ref.acq_positions, "the acquisition blocks' read pointers differ (%zu vs %zu calls)", hip.acq_positions.size(), ref.acq_positions.size());
    size_t compared = 0;
    const double exact = compare_positions("channel life, tracking block", hip.trk_positions, ref.trk_positions, 0, &compared);
    EXPECT(compared + 8 >= ref.trk_positions.size() && exact >= 0.999, "tracking read pointers: %zu of the reference's %zu compared, %.4f %% exact", compared, ref.trk_positions.size(), 100.0 * exact);
    // the first tracking window after each hand-over: the read pointer the pull-in leaves (trk.cc:1949-1978) -- exactly the reference's
    for (size_t i = 0; i < std::min(hip.handovers.size(), ref.handovers.size()); i++)
        {
            auto first_after = [](const std::vector<uint64_t>& pos, uint64_t head, int vl) {
                // the first two read pointers that are NOT whole-buffer standby consumption: pull-in alignment, then the first period
                for (size_t k = 0; k + 1 < pos.size(); k++)
                    if (pos[k] >= head - std::min<uint64_t>(head, static_cast<uint64_t>(8 * vl)) && pos[k + 1] - pos[k] != 0 && (pos[k + 1] - pos[k]) % static_cast<uint64_t>(vl / 2) != 0) return pos[k + 1];
                return static_cast<uint64_t>(0);
            };
            const uint64_t a = first_after(hip.trk_positions, hip.handovers[i].source_head, vlen), b = first_after(ref.trk_positions, ref.handovers[i].source_head, vlen);
            EXPECT(a == b && a != 0, "hand-over %zu: first tracking window at %llu vs the reference receiver's %llu", i, static_cast<unsigned long long>(a), static_cast<unsigned long long>(b));
        }
    // ---- the items the telemetry decoder received: same count, same timing, same flags; values as close as everywhere else
    EXPECT(hip.items.size() == ref.items.size() && !ref.items.empty(), "%zu items vs the reference receiver's %zu", hip.items.size(), ref.items.size());
    if (std::getenv("CHANNEL_DEBUG") != nullptr)
        for (const auto* L : {&hip, &ref})
            {
                std::printf("  trk read pointers:");
                for (size_t i = 0; i < std::min<size_t>(14, L->trk_positions.size()); i++) std::printf(" %llu", static_cast<unsigned long long>(L->trk_positions[i]));
                std::printf("\n  items:");
                for (size_t i = 0; i < L->items.size(); i++)
                    if (i < 6 || i + 3 > L->items.size() || !L->items[i].Flag_valid_symbol_output)
                        std::printf(" [%zu] %llu v%d cn0 %.1f", i, static_cast<unsigned long long>(L->items[i].Tracking_sample_counter), L->items[i].Flag_valid_symbol_output, L->items[i].CN0_dB_hz);
                std::printf("\n");
            }
    size_t valid = 0, lost = 0;
    double worst_prompt = 0.0, worst_doppler = 0.0;
    for (size_t i = 0; i < std::min(hip.items.size(), ref.items.size()); i++)
        {
            const Gnss_Synchro &a = hip.items[i], &b = ref.items[i];
            const bool near = a.Tracking_sample_counter == b.Tracking_sample_counter || a.Tracking_sample_counter + 1 == b.Tracking_sample_counter || b.Tracking_sample_counter + 1 == a.Tracking_sample_counter;
            if (!(near && a.Flag_valid_symbol_output == b.Flag_valid_symbol_output && a.PRN == b.PRN && a.Channel_ID == b.Channel_ID && a.System == b.System &&
                    a.Acq_samplestamp_samples == b.Acq_samplestamp_samples && a.Acq_delay_samples == b.Acq_delay_samples && a.Acq_doppler_hz == b.Acq_doppler_hz))
                {
                    EXPECT(false, "item %zu: sample counter %llu / %llu, valid %d / %d, PRN %u / %u", i, static_cast<unsigned long long>(a.Tracking_sample_counter),
                        static_cast<unsigned long long>(b.Tracking_sample_counter), a.Flag_valid_symbol_output, b.Flag_valid_symbol_output, a.PRN, b.PRN);
                    break;
                }
            if (!a.Flag_valid_symbol_output)
                {
                    lost++;
                    continue;
                }
            valid++;
            worst_prompt = std::max(worst_prompt, std::fabs(a.Prompt_I - b.Prompt_I) / std::max(1.0, std::fabs(b.Prompt_I)));
            worst_doppler = std::max(worst_doppler, std::fabs(a.Carrier_Doppler_hz - b.Carrier_Doppler_hz));
        }
    EXPECT(valid >= 20 && lost == 1, "%zu valid symbols, %zu loss-of-lock items (1 expected: trk.cc:2009-2014)", valid, lost);
    EXPECT(worst_prompt < 2e-2 && worst_doppler < 1.0, "symbols: Prompt_I differs by %.3e, Doppler by %.3f Hz", worst_prompt, worst_doppler);
    EXPECT(hip.nav_resets == ref.nav_resets && ref.nav_resets == static_cast<int>(count(ref.events, 0) + count(ref.events, 2)) + 1,
        "the FSM reset the telemetry decoder %d times (reference receiver %d)", hip.nav_resets, ref.nav_resets);
    std::printf("channel life: %zu of %zu events identical (who, what, satellite, source position); acquisition read pointers identical (%zu calls); tracking read pointers %.4f %% exact "
                "over %zu calls, the rest one sample off; %zu symbols, worst |dPrompt_I| rel %.2e, worst |dDoppler| %.3f Hz\n",
        same, ref.events.size(), ref.acq_positions.size(), 100.0 * exact, compared, valid, worst_prompt, worst_doppler);
}

// ================================================================ many channels, free-running threads, a third of them churning =====================================
struct Churn_Result
{
    bool ok{false};
    std::vector<std::vector<Event>> events;  // per channel
    std::vector<std::vector<uint64_t>> trk_positions;
    std::vector<size_t> valid_symbols, lost_items;
    std::vector<std::vector<Handover>> handovers;
    std::vector<double> start_calls, stop_calls;  // seconds inside start_tracking / stop_tracking, all channels
    std::vector<double> longest_gap;              // per channel: the tracking block's longest wall-clock time between two calls that consumed
    std::vector<int> faults_sent;
    std::vector<std::string> errors;  // HIP receiver: the tracking blocks' last engine errors
    std::vector<uint64_t> final_read;  // the tracking blocks' read pointers when the flowgraph had drained
    double seconds{0.0};
};

// fault_period_samples > 0: churner c's decoder reports a fault whenever the source passes (k + 1) * fault_period_samples + c * fault_period_samples / n_churn
Churn_Result run_churn(const std::string& kind, const Props& props, const std::vector<std::complex<float>>& x, int vlen, int n_channels, int n_churn, uint64_t fault_period_samples,
    double timeout_s = 300.0)
{
    Churn_Result R;
    std::vector<std::vector<uint32_t>> cand;
    for (int c = 0; c < n_channels; c++) cand.push_back({static_cast<uint32_t>(c + 1)});
    auto r = make_receiver(kind, props, n_channels, cand, x, static_cast<size_t>(8 * vlen), static_cast<size_t>(vlen / 2), false);
    if (!r->usable) return R;
    for (int c = 0; c < n_channels; c++) r->fg->enable_trace(r->ch[static_cast<size_t>(c)]->get_left_block_trk());
    if (n_churn > 0 && fault_period_samples > 0)
        {
            auto next_fault = std::make_shared<std::vector<uint64_t>>();
            for (int c = 0; c < n_churn; c++) next_fault->push_back(fault_period_samples + static_cast<uint64_t>(c) * fault_period_samples / static_cast<uint64_t>(n_churn));
            Receiver* rp = r.get();
            r->fg->set_on_release([rp, next_fault, fault_period_samples](uint64_t head) {
                for (size_t c = 0; c < next_fault->size(); c++)
                    if (head >= (*next_fault)[c])
                        {
                            (*next_fault)[c] += fault_period_samples;
                            rp->nav[c]->blk->report_fault();
                        }
            });
        }
    const auto t0 = Clock::now();
    start_receiver(*r, false);
    R.ok = finish_receiver(*r, timeout_s);
    R.seconds = std::chrono::duration<double>(Clock::now() - t0).count();
    if (kind == "hip")  // what a block gave a channel up for, if it did
        for (int c = 0; c < n_channels; c++)
            if (auto hip = std::dynamic_pointer_cast<DllPllTrackingHip>(r->trk[static_cast<size_t>(c)]->inner))
                if (hip->block() && !hip->block()->last_error().empty()) R.errors.push_back("channel " + std::to_string(c) + ": " + hip->block()->last_error());
    R.events.assign(static_cast<size_t>(n_channels), {});
    for (const Event& e : r->events) R.events[static_cast<size_t>(e.who)].push_back(e);
    const auto stats = r->fg->stats();
    for (int c = 0; c < n_channels; c++)
        {
            const auto blk = r->ch[static_cast<size_t>(c)]->get_left_block_trk();
            R.trk_positions.push_back(r->fg->trace_of(blk));
            R.final_read.push_back(r->fg->items_read(blk));
            if (std::getenv("GSH_TEST_LOG") != nullptr && R.final_read.back() + static_cast<uint64_t>(3 * vlen) < x.size())
                {
                    std::printf("  (channel %d: read pointer %llu of %zu when the flowgraph had drained)\n", c, static_cast<unsigned long long>(R.final_read.back()), x.size());
                    r->fg->dump_state();
                }
            size_t v = 0, l = 0;
            {
                std::lock_guard<std::mutex> lk(r->nav[static_cast<size_t>(c)]->blk->mu);
                for (const auto& it : r->nav[static_cast<size_t>(c)]->blk->items) (it.Flag_valid_symbol_output ? v : l)++;
                R.faults_sent.push_back(r->nav[static_cast<size_t>(c)]->blk->faults_sent);
            }
            R.valid_symbols.push_back(v);
            R.lost_items.push_back(l);
            R.handovers.push_back(r->trk[static_cast<size_t>(c)]->handovers);
            for (const auto& h : r->trk[static_cast<size_t>(c)]->handovers) R.start_calls.push_back(h.call_seconds);
            for (const double s : r->trk[static_cast<size_t>(c)]->stop_seconds) R.stop_calls.push_back(s);
            double gap = 0.0;
            const std::string name = blk->name() + "#" + std::to_string(blk->unique_id());
            for (const auto& st : stats)
                if (st.name == name) gap = st.longest_gap_s;
            R.longest_gap.push_back(gap);
        }
    return R;
}

std::vector<std::complex<float>> churn_stream(long fs, int vlen, int n_channels, double seconds, int threads)
{
    std::vector<Sat> sats;
    std::mt19937 gen(5);
    for (int c = 0; c < n_channels; c++)
        {
            Sat s;
            s.prn = static_cast<uint32_t>(c + 1);
            s.fd = -2200.0 + 4400.0 * c / std::max(1, n_channels - 1) + 13.0 * (c % 3);
            s.delay = static_cast<double>(gen() % static_cast<unsigned>(vlen)) + 0.25 * (c % 4);
            s.amp = amp_for_cn0(46.0, static_cast<double>(fs));
            sats.push_back(s);
        }
    return synth(sats, static_cast<double>(fs), static_cast<size_t>(seconds * static_cast<double>(fs)), 23U, threads);
}

void test_channel_churn(int n_channels, int n_churn, double seconds, bool with_reference)
{
    const long fs = 4000000;
    const int vlen = 4000;
    const int threads = std::max(2, static_cast<int>(std::thread::hardware_concurrency()));
    const auto x = churn_stream(fs, vlen, n_channels, seconds, threads);
    if (fake_gsh_set_reference_stream != nullptr) fake_gsh_set_reference_stream(reinterpret_cast<const float*>(x.data()), x.size());
    // every churner's telemetry decoder reports a fault every 200 ms of stream (staggered over the churners): forced loss of lock (trk.cc:757-769), "events" 3, standby,
    // re-acquisition through the FSM, start_tracking into the residency the other channels share
    const uint64_t fault_period = static_cast<uint64_t>(0.2 * static_cast<double>(fs));
    const Props props = receiver_props(fs, {{"Tracking_1C.cn0_min", "30"}, {"Tracking_1C.max_lock_fail", "50"}});
    const Churn_Result hip = run_churn("hip", props, x, vlen, n_channels, n_churn, fault_period);
    check_fake_engine("churn");
    if (!hip.ok) return;
    for (const auto& e : hip.errors) std::printf("churn: engine error seen by %s\n", e.c_str());
    EXPECT(hip.errors.empty(), "%zu tracking blocks gave a channel up for an engine error (no failure was injected)", hip.errors.size());
    auto pct = [](std::vector<double> v, double q) {
        if (v.empty()) return 0.0;
        std::sort(v.begin(), v.end());
        return v[std::min(v.size() - 1, static_cast<size_t>(q * static_cast<double>(v.size())))];
    };
    size_t restarts = 0;
    for (int c = 0; c < n_churn; c++) restarts += hip.handovers[static_cast<size_t>(c)].size();
    std::printf("churn: %d channels on one %ld Msps stream (%.1f s of signal in %.1f s), %d of them churning: %zu hand-overs of the churners; start_tracking calls: median %.0f us, "
                "worst %.0f us (%zu calls); stop_tracking calls: median %.0f us, worst %.0f us (%zu calls)\n",
        n_channels, fs / 1000000, seconds, hip.seconds, n_churn, restarts, pct(hip.start_calls, 0.5) * 1e6, pct(hip.start_calls, 1.0) * 1e6, hip.start_calls.size(),
        pct(hip.stop_calls, 0.5) * 1e6, pct(hip.stop_calls, 1.0) * 1e6, hip.stop_calls.size());
    // ---- the churners went round the whole circle several times, each time through the FSM
    for (int c = 0; c < n_churn; c++)
        {
            const auto& ev = hip.events[static_cast<size_t>(c)];
            const auto wins = std::count_if(ev.begin(), ev.end(), [](const Event& e) { return e.what == 1; });
            const auto losses = std::count_if(ev.begin(), ev.end(), [](const Event& e) { return e.what == 2; });
            // (a fault reported while the channel is being re-acquired finds no tracking to stop: start_tracking clears it, trk.cc:793-866.  And a hand-over that finds the
            //  tracking block more than a code period behind the acquisition's stamp is dropped 20 periods later by the reference's own arithmetic -- the time-limit test of
            //  trk.cc:2002 on a wrapped unsigned difference, pinned in test_tracking_adapters.cc -- and acquired again: a loss without a fault)
            EXPECT(hip.faults_sent[static_cast<size_t>(c)] >= 2 && losses >= 1 && losses + 1 >= hip.faults_sent[static_cast<size_t>(c)] && wins >= losses && wins <= losses + 1,
                "churning channel %d: %d telemetry faults sent, %ld losses of lock, %ld acquisitions: %s", c, hip.faults_sent[static_cast<size_t>(c)], static_cast<long>(losses),
                static_cast<long>(wins), events_string(ev, 24).c_str());
            EXPECT(hip.lost_items[static_cast<size_t>(c)] == static_cast<size_t>(losses), "churning channel %d: %zu loss-of-lock items for %ld losses", c, hip.lost_items[static_cast<size_t>(c)],
                static_cast<long>(losses));
            // a legal walk through ChannelFsm's states: 1 (tracking) and 2 (standby) alternate, failures (0) only between a 2 / the start and a 1
            int state = 1;  // acquiring
            for (const Event& e : ev)
                {
                    const bool legal = (e.what == 1 && state == 1) || (e.what == 0 && state == 1) || (e.what == 2 && state == 2);
                    EXPECT(legal, "churning channel %d: event %d in FSM state %d: %s", c, e.what, state, events_string(ev, 24).c_str());
                    if (!legal) break;
                    state = (e.what == 1) ? 2 : 1;
                }
        }
    // ---- the steady channels never noticed: no loss of lock once they track, every window taken.  (The one loss a steady channel may see is the reference's own: a hand-over
    // that finds the tracking block more than a code period behind the acquisition's stamp is dropped when the C/N0 buffer has its 20 prompts -- trk.cc:2002 on a wrapped
    // unsigned difference, pinned to the reference block in test_tracking_adapters.cc -- and acquired again.  Whether a free-running acquisition thread gets that far ahead is
    // the scheduler's choice, so such a drop within 40 code periods of its hand-over is counted, not failed; anything later is a failure.)
    double worst_gap = 0.0;
    size_t early_drops = 0;
    std::vector<uint64_t> steady_from(static_cast<size_t>(n_channels), 0);
    for (int c = n_churn; c < n_channels; c++)
        {
            const auto& ev = hip.events[static_cast<size_t>(c)];
            size_t drops = 0;
            bool legal = ev.size() >= 1 && ev.back().what == 1;
            uint64_t last_win = 0;
            for (const Event& e : ev)
                {
                    if (e.what == 1) last_win = e.source_head;
                    if (e.what == 2)
                        {
                            drops++;
                            if (e.source_head > last_win + static_cast<uint64_t>(40 * vlen)) legal = false;
                        }
                }
            EXPECT(legal, "steady channel %d: %s", c, events_string(ev, 12).c_str());
            EXPECT(hip.lost_items[static_cast<size_t>(c)] == drops, "steady channel %d published %zu loss-of-lock items for %zu hand-overs dropped by the time limit", c,
                hip.lost_items[static_cast<size_t>(c)], drops);
            early_drops += drops;
            steady_from[static_cast<size_t>(c)] = drops ? last_win : 0;
            worst_gap = std::max(worst_gap, hip.longest_gap[static_cast<size_t>(c)]);
            // every window taken: from the first period of the tracking that lasted, the read pointer advances by one code period (+-1 sample) per call, to the end of the stream
            const auto& pos = hip.trk_positions[static_cast<size_t>(c)];
            size_t k = 0;
            while (k + 1 < pos.size() && pos[k] < steady_from[static_cast<size_t>(c)]) k++;
            const size_t k0 = k;
            while (k + 1 < pos.size() && (pos[k + 1] - pos[k] < static_cast<uint64_t>(vlen - 2) || pos[k + 1] - pos[k] > static_cast<uint64_t>(vlen + 2) || k < k0 + 8)) k++;
            size_t odd = 0;
            for (size_t i = k; i + 1 < pos.size(); i++)
                if (pos[i + 1] - pos[i] < static_cast<uint64_t>(vlen - 2) || pos[i + 1] - pos[i] > static_cast<uint64_t>(vlen + 2)) odd++;
            EXPECT(odd == 0 && !pos.empty() && pos.back() + static_cast<uint64_t>(3 * vlen) >= x.size(), "steady channel %d: %zu read-pointer steps that are not one code period; stopped at %llu of %zu", c, odd,
                pos.empty() ? 0ULL : static_cast<unsigned long long>(pos.back()), x.size());
        }
    if (early_drops) std::printf("churn: %zu hand-overs of steady channels found the tracking block more than a code period behind the stamp and were dropped by the time limit (trk.cc:2002)\n", early_drops);
    std::printf("churn: the %d steady channels: no loss of lock once tracking, every code period taken to the end of the stream; their longest wall-clock pause between two "
                "periods: %.1f ms\n",
        n_channels - n_churn, worst_gap * 1e3);
    if (!with_reference) return;
    // ---- the same receiver over the reference's blocks: its steady channels stop at the same read pointers (the moment of each acquisition differs with the threads'
    // timing, the code periods tracked do not)
    Churn_Result ref = run_churn("reference", props, x, vlen, n_channels, n_churn, fault_period);
    if (!ref.ok) return;
    // (Round 6, profiles/ab/r06/session53.txt: on a loaded host -- 16 busy processes beside the receivers' hundred threads -- the REFERENCE receiver of this harness now and
    //  then ends with a steady channel that was acquired, kept every code period (its read pointers pass the comparison below) and yet handed its decoder not one valid
    //  symbol: 2 of 12 loaded runs, 1 of ~8 suite runs on an idle box.  That is the yardstick failing, not the engine; a comparison needs a usable reference run, so a
    //  reference run with such a channel is repeated ONCE, loudly.  The HIP receiver's run is never repeated.)
    {
        int silent = -1;
        for (int c = n_churn; c < n_channels && silent < 0; c++)
            if (ref.valid_symbols[static_cast<size_t>(c)] == 0 && hip.valid_symbols[static_cast<size_t>(c)] > 80 && !ref.events[static_cast<size_t>(c)].empty() && ref.events[static_cast<size_t>(c)].back().what == 1) silent = c;
        if (silent >= 0)
            {
                std::printf("churn: the reference receiver's steady channel %d was acquired (%s) and produced no valid symbol (the HIP receiver's: %zu): reference run repeated once\n", silent,
                    events_string(ref.events[static_cast<size_t>(silent)], 12).c_str(), hip.valid_symbols[static_cast<size_t>(silent)]);
                ref = run_churn("reference", props, x, vlen, n_channels, n_churn, fault_period);
                if (!ref.ok) return;
            }
    }
    size_t total = 0;
    double worst_exact = 1.0;
    for (int c = n_churn; c < n_channels; c++)
        {
            const auto &a = hip.trk_positions[static_cast<size_t>(c)], &b = ref.trk_positions[static_cast<size_t>(c)];
            // compare from where both are tracking (past both pull-ins)
            const auto &hh = hip.handovers[static_cast<size_t>(c)], &rh = ref.handovers[static_cast<size_t>(c)];
            const uint64_t from = std::max(hh.empty() ? 0 : hh.back().source_head, rh.empty() ? 0 : rh.back().source_head) + static_cast<uint64_t>(16 * vlen);
            // (free-running threads: the two receivers acquire a satellite at different moments -- a failed first dwell, a hand-over dropped by the time limit -- and the
            //  one that started later has fewer 20 ms symbols by the difference)
            const uint64_t h_last = hh.empty() ? 0 : hh.back().source_head, r_last = rh.empty() ? 0 : rh.back().source_head;
            const size_t late = static_cast<size_t>((h_last > r_last ? h_last - r_last : r_last - h_last) / static_cast<uint64_t>(20 * vlen)) + 2;
            size_t compared = 0;
            const double exact = compare_positions(("churn, steady channel " + std::to_string(c)).c_str(), a, b, from, &compared);
            // (free-running threads: the two receivers acquire at different moments, so the two loops start from different estimates and carry code phases a few
            //  hundredths of a sample apart: a window boundary falls on the other side of an integer one period in ten.  EVERY pointer is within one sample --
            //  compare_positions fails otherwise; exact equality is the token-scheduled test's claim, where both receivers acquire at the same sample)
            EXPECT(compared > 100 && exact >= 0.5, "steady channel %d: %zu read pointers compared with the reference receiver's, %.3f %% exact", c, compared, 100.0 * exact);
            total += compared;
            worst_exact = std::min(worst_exact, exact);
            EXPECT(hip.valid_symbols[static_cast<size_t>(c)] + 60 + late >= ref.valid_symbols[static_cast<size_t>(c)] && ref.valid_symbols[static_cast<size_t>(c)] + 60 + late >= hip.valid_symbols[static_cast<size_t>(c)],
                "steady channel %d: %zu symbols vs the reference receiver's %zu (last hand-overs at source positions %llu and %llu; reference events %s)", c,
                hip.valid_symbols[static_cast<size_t>(c)], ref.valid_symbols[static_cast<size_t>(c)], static_cast<unsigned long long>(h_last), static_cast<unsigned long long>(r_last),
                events_string(ref.events[static_cast<size_t>(c)], 12).c_str());
        }
    std::printf("churn: steady channels against the reference receiver (%.1f s): %zu read pointers compared, at least %.3f %% of a channel's exactly the reference block's, the rest one sample off\n",
        ref.seconds, total, 100.0 * worst_exact);
}

// ================================================================ injected engine failures (fake engine) ===========================================================
enum
{
    FAULT_PUSH = 0,
    FAULT_LIVE_TAKE = 1,
    FAULT_RESIDENCY = 2,
    FAULT_ACQ_DWELL = 3,
    FAULT_TRK_START = 4,
    FAULT_RUN = 5
};

void fault_case(const char* name, int kind, long after, long count, int channel, const std::vector<std::complex<float>>& x, int n_channels, const Props& extra,
    const std::function<void(const Churn_Result&, long hits)>& check)
{
    const long fs = 4000000;
    fake_gsh_clear_faults();
    fake_gsh_inject_fault(kind, after, count, channel);
    const auto t0 = Clock::now();
    Props props = receiver_props(fs, {{"Tracking_1C.cn0_min", "30"}, {"Tracking_1C.max_lock_fail", "50"}});
    for (const auto& kv : extra) props[kv.first] = kv.second;
    Churn_Result r;
    bool threw = false;
    try
        {
            r = run_churn("hip", props, x, 4000, n_channels, 0, 0, 240.0);
        }
    catch (const std::exception& e)
        {
            threw = true;
            EXPECT(false, "%s: an exception crossed the flowgraph: %s", name, e.what());
        }
    const long hits = fake_gsh_fault_hits(kind);
    fake_gsh_clear_faults();
    const double dt = std::chrono::duration<double>(Clock::now() - t0).count();
    EXPECT(r.ok && !threw, "%s: the receiver did not come to its end (a thread hangs?)", name);
    if (!r.ok) return;
    size_t losses = 0, wins = 0, fails_acq = 0;
    for (const auto& ev : r.events)
        for (const Event& e : ev) (e.what == 2 ? losses : e.what == 1 ? wins : fails_acq)++;
    std::printf("fault \"%s\": %ld injected failures -> %zu losses of lock, %zu failed acquisitions, %zu successful acquisitions over %d channels, every thread joined (%.1f s)\n", name, hits,
        losses, fails_acq, wins, n_channels, dt);
    check(r, hits);
    check_fake_engine(name);
}

void test_faults()
{
    if (fake_gsh_inject_fault == nullptr)
        {
            std::printf("faults: only with the fake engine (tests/host/test_channel_fake)\n");
            return;
        }
    const long fs = 4000000;
    const int n_channels = 6;
    const int threads = std::max(2, static_cast<int>(std::thread::hardware_concurrency()));
    const auto x = churn_stream(fs, 4000, n_channels, 1.2, threads);
    fake_gsh_set_reference_stream(reinterpret_cast<const float*>(x.data()), x.size());
    auto losses_of = [](const Churn_Result& r, int c) { return std::count_if(r.events[static_cast<size_t>(c)].begin(), r.events[static_cast<size_t>(c)].end(), [](const Event& e) { return e.what == 2; }); };
    auto wins_of = [](const Churn_Result& r, int c) { return std::count_if(r.events[static_cast<size_t>(c)].begin(), r.events[static_cast<size_t>(c)].end(), [](const Event& e) { return e.what == 1; }); };
    auto tracked_to_the_end = [&](const Churn_Result& r, int c) { return !r.trk_positions[static_cast<size_t>(c)].empty() && r.trk_positions[static_cast<size_t>(c)].back() + 3 * 4000 >= x.size() && wins_of(r, c) == losses_of(r, c) + 1; };

    auto where_it_stopped = [&](const Churn_Result& r, int c) {
        const auto& p = r.trk_positions[static_cast<size_t>(c)];
        return (p.empty() ? std::string("no period taken") : std::to_string(p.size()) + " periods, the last at " + std::to_string(p.back())) + " of " + std::to_string(x.size()) + ", read pointer at the end " +
               std::to_string(r.final_read[static_cast<size_t>(c)]) + (static_cast<size_t>(c) < r.errors.size() && !r.errors[static_cast<size_t>(c)].empty() ? "; last engine error: " + r.errors[static_cast<size_t>(c)] : std::string());
    };
    // 1. one push fails, 40 pushes into the run: the block that was pushing drops its channel ("events" 3: trk.cc:1208-1221's message), the FSM re-acquires it; the ring is
    //    whole again with the next block's push; every channel is tracking at the end
    fault_case("one push fails", FAULT_PUSH, 40, 1, -1, x, n_channels, {}, [&](const Churn_Result& r, long hits) {
        EXPECT(hits == 1, "the injected push failure was hit %ld times", hits);
        long losses = 0;
        for (int c = 0; c < n_channels; c++)
            {
                losses += losses_of(r, c);
                EXPECT(tracked_to_the_end(r, c), "channel %d is not tracking at the end of the stream: %s(%s)", c, events_string(r.events[static_cast<size_t>(c)], 16).c_str(), where_it_stopped(r, c).c_str());
            }
        EXPECT(losses >= 1 && losses <= 2, "%ld channels dropped for one failed push (the block that pushed, and at most the one that found the ring behind)", losses);
    });
    // 2. every push fails from then on: no channel can track any more; each goes round acquisition -> start -> "events" 3 until the stream ends.  Nothing throws, nothing hangs.
    fault_case("every push fails", FAULT_PUSH, 40, -1, -1, x, n_channels, {}, [&](const Churn_Result& r, long hits) {
        EXPECT(hits > 10, "only %ld pushes failed", hits);
        for (int c = 0; c < n_channels; c++)
            EXPECT(losses_of(r, c) >= 1 && r.lost_items[static_cast<size_t>(c)] == 0, "channel %d: %ld losses of lock, %zu loss-of-lock ITEMS (an engine failure has no symbol to flag)", c,
                static_cast<long>(losses_of(r, c)), r.lost_items[static_cast<size_t>(c)]);
    });
    // 3. gsh_trk_live_take fails for the device channel 2 of the group, once, 300 takes in: that channel alone drops and comes back
    fault_case("a live take fails", FAULT_LIVE_TAKE, 300, 1, 2, x, n_channels, {}, [&](const Churn_Result& r, long hits) {
        EXPECT(hits == 1, "the injected take failure was hit %ld times", hits);
        long losses = 0;
        for (int c = 0; c < n_channels; c++)
            {
                losses += losses_of(r, c);
                EXPECT(tracked_to_the_end(r, c), "channel %d is not tracking at the end of the stream: %s(%s)", c, events_string(r.events[static_cast<size_t>(c)], 16).c_str(), where_it_stopped(r, c).c_str());
            }
        EXPECT(losses == 1, "%ld channels dropped for one failed take", losses);
    });
    // 4. a residency is queued and never reports (a hung device): after Tracking_1C.hip_record_timeout_ms without a record for a resident window the blocks give their
    //    channels up; the stop that follows quiesces the residency, the next one works again
    fault_case("a residency never reports", FAULT_RESIDENCY, 3, 1, -1, x, n_channels, {{"Tracking_1C.hip_record_timeout_ms", "150"}}, [&](const Churn_Result& r, long hits) {
        EXPECT(hits == 1, "the injected dead residency was hit %ld times", hits);
        long losses = 0;
        for (int c = 0; c < n_channels; c++)
            {
                losses += losses_of(r, c);
                EXPECT(tracked_to_the_end(r, c), "channel %d is not tracking at the end of the stream: %s(%s)", c, events_string(r.events[static_cast<size_t>(c)], 16).c_str(), where_it_stopped(r, c).c_str());
            }
        EXPECT(losses >= 1, "no channel was given up although the device delivered nothing");
    });
    // 5. a dwell fails: negative acquisition ("events" 2, acq.cc:344-351's message), the FSM asks for a satellite, the channel is acquired at the next attempt
    fault_case("a dwell fails", FAULT_ACQ_DWELL, 2, 2, -1, x, n_channels, {}, [&](const Churn_Result& r, long hits) {
        EXPECT(hits == 2, "the injected dwell failure was hit %ld times", hits);
        size_t failed = 0;
        for (int c = 0; c < n_channels; c++)
            {
                failed += static_cast<size_t>(std::count_if(r.events[static_cast<size_t>(c)].begin(), r.events[static_cast<size_t>(c)].end(), [](const Event& e) { return e.what == 0; }));
                EXPECT(tracked_to_the_end(r, c), "channel %d is not tracking at the end of the stream: %s(%s)", c, events_string(r.events[static_cast<size_t>(c)], 16).c_str(), where_it_stopped(r, c).c_str());
            }
        EXPECT(failed >= 2, "%zu failed acquisitions for two failed dwells", failed);  // (a first dwell on a weak satellite fails now and then without any help)
    });
    // 6. gsh_trk_start fails for one channel at its first hand-over: "events" 3 from the pull-in call, re-acquired, tracked
    fault_case("a tracking start fails", FAULT_TRK_START, 0, 1, 1, x, n_channels, {}, [&](const Churn_Result& r, long hits) {
        EXPECT(hits == 1, "the injected start failure was hit %ld times", hits);
        long losses = 0;
        for (int c = 0; c < n_channels; c++)
            {
                losses += losses_of(r, c);
                EXPECT(tracked_to_the_end(r, c), "channel %d is not tracking at the end of the stream: %s(%s)", c, events_string(r.events[static_cast<size_t>(c)], 16).c_str(), where_it_stopped(r, c).c_str());
            }
        EXPECT(losses == 1, "%ld channels dropped for one failed start", losses);
    });
    // 7. launched mode (hip_live = false): a launch fails once -- the channels of that launch drop and come back
    fault_case("a launch fails (launched mode)", FAULT_RUN, 40, 1, -1, x, n_channels, {{"Tracking_1C.hip_live", "false"}}, [&](const Churn_Result& r, long hits) {
        EXPECT(hits == 1, "the injected launch failure was hit %ld times", hits);
        long losses = 0;
        for (int c = 0; c < n_channels; c++)
            {
                losses += losses_of(r, c);
                EXPECT(tracked_to_the_end(r, c), "channel %d is not tracking at the end of the stream: %s(%s)", c, events_string(r.events[static_cast<size_t>(c)], 16).c_str(), where_it_stopped(r, c).c_str());
            }
        EXPECT(losses >= 1, "no channel dropped for a failed launch");
    });
}
// ---- <role>.hip_devices of the acquisition role: channel c searches on GPU c mod G; run with FAKE_GSH_DEVICES=3 on the CPU
void test_acquisition_devices()
{
    if (gsh_device_count() < 3)
        {
            std::printf("acquisition devices: needs three devices (FAKE_GSH_DEVICES=3 with the fake engine)\n");
            return;
        }
    Props p = receiver_props(4000000, {{"Acquisition_1C.hip_devices", "0,1,2"}, {"Tracking_1C.hip_devices", "0,1,2"}});
    p.erase("Acquisition_1C.hip_device");
    p.erase("Tracking_1C.hip_device");
    auto cfg = std::make_shared<InMemoryConfiguration>();
    for (const auto& kv : p) cfg->set_property(kv.first, kv.second);
    std::vector<int> acq_dev, trk_dev;
    std::vector<std::shared_ptr<GpsL1CaPcpsAcquisitionHip>> acq;
    std::vector<std::shared_ptr<GpsL1CaDllPllTrackingHip>> trk;
    for (int c = 0; c < 7; c++)
        {
            acq.push_back(std::make_shared<GpsL1CaPcpsAcquisitionHip>(cfg.get(), "Acquisition_1C", 1, 0));
            trk.push_back(std::make_shared<GpsL1CaDllPllTrackingHip>(cfg.get(), "Tracking_1C", 1, 1));
            EXPECT(acq.back()->item_size() != 0 && trk.back()->item_size() != 0, "channel %d: unusable block", c);
            if (acq.back()->item_size() == 0 || trk.back()->item_size() == 0) return;
            acq_dev.push_back(acq.back()->device());
            trk_dev.push_back(trk.back()->block()->runtime()->device());
        }
    const std::vector<int> want{0, 1, 2, 0, 1, 2, 0};
    EXPECT(acq_dev == want, "acquisition blocks dealt to devices %d %d %d %d %d %d %d", acq_dev[0], acq_dev[1], acq_dev[2], acq_dev[3], acq_dev[4], acq_dev[5], acq_dev[6]);
    EXPECT(trk_dev == want, "tracking blocks dealt to devices %d %d %d %d %d %d %d (a channel's two blocks belong on one GPU)", trk_dev[0], trk_dev[1], trk_dev[2], trk_dev[3], trk_dev[4],
        trk_dev[5], trk_dev[6]);
    cfg->set_property("Acquisition_1C.hip_device", "2");
    GpsL1CaPcpsAcquisitionHip pinned(cfg.get(), "Acquisition_1C", 1, 0);
    EXPECT(pinned.item_size() != 0 && pinned.device() == 2, "hip_device must pin the block (device %d)", pinned.device());
    std::printf("acquisition devices: seven channels' acquisition and tracking blocks dealt 0 1 2 0 1 2 0, hip_device pins\n");
}
}  // namespace

int main(int argc, char** argv)
{
    const std::string mode = argc > 1 ? argv[1] : "all";
    if (gsh_device_count() < 1)
        {
            std::printf("no HIP device\n");
            return 2;
        }
    const bool fake = fake_gsh_inject_fault != nullptr;
    if (mode == "life" || mode == "all") test_channel_life();
    if (mode == "churn" || mode == "all")
        {
            // test_channel churn [channels churners seconds reference(0/1)]: the GPU run takes BASELINE config 2's 32 channels with 8 churners; the CPU suite a smaller receiver
            const int n = argc > 2 ? std::atoi(argv[2]) : (fake ? 12 : 32);
            const int k = argc > 3 ? std::atoi(argv[3]) : (fake ? 4 : 8);
            const double s = argc > 4 ? std::atof(argv[4]) : (fake ? 1.6 : 2.4);
            const bool with_ref = argc > 5 ? std::atoi(argv[5]) != 0 : true;
            test_channel_churn(n, k, s, with_ref);
        }
    if (mode == "faults" || (mode == "all" && fake)) test_faults();
    if (mode == "devices") test_acquisition_devices();
    std::printf(fails == 0 ? "CHANNEL OK\n" : "%d failure(s)\n", fails);
    return fails == 0 ? 0 : 1;
}
