/*!
 * \file dll_pll_tracking_hip.cc
 * \brief TrackingInterface adapters over the MI355X device-closed DLL/PLL loop; see the header.
 */
#include "dll_pll_tracking_hip.h"
#include "Beidou_B1I.h"
#include "Beidou_B3I.h"
#include "GLONASS_L1_L2_CA.h"
#include "GPS_L1_CA.h"
#include "GPS_L2C.h"
#include "GPS_L5.h"
#include "Galileo_E1.h"
#include "Galileo_E5a.h"
#include "Galileo_E5b.h"
#include "Galileo_E6.h"
#include "qzss.h"
#include "configuration_interface.h"
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdlib>
#include <iostream>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#if USE_GLOG_AND_GFLAGS
#include <glog/logging.h>
#else
#include <absl/log/log.h>
#endif

namespace
{
// Which blocks share one Hip_Tracking_Runtime and, through it, one device sample ring -- the stream then crosses PCIe once for all of them and one resident
// loop kernel (or one launch) advances all of them (hip_tracking_runtime.h):
//   <role>.hip_shared_ring  absent / -2  every block of this role (= signal class: "Tracking_1C", ...) and RF chain (Channels_<signal>.RF_channel_ID) on this device:
//                                        the reference's channels of one signal read one buffer (gnss_flowgraph.cc:1227-1231), so this is the default.  Receivers
//                                        that route single channels elsewhere (Channel<i>.RF_channel_ID) fall back to -1 (create_tracking_block)
//                           >= 0         every block on this device that names the same id, whatever its role (signals of one RF stream: L1 C/A + E1)
//                           -1           a runtime (and ring) of the block's own.  Needed when blocks of one role are fed from DIFFERENT streams (receivers
//                                        with several RF chains per signal): the ring de-duplicates pushes by absolute sample index, which only holds for
//                                        blocks that see the same stream.
// The ring is sized by time, not by the first joiner's code period, so that signals with different periods on the same RF stream (L1 C/A 1 ms, E1 4 ms,
// L2C 20 ms) fit: 256 ms resident, windows of up to 40 ms contiguous; never less than 64 / 2 of the joiner's periods.
// <role>.hip_devices = 0,1,2,...: the stream is resident in several GPUs of the node and the blocks of the role are dealt to them in turn (SURVEY.md 8e:
// channel c -> GPU c mod G; <role>.hip_device, when given, pins a block instead).  The ring then is ONE Hip_Sample_Ring over the engine's stream group: whichever
// block is offered new samples first pushes them once -- into the ingest GPU, devices[0], over PCIe -- and RCCL replicates them over xGMI into every device's
// ring; each device has its own Hip_Tracking_Runtime (its handles and residencies) on the shared ring.
struct SharedStream
{
    std::weak_ptr<Hip_Sample_Ring> ring;
    unsigned dealt{0};  // blocks handed a device so far (round robin)
};

std::vector<int> parse_devices(const std::string& list)
{
    std::vector<int> out;
    size_t at = 0;
    while (at < list.size())
        {
            const size_t comma = list.find(',', at);
            const std::string tok = list.substr(at, comma == std::string::npos ? std::string::npos : comma - at);
            if (!tok.empty()) out.push_back(std::atoi(tok.c_str()));
            if (comma == std::string::npos) break;
            at = comma + 1;
        }
    return out;
}

std::shared_ptr<Hip_Tracking_Runtime> runtime_for(int device, bool device_given, const std::vector<int>& devices, int id, const std::string& role, const Dll_Pll_Conf& p,
    int periods_per_launch, bool register_input, int channels_per_launch, bool live, int* device_out)
{
    static std::mutex mu;
    static std::map<std::pair<int, std::string>, std::weak_ptr<Hip_Tracking_Runtime>> runtimes;
    static std::map<std::string, SharedStream> streams;
    const uint64_t vlen = std::max<uint32_t>(p.vector_length, 1U);
    const bool grouped = devices.size() > 1 && id != -1;
    auto make_ring = [&](uint64_t capacity, uint64_t window) -> std::shared_ptr<Hip_Sample_Ring> {
        auto ring = grouped ? std::make_shared<Hip_Sample_Ring>(devices, capacity, static_cast<uint32_t>(window))
                            : std::make_shared<Hip_Sample_Ring>(device, capacity, static_cast<uint32_t>(window));
        if (!ring->ok())
            {
                LOG(ERROR) << "hip sample ring (" << capacity << " samples): " << ring->last_error();
                return nullptr;
            }
        ring->set_auto_register(register_input);
        return ring;
    };
    *device_out = device;
    if (id == -1)
        {
            auto ring = make_ring(std::max<uint64_t>(16, 4ULL * (static_cast<uint64_t>(periods_per_launch) + 2)) * vlen, 2 * vlen);
            return ring ? std::make_shared<Hip_Tracking_Runtime>(device, std::move(ring), periods_per_launch, 1, live) : nullptr;
        }
    std::lock_guard<std::mutex> lk(mu);
    const std::string key = id >= 0 ? "id:" + std::to_string(id) : "role:" + role;
    std::shared_ptr<Hip_Sample_Ring> ring;
    if (grouped)
        {
            SharedStream& st = streams[key];
            ring = st.ring.lock();
            if (!ring)
                {
                    const auto fs = static_cast<uint64_t>(std::max(p.fs_in, 1.0));
                    ring = make_ring(std::max<uint64_t>(64 * vlen, fs * 256 / 1000), std::max<uint64_t>(2 * vlen, fs * 40 / 1000));
                    if (!ring) return nullptr;
                    st.ring = ring;
                    st.dealt = 0;
                }
            if (!device_given) device = devices[st.dealt++ % devices.size()];
            if (ring->handle_for(device) == nullptr)
                {
                    LOG(ERROR) << role << ": hip_device " << device << " is not one of hip_devices";
                    return nullptr;
                }
            *device_out = device;
        }
    auto& slot = runtimes[{device, key}];
    auto rt = slot.lock();
    if (!rt)
        {
            if (!ring)
                {
                    const auto fs = static_cast<uint64_t>(std::max(p.fs_in, 1.0));
                    ring = make_ring(std::max<uint64_t>(64 * vlen, fs * 256 / 1000), std::max<uint64_t>(2 * vlen, fs * 40 / 1000));
                    if (!ring) return nullptr;
                }
            rt = std::make_shared<Hip_Tracking_Runtime>(device, std::move(ring), periods_per_launch, channels_per_launch, live);
            slot = rt;
        }
    return rt;
}

void set_signal(Dll_Pll_Conf& p, char system, char s0, char s1)
{
    p.system = system;
    const std::array<char, 3> sig{s0, s1, '\0'};
    std::copy_n(sig.data(), 3, p.signal);
}

void warn_narrow(const Dll_Pll_Conf& p, const char* name)
{
    if ((p.extend_correlation_symbols > 1) && (p.pll_bw_narrow_hz > p.pll_bw_hz || p.dll_bw_narrow_hz > p.dll_bw_hz))
        std::cout << "WARNING: " << name << ". PLL or DLL narrow tracking bandwidth is higher than wide tracking one\n";
}
}  // namespace


DllPllTrackingHip::DllPllTrackingHip(const ConfigurationInterface* configuration, std::string role, unsigned int in_streams, unsigned int out_streams)
    : role_(std::move(role)), item_size_(sizeof(gr_complex))
{
    trk_params_.SetFromConfiguration(configuration, role_);  // base_dll_pll_tracking.cc:33
    if (in_streams > 1) LOG(ERROR) << "Only one input stream is supported.";
    if (out_streams > 1) LOG(ERROR) << "Only one output stream is supported.";
    DLOG(INFO) << "role " << role_;
}


void DllPllTrackingHip::create_tracking_block(const ConfigurationInterface* configuration)
{
    const int device_key = configuration->property(role_ + ".hip_device", -1);
    const std::vector<int> devices = parse_devices(configuration->property(role_ + ".hip_devices", std::string("")));
    int device = device_key >= 0 ? device_key : (devices.empty() ? 0 : devices[0]);
    // code periods one general_work call may consume and emit when its input covers them: 1 is the reference's cadence (trk.cc:1898-2001: one period, then
    // back to the scheduler); larger values save scheduler round trips when the receiver post-processes a file faster than real time (the runtime's throughput
    // at 1 / 20: bench.py -> dropin).  The device works ahead of the blocks either way -- this key only sets how much a block takes per call.
    const int periods = configuration->property(role_ + ".hip_periods_per_call", 1);
    int ring_id = configuration->property(role_ + ".hip_shared_ring", -2);
    std::string stream_tag;
    if (ring_id == -2)
        {
            // "Every block of the role shares one ring" holds only for blocks that are fed the SAME stream: the ring de-duplicates pushes by absolute sample index.
            // The flowgraph connects channel i to sig_conditioner_[RF_channel_ID], taken from Channels_<signal>.RF_channel_ID and overridden per channel by
            // Channel<i>.RF_channel_ID (gnss_flowgraph.cc:1107-1108, 1227-1231).  The signal's chain is part of the sharing key; a receiver that routes single channels
            // to other chains is not shared by default at all (the block does not know its channel number here: the factory passes the role only) -- such receivers
            // name their groups themselves, <role>.hip_shared_ring = id per chain.
            const std::string sig(trk_params_.signal);
            const int rf_chain = configuration->property("Channels_" + sig + ".RF_channel_ID", 0);
            stream_tag = ":rf" + std::to_string(rf_chain);
            int n_channels = 0;
            for (const char* s : {"1C", "2S", "L5", "1B", "5X", "7X", "E6", "1G", "2G", "B1", "B3", "J1", "J5"})
                n_channels += std::max(0, configuration->property(std::string("Channels_") + s + ".count", 0));
            n_channels = std::min(n_channels, 512);
            for (int i = 0; i < n_channels; i++)
                {
                    const int own = configuration->property("Channel" + std::to_string(i) + ".RF_channel_ID", rf_chain);
                    if (own != rf_chain)
                        {
                            LOG(WARNING) << role_ << ": Channel" << i << ".RF_channel_ID = " << own << " routes a channel to another RF chain than Channels_" << sig
                                         << ".RF_channel_ID = " << rf_chain << ": the blocks of this role get a sample ring each (set " << role_
                                         << ".hip_shared_ring = <id per chain> to share one per chain)";
                            ring_id = -1;
                            break;
                        }
                }
        }
    // the loop kernel stays resident and follows the ring (hip_tracking_runtime.h, "LIVE mode"); false: one launch per batch of periods, as in round 3
    const bool live = configuration->property(role_ + ".hip_live", true);
    const int per_launch = configuration->property(role_ + ".hip_periods_per_launch", std::max(16, periods));
    if (trk_params_.item_type != "gr_complex")  // as the reference adapters: item_size 0 tells the factory the block is unusable
        {
            item_size_ = 0;
            tracking_sptr_ = nullptr;
            LOG(WARNING) << trk_params_.item_type << " unknown tracking item type.";
            return;
        }
    for (const int d : devices)
        if (d < 0 || gsh_device_count() <= d) device = d;  // (reported by the check below)
    if (device < 0 || gsh_device_count() <= device)
        {
            item_size_ = 0;
            tracking_sptr_ = nullptr;
            LOG(ERROR) << role_ << ": HIP device " << device << " not present (the MI355X tracking block has no CPU fallback)";
            return;
        }
    // the scheduler re-uses one input buffer for the whole run: page-locking it (lazily, the part each call shows) turns every push into a true DMA
    const bool register_input = configuration->property(role_ + ".hip_register_input_buffer", true);
    // most channels of one loop configuration that share a launch (one work-group each; a further handle is opened beyond that).  Every launch brings the
    // records of all the handle's slots back, so the default stays at what BASELINE's configurations put on one GPU (32 - 50 channels per stream)
    const int per_handle = configuration->property(role_ + ".hip_channels_per_launch", 64);
    auto runtime = runtime_for(device, device_key >= 0, devices, ring_id, role_ + stream_tag, trk_params_, per_launch, register_input, per_handle, live, &device);
    if (!runtime)
        {
            item_size_ = 0;
            tracking_sptr_ = nullptr;
            return;
        }
    // live mode's watchdog: a resident window without a record for this long = the device is not delivering; the channel is given up ("events" 3)
    runtime->set_record_timeout_ms(configuration->property(role_ + ".hip_record_timeout_ms", 1000));
    // launched mode: work-groups that share every window of a channel (hip_tracking_runtime.h); the blocks of a role share the runtime, the first one's value stands
    if (configuration->property(role_ + ".hip_work_groups_per_channel", 1) != 1)
        runtime->set_work_groups_per_channel(configuration->property(role_ + ".hip_work_groups_per_channel", 1));
    tracking_sptr_ = dll_pll_veml_make_tracking_hip(trk_params_, periods, std::move(runtime));
    if (!tracking_sptr_->usable())
        {
            LOG(WARNING) << role_ << ": " << tracking_sptr_->last_error();
            item_size_ = 0;
            tracking_sptr_ = nullptr;
            return;
        }
    DLOG(INFO) << "tracking(" << tracking_sptr_->unique_id() << ")";
}


void DllPllTrackingHip::connect(gr::top_block_sptr top_block)
{
    if (top_block)
        { /* no connection needed */
        }
}


void DllPllTrackingHip::disconnect(gr::top_block_sptr top_block)
{
    if (top_block)
        { /* no disconnection needed */
        }
}


gr::basic_block_sptr DllPllTrackingHip::get_left_block() { return tracking_sptr_; }
gr::basic_block_sptr DllPllTrackingHip::get_right_block() { return tracking_sptr_; }
// (an unusable adapter -- item_size() == 0: no GPU, unsupported item type -- has no block: the factory drops it, gnss_block_factory.cc:1048-1052, but a Channel that is
//  built all the same calls these from its constructor, channel.cc:53-61; they must not crash)
void DllPllTrackingHip::set_channel(unsigned int channel)
{
    if (tracking_sptr_) tracking_sptr_->set_channel(channel);
}
void DllPllTrackingHip::set_gnss_synchro(Gnss_Synchro* p_gnss_synchro)
{
    if (tracking_sptr_) tracking_sptr_->set_gnss_synchro(p_gnss_synchro);
}
void DllPllTrackingHip::start_tracking()
{
    if (tracking_sptr_) tracking_sptr_->start_tracking();
}
void DllPllTrackingHip::stop_tracking()
{
    if (tracking_sptr_) tracking_sptr_->stop_tracking();
}


GpsL1CaDllPllTrackingHip::GpsL1CaDllPllTrackingHip(const ConfigurationInterface* configuration, const std::string& role, unsigned int in_streams,
    unsigned int out_streams)
    : DllPllTrackingHip(configuration, role, in_streams, out_streams)
{
    // gps_l1_ca_dll_pll_tracking.cc:50-95
    Dll_Pll_Conf& p = config_params();
    set_signal(p, 'G', '1', 'C');
    p.vector_length = static_cast<uint32_t>(static_cast<int>(std::round(p.fs_in / (GPS_L1_CA_CODE_RATE_CPS / GPS_L1_CA_CODE_LENGTH_CHIPS))));
    if (p.extend_correlation_symbols < 1)
        {
            p.extend_correlation_symbols = 1;
            std::cout << "WARNING: GPS L1 C/A: extend_correlation_symbols must be > 0. Coherent integration set to 1 ms.\n";
        }
    else if (p.extend_correlation_symbols > 20)
        {
            p.extend_correlation_symbols = 20;
            std::cout << "WARNING: GPS L1 C/A: extend_correlation_symbols limited to 20 (20 ms).\n";
        }
    p.track_pilot = configuration->property(role + ".track_pilot", false);
    if (p.track_pilot)
        {
            p.track_pilot = false;
            std::cout << "WARNING: GPS L1 C/A does not have pilot signal. Data tracking enabled instead.\n";
        }
    warn_narrow(p, "GPS L1 C/A");
    create_tracking_block(configuration);
}


GalileoE1DllPllVemlTrackingHip::GalileoE1DllPllVemlTrackingHip(const ConfigurationInterface* configuration, const std::string& role,
    unsigned int in_streams, unsigned int out_streams)
    : DllPllTrackingHip(configuration, role, in_streams, out_streams)
{
    // galileo_e1_dll_pll_veml_tracking.cc:48-72
    Dll_Pll_Conf& p = config_params();
    p.vector_length = static_cast<uint32_t>(static_cast<int>(std::round(p.fs_in / (GALILEO_E1_CODE_CHIP_RATE_CPS / GALILEO_E1_B_CODE_LENGTH_CHIPS))));
    set_signal(p, 'E', '1', 'B');
    if (p.extend_correlation_symbols < 1)
        {
            p.extend_correlation_symbols = 1;
            std::cout << "WARNING: Galileo E1. extend_correlation_symbols must be bigger than 0. Coherent integration has been set to 1 symbol (4 ms)\n";
        }
    else if (!p.track_pilot && p.extend_correlation_symbols > 1)
        {
            p.extend_correlation_symbols = 1;
            std::cout << "WARNING: Galileo E1. Extended coherent integration is not allowed when tracking the data component. Coherent integration has been set to 4 ms (1 symbol)\n";
        }
    warn_narrow(p, "Galileo E1");
    create_tracking_block(configuration);
}


GpsL5DllPllTrackingHip::GpsL5DllPllTrackingHip(const ConfigurationInterface* configuration, const std::string& role, unsigned int in_streams,
    unsigned int out_streams)
    : DllPllTrackingHip(configuration, role, in_streams, out_streams)
{
    // gps_l5_dll_pll_tracking.cc:48-70
    Dll_Pll_Conf& p = config_params();
    p.vector_length = static_cast<uint32_t>(static_cast<int>(
        std::round(static_cast<double>(p.fs_in) / (static_cast<double>(GPS_L5I_CODE_RATE_CPS) / static_cast<double>(GPS_L5I_CODE_LENGTH_CHIPS)))));
    if (p.extend_correlation_symbols < 1)
        {
            p.extend_correlation_symbols = 1;
            std::cout << "WARNING: GPS L5. extend_correlation_symbols must be bigger than 0. Coherent integration has been set to 1 symbol (1 ms)\n";
        }
    else if (!p.track_pilot && p.extend_correlation_symbols > GPS_L5I_NH_CODE_LENGTH)
        {
            p.extend_correlation_symbols = GPS_L5I_NH_CODE_LENGTH;
            std::cout << "WARNING: GPS L5. extend_correlation_symbols must be lower than 11 when tracking the data component. Coherent integration has been set to 10 symbols (10 ms)\n";
        }
    warn_narrow(p, "GPS L5");
    set_signal(p, 'G', 'L', '5');
    create_tracking_block(configuration);
}


GalileoE5aDllPllTrackingHip::GalileoE5aDllPllTrackingHip(const ConfigurationInterface* configuration, const std::string& role, unsigned int in_streams,
    unsigned int out_streams)
    : DllPllTrackingHip(configuration, role, in_streams, out_streams)
{
    // galileo_e5a_dll_pll_tracking.cc:48-70
    Dll_Pll_Conf& p = config_params();
    p.vector_length = static_cast<uint32_t>(static_cast<int>(std::round(p.fs_in / (GALILEO_E5A_CODE_CHIP_RATE_CPS / GALILEO_E5A_CODE_LENGTH_CHIPS))));
    set_signal(p, 'E', '5', 'X');
    if (p.extend_correlation_symbols < 1)
        {
            p.extend_correlation_symbols = 1;
            std::cout << "WARNING: Galileo E5a. extend_correlation_symbols must be bigger than 0. Coherent integration has been set to 1 symbol (1 ms)\n";
        }
    else if (!p.track_pilot && p.extend_correlation_symbols > GALILEO_E5A_I_SECONDARY_CODE_LENGTH)
        {
            p.extend_correlation_symbols = GALILEO_E5A_I_SECONDARY_CODE_LENGTH;
            std::cout << "WARNING: Galileo E5a. extend_correlation_symbols must be lower than 21 when tracking the data component. Coherent integration has been set to 20 symbols (20 ms)\n";
        }
    warn_narrow(p, "Galileo E5a");
    create_tracking_block(configuration);
}


namespace
{
// the checks most adapters share: extend_correlation_symbols in [1, limit], no pilot component
void clamp_extend(Dll_Pll_Conf& p, int limit, const char* name, const char* unit)
{
    if (p.extend_correlation_symbols < 1)
        {
            p.extend_correlation_symbols = 1;
            std::cout << "WARNING: " << name << ": extend_correlation_symbols must be > 0. Coherent integration set to 1 " << unit << ".\n";
        }
    else if (p.extend_correlation_symbols > limit)
        {
            p.extend_correlation_symbols = limit;
            std::cout << "WARNING: " << name << ": extend_correlation_symbols limited to " << limit << ".\n";
        }
}

void no_pilot(Dll_Pll_Conf& p, const ConfigurationInterface* configuration, const std::string& role, const char* name)
{
    p.track_pilot = configuration->property(role + ".track_pilot", false);
    if (p.track_pilot)
        {
            p.track_pilot = false;
            std::cout << "WARNING: " << name << " does not have pilot signal. Data tracking enabled instead.\n";
        }
}

uint32_t period_samples(double fs_in, double chip_rate, double code_length_chips) { return static_cast<uint32_t>(static_cast<int>(std::round(fs_in / (chip_rate / code_length_chips)))); }
}  // namespace


GpsL2MDllPllTrackingHip::GpsL2MDllPllTrackingHip(const ConfigurationInterface* configuration, const std::string& role, unsigned int in_streams, unsigned int out_streams)
    : DllPllTrackingHip(configuration, role, in_streams, out_streams)
{
    // gps_l2_m_dll_pll_tracking.cc: one 20 ms code period per symbol, no extended integration, no pilot
    Dll_Pll_Conf& p = config_params();
    p.vector_length = period_samples(static_cast<double>(p.fs_in), GPS_L2_M_CODE_RATE_CPS, GPS_L2_M_CODE_LENGTH_CHIPS);
    if (p.extend_correlation_symbols != 1)
        {
            p.extend_correlation_symbols = 1;
            std::cout << "WARNING: Extended coherent integration is not allowed in GPS L2. Coherent integration has been set to 20 ms (1 symbol)\n";
        }
    no_pilot(p, configuration, role, "GPS L2");
    set_signal(p, 'G', '2', 'S');
    create_tracking_block(configuration);
}


GalileoE5bDllPllTrackingHip::GalileoE5bDllPllTrackingHip(const ConfigurationInterface* configuration, const std::string& role, unsigned int in_streams, unsigned int out_streams)
    : DllPllTrackingHip(configuration, role, in_streams, out_streams)
{
    Dll_Pll_Conf& p = config_params();
    p.vector_length = period_samples(static_cast<double>(p.fs_in), GALILEO_E5B_CODE_CHIP_RATE_CPS, GALILEO_E5B_CODE_LENGTH_CHIPS);
    if (p.extend_correlation_symbols < 1)
        {
            p.extend_correlation_symbols = 1;
            std::cout << "WARNING: Galileo E5b. extend_correlation_symbols must be bigger than 0. Coherent integration has been set to 1 symbol (1 ms)\n";
        }
    else if (!p.track_pilot && p.extend_correlation_symbols > GALILEO_E5B_I_SECONDARY_CODE_LENGTH)
        {
            p.extend_correlation_symbols = GALILEO_E5B_I_SECONDARY_CODE_LENGTH;
            std::cout << "WARNING: Galileo E5b. extend_correlation_symbols must be lower than 5 when tracking the data component. Coherent integration has been set to 4 symbols (4 ms)\n";
        }
    warn_narrow(p, "Galileo E5b");
    set_signal(p, 'E', '7', 'X');
    create_tracking_block(configuration);
}


GalileoE6DllPllTrackingHip::GalileoE6DllPllTrackingHip(const ConfigurationInterface* configuration, const std::string& role, unsigned int in_streams, unsigned int out_streams)
    : DllPllTrackingHip(configuration, role, in_streams, out_streams)
{
    // galileo_e6_dll_pll_tracking.cc.  NOTE: the reference adapter of the version at hand writes the signal tag {'5', 'X'} (its :62-64), so the block it
    // creates takes the E5a branch of its constructor with E6's window length.  <role>.hip_signal_e6 = true (default) hands the block "E6" -- the
    // branch the reference block itself has for this signal (trk.cc:349-372, :904-921); false reproduces the adapter as written.
    Dll_Pll_Conf& p = config_params();
    p.vector_length = period_samples(static_cast<double>(p.fs_in), GALILEO_E6_B_CODE_CHIP_RATE_CPS, GALILEO_E6_B_CODE_LENGTH_CHIPS);
    if (configuration->property(role + ".hip_signal_e6", true))
        set_signal(p, 'E', 'E', '6');
    else
        set_signal(p, 'E', '5', 'X');
    if (p.extend_correlation_symbols < 1)
        {
            p.extend_correlation_symbols = 1;
            std::cout << "WARNING: Galileo E6. extend_correlation_symbols must be bigger than 0. Coherent integration has been set to 1 symbol (1 ms)\n";
        }
    else if (!p.track_pilot && p.extend_correlation_symbols > 1)
        {
            p.extend_correlation_symbols = 1;
            std::cout << "WARNING: Galileo E6. Extended coherent integration is not allowed when tracking the data component. Coherent integration has been set to 1 ms (1 symbol)\n";
        }
    warn_narrow(p, "Galileo E6");
    create_tracking_block(configuration);
}


BeidouB1iDllPllTrackingHip::BeidouB1iDllPllTrackingHip(const ConfigurationInterface* configuration, const std::string& role, unsigned int in_streams, unsigned int out_streams)
    : DllPllTrackingHip(configuration, role, in_streams, out_streams)
{
    Dll_Pll_Conf& p = config_params();
    p.vector_length = period_samples(static_cast<double>(p.fs_in), BEIDOU_B1I_CODE_RATE_CPS, BEIDOU_B1I_CODE_LENGTH_CHIPS);
    set_signal(p, 'C', 'B', '1');
    clamp_extend(p, 20, "BEIDOU B1I", "symbol (1 ms)");
    no_pilot(p, configuration, role, "BEIDOU B1I");
    warn_narrow(p, "BEIDOU B1I");
    create_tracking_block(configuration);
}


BeidouB3iDllPllTrackingHip::BeidouB3iDllPllTrackingHip(const ConfigurationInterface* configuration, const std::string& role, unsigned int in_streams, unsigned int out_streams)
    : DllPllTrackingHip(configuration, role, in_streams, out_streams)
{
    Dll_Pll_Conf& p = config_params();
    p.vector_length = period_samples(static_cast<double>(p.fs_in), BEIDOU_B3I_CODE_RATE_CPS, BEIDOU_B3I_CODE_LENGTH_CHIPS);
    set_signal(p, 'C', 'B', '3');
    p.track_pilot = configuration->property(role + ".track_pilot", false);  // beidou_b3i_dll_pll_tracking.cc reads the key; the block ignores it (trk.cc:442)
    clamp_extend(p, 20, "BEIDOU B3I", "symbol (1 ms)");
    create_tracking_block(configuration);
}


GlonassL1CaDllPllTrackingHip::GlonassL1CaDllPllTrackingHip(const ConfigurationInterface* configuration, const std::string& role, unsigned int in_streams, unsigned int out_streams)
    : DllPllTrackingHip(configuration, role, in_streams, out_streams)
{
    Dll_Pll_Conf& p = config_params();
    set_signal(p, 'R', '1', 'G');
    p.vector_length = period_samples(static_cast<double>(p.fs_in), GLONASS_L1_CA_CODE_RATE_CPS, GLONASS_L1_CA_CODE_LENGTH_CHIPS);
    clamp_extend(p, 10, "Glonass L1", "ms");
    no_pilot(p, configuration, role, "Glonass L1");
    warn_narrow(p, "Glonass L1");
    create_tracking_block(configuration);
}


GlonassL2CaDllPllTrackingHip::GlonassL2CaDllPllTrackingHip(const ConfigurationInterface* configuration, const std::string& role, unsigned int in_streams, unsigned int out_streams)
    : DllPllTrackingHip(configuration, role, in_streams, out_streams)
{
    Dll_Pll_Conf& p = config_params();
    set_signal(p, 'R', '2', 'G');
    p.vector_length = period_samples(static_cast<double>(p.fs_in), GLONASS_L2_CA_CODE_RATE_CPS, GLONASS_L2_CA_CODE_LENGTH_CHIPS);
    clamp_extend(p, 10, "Glonass L2", "ms");
    no_pilot(p, configuration, role, "Glonass L2");
    warn_narrow(p, "Glonass L2");
    create_tracking_block(configuration);
}


QzssL1DllPllTrackingHip::QzssL1DllPllTrackingHip(const ConfigurationInterface* configuration, const std::string& role, unsigned int in_streams, unsigned int out_streams)
    : DllPllTrackingHip(configuration, role, in_streams, out_streams)
{
    Dll_Pll_Conf& p = config_params();
    set_signal(p, 'J', 'J', '1');
    p.vector_length = period_samples(static_cast<double>(p.fs_in), QZSS_L1_CHIP_RATE, QZSS_L1_CODE_LENGTH);
    clamp_extend(p, 20, "QZSS L1 C/A", "ms");
    no_pilot(p, configuration, role, "QZSS L1 C/A");
    warn_narrow(p, "QZSS L1 C/A");
    create_tracking_block(configuration);
}


QzssL5DllPllTrackingHip::QzssL5DllPllTrackingHip(const ConfigurationInterface* configuration, const std::string& role, unsigned int in_streams, unsigned int out_streams)
    : DllPllTrackingHip(configuration, role, in_streams, out_streams)
{
    Dll_Pll_Conf& p = config_params();
    p.vector_length = period_samples(static_cast<double>(p.fs_in), static_cast<double>(QZSS_L5_CHIP_RATE), static_cast<double>(QZSS_L5_CODE_LENGTH));
    if (p.extend_correlation_symbols < 1)
        {
            p.extend_correlation_symbols = 1;
            std::cout << "WARNING: QZSS L5. extend_correlation_symbols must be bigger than 0. Coherent integration has been set to 1 symbol (1 ms)\n";
        }
    else if (!p.track_pilot && p.extend_correlation_symbols > QZSS_L5I_NH_CODE_LENGTH)
        {
            p.extend_correlation_symbols = QZSS_L5I_NH_CODE_LENGTH;
            std::cout << "WARNING: QZSS L5. extend_correlation_symbols must be lower than 11 when tracking the data component. Coherent integration has been set to 10 symbols (10 ms)\n";
        }
    warn_narrow(p, "QZSS L5");
    set_signal(p, 'J', 'J', '5');
    create_tracking_block(configuration);
}
