/*!
 * \file base_pcps_acquisition_hip.cc
 * \brief See the header.  BUILT ONLY INSIDE A gnss-sdr TREE.
 */
#include "base_pcps_acquisition_hip.h"
#include "configuration_interface.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

namespace
{
// Channels configured with the same <role>.hip_shared_acquisition id (>= 0) on the same device share one Hip_Acquisition_Runtime: blocks that
// search at the same time join one dwell batch (forward transforms once for all of them).  The first block's dwell geometry defines the
// runtime; a block whose geometry differs (another signal, another Doppler grid) keeps to its own handle.
std::shared_ptr<Hip_Acquisition_Runtime> acquisition_runtime_for(int device, int id, const Hip_Acq_Conf& conf, int max_channels, int max_wait_us)
{
    static std::mutex mu;
    static std::map<std::pair<int, int>, std::weak_ptr<Hip_Acquisition_Runtime>> runtimes;
    if (id < 0) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    auto& slot = runtimes[{device, id}];
    auto rt = slot.lock();
    if (!rt)
        {
            Hip_Pcps_Acquisition_Core probe(conf, device);  // derives the engine-side geometry exactly as every block's core does
            if (!probe.ok()) return nullptr;
            rt = std::make_shared<Hip_Acquisition_Runtime>(device, probe.engine_conf(), max_channels, std::chrono::microseconds(max_wait_us));
            if (!rt->ok()) return nullptr;
            slot = rt;
        }
    return rt;
}

// <role>.hip_devices = 0,1,2,...: the acquisition blocks of a role are dealt over these GPUs in the order they are built -- the factory builds channel 0, 1, 2, ... in turn
// (gnss_block_factory.cc:1040-1054), so channel c searches on GPU c mod G (SURVEY 8e "PRN p -> GPU p mod G": a channel searches one PRN at a time), next to its tracking block,
// which <role>.hip_devices of the tracking role deals the same way (dll_pll_tracking_hip.cc).  <role>.hip_device, when given, pins a block.  An acquisition block reads its
// own input buffer (acq.cc:790-815), not the device sample ring, so nothing has to be replicated for it.
int acquisition_device_for(const ConfigurationInterface* configuration, const std::string& role)
{
    const int pinned = configuration->property(role + ".hip_device", -1);
    if (pinned >= 0) return pinned;
    const std::string list = configuration->property(role + ".hip_devices", std::string(""));
    std::vector<int> devices;
    size_t at = 0;
    while (at < list.size())
        {
            const size_t comma = list.find(',', at);
            const std::string tok = list.substr(at, comma == std::string::npos ? std::string::npos : comma - at);
            if (!tok.empty()) devices.push_back(std::atoi(tok.c_str()));
            if (comma == std::string::npos) break;
            at = comma + 1;
        }
    if (devices.empty()) return 0;
    static std::mutex mu;
    static std::map<std::string, unsigned> dealt;
    std::lock_guard<std::mutex> lk(mu);
    return devices[dealt[role + "|" + list]++ % devices.size()];
}

// base_pcps_acquisition.cc:38-66 without the command-line flag overrides
Acq_Conf get_acq_conf(const ConfigurationInterface* configuration, const std::string& role, double chip_rate, double opt_freq, uint32_t ms_per_code)
{
    Acq_Conf acq_parameters;
    acq_parameters.ms_per_code = ms_per_code;
    acq_parameters.sampled_ms = ms_per_code;  // default value
    acq_parameters.SetFromConfiguration(configuration, role, chip_rate, opt_freq);
    return acq_parameters;
}

// the fields of Acq_Conf the arithmetic depends on (acq_conf.h:33-87) -> the engine-side mirror
Hip_Acq_Conf to_hip_conf(const Acq_Conf& a)
{
    Hip_Acq_Conf h;
    h.fs_in = a.fs_in;
    h.resampled_fs = a.resampled_fs;
    h.samples_per_ms = a.samples_per_ms;
    h.threshold = a.threshold;
    h.pfa = a.pfa;
    h.pfa2 = a.pfa2;
    h.samples_per_code = a.samples_per_code;
    h.resampler_ratio = a.resampler_ratio;
    h.sampled_ms = a.sampled_ms;
    h.ms_per_code = a.ms_per_code;
    h.samples_per_chip = a.samples_per_chip;
    h.chips_per_second = a.chips_per_second;
    h.max_dwells = a.max_dwells;
    h.resampler_latency_samples = a.resampler_latency_samples;
    h.doppler_max = a.doppler_max;
    h.doppler_step = a.doppler_step;
    h.doppler_step2 = a.doppler_step2;
    h.num_doppler_bins_step2 = a.num_doppler_bins_step2;
    h.make_2_steps = a.make_2_steps;
    h.cshort = (a.item_type == "cshort");
    h.bit_transition_flag = a.bit_transition_flag;
    h.use_CFAR_algorithm_flag = a.use_CFAR_algorithm_flag;
    h.use_automatic_resampler = a.use_automatic_resampler;
    h.dump = a.dump;
    h.dump_filename = a.dump_filename;
    h.dump_channel = a.dump_channel;
    return h;
}
}  // namespace


BasePcpsAcquisitionHip::BasePcpsAcquisitionHip(const ConfigurationInterface* configuration, const std::string& role, unsigned int /*in_streams*/,
    unsigned int /*out_streams*/, double chip_rate, double opt_freq, double code_length_chips, uint32_t ms_per_code)
    : acq_parameters_(get_acq_conf(configuration, role, chip_rate, opt_freq, ms_per_code)),
      role_(role),
      // base_pcps_acquisition.cc:79-81
      vector_length_(static_cast<unsigned int>(std::floor(acq_parameters_.sampled_ms * acq_parameters_.samples_per_ms) * (acq_parameters_.bit_transition_flag ? 2.0 : 1.0))),
      code_length_(static_cast<unsigned int>(std::floor(static_cast<double>(acq_parameters_.resampled_fs) / (chip_rate / code_length_chips)))),
      code_(std::max(vector_length_, code_length_))
{
    // item types the engine ingests directly; the reference routes cbyte through a converter block (base_pcps_acquisition.cc:89-93)
    if (acq_parameters_.item_type == "gr_complex" || acq_parameters_.item_type == "cshort")
        {
            const int device = acquisition_device_for(configuration, role);
            device_ = device;
            const Hip_Acq_Conf hconf = to_hip_conf(acq_parameters_);
            auto runtime = acquisition_runtime_for(device, configuration->property(role + ".hip_shared_acquisition", -1), hconf,
                configuration->property(role + ".hip_shared_acquisition_channels", 64), configuration->property(role + ".hip_shared_acquisition_wait_us", 2000));
            acquisition_ = pcps_make_acquisition_hip(hconf, device, acq_parameters_.blocking_on_standby, std::move(runtime));
            if (!acquisition_->ok()) acquisition_.reset();  // item_size() == 0 -> the factory rejects the block instead of running without a GPU
        }
}


void BasePcpsAcquisitionHip::set_local_code()
{
    if (!acquisition_ || gnss_synchro_ == nullptr) return;
    // base_pcps_acquisition.cc:206-222
    std::vector<std::complex<float>> code(code_length_);
    const auto sampling_freq = acq_parameters_.use_automatic_resampler ? acq_parameters_.resampled_fs : acq_parameters_.fs_in;
    code_gen_complex_sampled(code, gnss_synchro_->PRN, static_cast<int32_t>(sampling_freq));
    const auto num_codes = acq_parameters_.sampled_ms / acq_parameters_.ms_per_code;
    for (unsigned int i = 0; i < num_codes; i++)
        {
            std::copy_n(code.data(), code_length_, code_.data() + static_cast<size_t>(i) * code_length_);
        }
    acquisition_->set_local_code(code_.data());
}


void BasePcpsAcquisitionHip::connect(gr::top_block_sptr /*top_block*/)
{
    // nothing to connect: gr_complex and cshort go straight into the block (base_pcps_acquisition.cc:128-131)
}


void BasePcpsAcquisitionHip::disconnect(gr::top_block_sptr /*top_block*/)
{
}
