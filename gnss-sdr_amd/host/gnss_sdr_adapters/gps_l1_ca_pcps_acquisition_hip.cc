/*!
 * \file gps_l1_ca_pcps_acquisition_hip.cc
 * \brief See the header.  BUILT ONLY INSIDE A gnss-sdr TREE.
 */
#include "gps_l1_ca_pcps_acquisition_hip.h"
#include "GPS_L1_CA.h"
#include "configuration_interface.h"
#include "gps_sdr_signal_replica.h"
#include <algorithm>
#include <cmath>

GpsL1CaPcpsAcquisitionHip::GpsL1CaPcpsAcquisitionHip(const ConfigurationInterface* configuration, const std::string& role,
    unsigned int /*in_streams*/, unsigned int /*out_streams*/) : role_(role)
{
    // the keys GPS_L1_CA_PCPS_Acquisition reads through Acq_Conf::SetFromConfiguration (acq_conf.cc:29-95)
    Hip_Acq_Conf& p = acq_parameters_;
    const int64_t fs_deprecated = configuration->property("GNSS-SDR.internal_fs_hz", static_cast<int64_t>(p.fs_in));
    p.fs_in = configuration->property("GNSS-SDR.internal_fs_sps", fs_deprecated);
    p.doppler_max = configuration->property(role + ".doppler_max", p.doppler_max);
    p.doppler_step = configuration->property(role + ".doppler_step", p.doppler_step);
    p.sampled_ms = configuration->property(role + ".coherent_integration_time_ms", p.sampled_ms);
    p.bit_transition_flag = configuration->property(role + ".bit_transition_flag", p.bit_transition_flag);
    p.max_dwells = configuration->property(role + ".max_dwells", p.max_dwells);
    p.threshold = configuration->property(role + ".threshold", p.threshold);
    p.pfa = configuration->property(role + ".pfa", p.pfa);
    if (p.pfa <= 0.0F) p.use_CFAR_algorithm_flag = false;  // acq_conf.cc:86-90
    p.ms_per_code = 1;
    p.chips_per_second = static_cast<uint32_t>(GPS_L1_CA_CODE_RATE_CPS);
    p.resampled_fs = p.fs_in;
    p.SetDerivedParams();
    const bool blocking_on_standby = configuration->property(role + ".blocking_on_standby", false);
    const int device = configuration->property(role + ".hip_device", 0);

    // gps_l1_ca_pcps_acquisition.cc:27-41: one code period at fs, repeated sampled_ms times
    const auto code_length = static_cast<unsigned int>(std::floor(static_cast<double>(p.fs_in) / (GPS_L1_CA_CODE_RATE_CPS / GPS_L1_CA_CODE_LENGTH_CHIPS)));
    vector_length_ = static_cast<unsigned int>(p.sampled_ms * p.samples_per_ms) * (p.bit_transition_flag ? 2U : 1U);
    code_.resize(std::max(vector_length_, code_length));
    acquisition_ = pcps_make_acquisition_hip(p, device, blocking_on_standby);
}


void GpsL1CaPcpsAcquisitionHip::set_local_code()
{
    // base_pcps_acquisition.cc:206-222 + gps_l1_ca_pcps_acquisition.cc:44-47
    const auto code_length = static_cast<unsigned int>(std::floor(static_cast<double>(acq_parameters_.fs_in) / (GPS_L1_CA_CODE_RATE_CPS / GPS_L1_CA_CODE_LENGTH_CHIPS)));
    std::vector<std::complex<float>> one_period(code_length);
    gps_l1_ca_code_gen_complex_sampled(one_period, gnss_synchro_->PRN, static_cast<int32_t>(acq_parameters_.fs_in), 0);
    for (unsigned int i = 0; i < acq_parameters_.sampled_ms; i++)
        {
            std::copy_n(one_period.begin(), code_length, code_.begin() + static_cast<size_t>(i) * code_length);
        }
    acquisition_->set_local_code(code_.data());
}


void GpsL1CaPcpsAcquisitionHip::connect(gr::top_block_sptr /*top_block*/) {}


void GpsL1CaPcpsAcquisitionHip::disconnect(gr::top_block_sptr /*top_block*/) {}
