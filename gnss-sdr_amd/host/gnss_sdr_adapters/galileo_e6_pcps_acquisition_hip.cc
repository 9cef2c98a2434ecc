/*!
 * \file galileo_e6_pcps_acquisition_hip.cc
 * \brief See the header.  BUILT ONLY INSIDE A gnss-sdr TREE.
 */
#include "galileo_e6_pcps_acquisition_hip.h"
#include "Galileo_E6.h"
#include "galileo_e6_signal_replica.h"

GalileoE6PcpsAcquisitionHip::GalileoE6PcpsAcquisitionHip(const ConfigurationInterface* configuration, const std::string& role, unsigned int in_streams, unsigned int out_streams)
    : BasePcpsAcquisitionHip(configuration, role, in_streams, out_streams, GALILEO_E6_B_CODE_CHIP_RATE_CPS, GALILEO_E6_OPT_ACQ_FS_SPS, GALILEO_E6_B_CODE_LENGTH_CHIPS, GALILEO_E6_CODE_PERIOD_MS)
{
}


void GalileoE6PcpsAcquisitionHip::code_gen_complex_sampled(own::span<std::complex<float>> dest, uint32_t prn, int32_t sampling_freq)
{
    galileo_e6_b_code_gen_complex_sampled(dest, prn, sampling_freq, 0);
}
