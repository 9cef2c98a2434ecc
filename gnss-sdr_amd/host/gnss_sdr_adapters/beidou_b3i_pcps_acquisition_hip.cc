/*!
 * \file beidou_b3i_pcps_acquisition_hip.cc
 * \brief See the header.  BUILT ONLY INSIDE A gnss-sdr TREE.
 */
#include "beidou_b3i_pcps_acquisition_hip.h"
#include "Beidou_B3I.h"
#include "beidou_b3i_signal_replica.h"

BeidouB3iPcpsAcquisitionHip::BeidouB3iPcpsAcquisitionHip(const ConfigurationInterface* configuration, const std::string& role, unsigned int in_streams, unsigned int out_streams)
    : BasePcpsAcquisitionHip(configuration, role, in_streams, out_streams, BEIDOU_B3I_CODE_RATE_CPS, BEIDOU_B3I_OPT_ACQ_FS_SPS, BEIDOU_B3I_CODE_LENGTH_CHIPS, BEIDOU_B3I_CODE_PERIOD_MS)
{
}


void BeidouB3iPcpsAcquisitionHip::code_gen_complex_sampled(own::span<std::complex<float>> dest, uint32_t prn, int32_t sampling_freq)
{
    beidou_b3i_code_gen_complex_sampled(dest, prn, sampling_freq, 0);
}
