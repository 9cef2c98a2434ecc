/*!
 * \file beidou_b1i_pcps_acquisition_hip.cc
 * \brief See the header.  BUILT ONLY INSIDE A gnss-sdr TREE.
 */
#include "beidou_b1i_pcps_acquisition_hip.h"
#include "Beidou_B1I.h"
#include "beidou_b1i_signal_replica.h"

BeidouB1iPcpsAcquisitionHip::BeidouB1iPcpsAcquisitionHip(const ConfigurationInterface* configuration, const std::string& role, unsigned int in_streams, unsigned int out_streams)
    : BasePcpsAcquisitionHip(configuration, role, in_streams, out_streams, BEIDOU_B1I_CODE_RATE_CPS, BEIDOU_B1I_OPT_ACQ_FS_SPS, BEIDOU_B1I_CODE_LENGTH_CHIPS, BEIDOU_B1I_CODE_PERIOD_MS)
{
}


void BeidouB1iPcpsAcquisitionHip::code_gen_complex_sampled(own::span<std::complex<float>> dest, uint32_t prn, int32_t sampling_freq)
{
    beidou_b1i_code_gen_complex_sampled(dest, prn, sampling_freq, 0);
}
