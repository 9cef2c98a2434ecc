/*!
 * \file qzss_l5i_pcps_acquisition_hip.cc
 * \brief See the header.  BUILT ONLY INSIDE A gnss-sdr TREE.
 */
#include "qzss_l5i_pcps_acquisition_hip.h"
#include "qzss.h"
#include "qzss_signal_replica.h"

QzssL5iPcpsAcquisitionHip::QzssL5iPcpsAcquisitionHip(const ConfigurationInterface* configuration, const std::string& role, unsigned int in_streams, unsigned int out_streams)
    : BasePcpsAcquisitionHip(configuration, role, in_streams, out_streams, QZSS_L5_CHIP_RATE, QZSS_L5_OPT_ACQ_FS_SPS, QZSS_L5_CODE_LENGTH, QZSS_L5I_PERIOD_MS)
{
}


void QzssL5iPcpsAcquisitionHip::code_gen_complex_sampled(own::span<std::complex<float>> dest, uint32_t prn, int32_t sampling_freq)
{
    qzss_l5i_code_gen_complex_sampled(dest, prn, sampling_freq);
}
