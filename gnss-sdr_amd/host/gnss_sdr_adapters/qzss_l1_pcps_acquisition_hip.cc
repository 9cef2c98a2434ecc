/*!
 * \file qzss_l1_pcps_acquisition_hip.cc
 * \brief See the header.  BUILT ONLY INSIDE A gnss-sdr TREE.
 */
#include "qzss_l1_pcps_acquisition_hip.h"
#include "qzss.h"
#include "qzss_signal_replica.h"

QzssL1PcpsAcquisitionHip::QzssL1PcpsAcquisitionHip(const ConfigurationInterface* configuration, const std::string& role, unsigned int in_streams, unsigned int out_streams)
    : BasePcpsAcquisitionHip(configuration, role, in_streams, out_streams, QZSS_L1_CHIP_RATE, QZSS_L1_OPT_ACQ_FS_SPS, QZSS_L1_CODE_LENGTH, QZSS_L1_PERIOD_MS)
{
}


void QzssL1PcpsAcquisitionHip::code_gen_complex_sampled(own::span<std::complex<float>> dest, uint32_t prn, int32_t sampling_freq)
{
    qzss_l1_code_gen_complex_sampled(dest, prn, sampling_freq);
}
