/*!
 * \file glonass_l2_ca_pcps_acquisition_hip.cc
 * \brief See the header.  BUILT ONLY INSIDE A gnss-sdr TREE.
 */
#include "glonass_l2_ca_pcps_acquisition_hip.h"
#include "GLONASS_L1_L2_CA.h"
#include "glonass_l2_signal_replica.h"

GlonassL2CaPcpsAcquisitionHip::GlonassL2CaPcpsAcquisitionHip(const ConfigurationInterface* configuration, const std::string& role, unsigned int in_streams, unsigned int out_streams)
    : BasePcpsAcquisitionHip(configuration, role, in_streams, out_streams, GLONASS_L2_CA_CODE_RATE_CPS, 100e6, GLONASS_L2_CA_CODE_LENGTH_CHIPS, GLONASS_L2_CA_CODE_PERIOD_MS)
{
}


void GlonassL2CaPcpsAcquisitionHip::code_gen_complex_sampled(own::span<std::complex<float>> dest, uint32_t /*prn*/, int32_t sampling_freq)
{
    glonass_l2_ca_code_gen_complex_sampled(dest, sampling_freq, 0);
}
