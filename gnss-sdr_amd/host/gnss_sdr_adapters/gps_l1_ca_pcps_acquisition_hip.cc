/*!
 * \file gps_l1_ca_pcps_acquisition_hip.cc
 * \brief See the header.  BUILT ONLY INSIDE A gnss-sdr TREE.
 */
#include "gps_l1_ca_pcps_acquisition_hip.h"
#include "GPS_L1_CA.h"
#include "gps_sdr_signal_replica.h"

GpsL1CaPcpsAcquisitionHip::GpsL1CaPcpsAcquisitionHip(const ConfigurationInterface* configuration, const std::string& role, unsigned int in_streams, unsigned int out_streams)
    : BasePcpsAcquisitionHip(configuration, role, in_streams, out_streams, GPS_L1_CA_CODE_RATE_CPS, GPS_L1_CA_OPT_ACQ_FS_SPS, GPS_L1_CA_CODE_LENGTH_CHIPS, GPS_L1_CA_CODE_PERIOD_MS)
{
}


void GpsL1CaPcpsAcquisitionHip::code_gen_complex_sampled(own::span<std::complex<float>> dest, uint32_t prn, int32_t sampling_freq)
{
    gps_l1_ca_code_gen_complex_sampled(dest, prn, sampling_freq, 0);
}
