/*!
 * \file gps_l2_m_pcps_acquisition_hip.cc
 * \brief See the header.  BUILT ONLY INSIDE A gnss-sdr TREE.
 */
#include "gps_l2_m_pcps_acquisition_hip.h"
#include "GPS_L2C.h"
#include "gps_l2c_signal_replica.h"

GpsL2MPcpsAcquisitionHip::GpsL2MPcpsAcquisitionHip(const ConfigurationInterface* configuration, const std::string& role, unsigned int in_streams, unsigned int out_streams)
    : BasePcpsAcquisitionHip(configuration, role, in_streams, out_streams, GPS_L2_M_CODE_RATE_CPS, GPS_L2C_OPT_ACQ_FS_SPS, GPS_L2_M_CODE_LENGTH_CHIPS, GPS_L2_M_CODE_PERIOD_MS)
{
}


void GpsL2MPcpsAcquisitionHip::code_gen_complex_sampled(own::span<std::complex<float>> dest, uint32_t prn, int32_t sampling_freq)
{
    gps_l2c_m_code_gen_complex_sampled(dest, prn, sampling_freq);
}
