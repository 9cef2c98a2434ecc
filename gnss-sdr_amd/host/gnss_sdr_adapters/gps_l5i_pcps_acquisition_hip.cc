/*!
 * \file gps_l5i_pcps_acquisition_hip.cc
 * \brief See the header.  BUILT ONLY INSIDE A gnss-sdr TREE.
 */
#include "gps_l5i_pcps_acquisition_hip.h"
#include "GPS_L5.h"
#include "gps_l5_signal_replica.h"

GpsL5iPcpsAcquisitionHip::GpsL5iPcpsAcquisitionHip(const ConfigurationInterface* configuration, const std::string& role, unsigned int in_streams, unsigned int out_streams)
    : BasePcpsAcquisitionHip(configuration, role, in_streams, out_streams, GPS_L5I_CODE_RATE_CPS, GPS_L5_OPT_ACQ_FS_SPS, GPS_L5I_CODE_LENGTH_CHIPS, GPS_L5I_PERIOD_MS)
{
}


void GpsL5iPcpsAcquisitionHip::code_gen_complex_sampled(own::span<std::complex<float>> dest, uint32_t prn, int32_t sampling_freq)
{
    gps_l5i_code_gen_complex_sampled(dest, prn, sampling_freq);
}
