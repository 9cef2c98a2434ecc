/*!
 * \file gps_l2_m_pcps_acquisition_hip.h
 * \brief AcquisitionInterface adapter "GPS_L2_M_PCPS_Acquisition_HIP" on an MI355X; the signal-specific part of the reference adapter
 *        (src/algorithms/acquisition/adapters/gps_l2_m_pcps_acquisition.cc:27-47: 20 ms L2C (M) code) over BasePcpsAcquisitionHip.
 *        BUILT ONLY INSIDE A gnss-sdr TREE.
 */
#ifndef GNSS_SDR_GPS_L2_M_PCPS_ACQUISITION_HIP_H
#define GNSS_SDR_GPS_L2_M_PCPS_ACQUISITION_HIP_H

#include "base_pcps_acquisition_hip.h"

class GpsL2MPcpsAcquisitionHip : public BasePcpsAcquisitionHip
{
public:
    GpsL2MPcpsAcquisitionHip(const ConfigurationInterface* configuration, const std::string& role, unsigned int in_streams, unsigned int out_streams);
    ~GpsL2MPcpsAcquisitionHip() override = default;
    std::string implementation() override { return "GPS_L2_M_PCPS_Acquisition_HIP"; }

private:
    void code_gen_complex_sampled(own::span<std::complex<float>> dest, uint32_t prn, int32_t sampling_freq) override;
};

#endif  // GNSS_SDR_GPS_L2_M_PCPS_ACQUISITION_HIP_H
