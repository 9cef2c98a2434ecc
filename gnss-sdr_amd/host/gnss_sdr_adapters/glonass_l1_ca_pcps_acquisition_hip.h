/*!
 * \file glonass_l1_ca_pcps_acquisition_hip.h
 * \brief AcquisitionInterface adapter "GLONASS_L1_CA_PCPS_Acquisition_HIP" on an MI355X; the signal-specific part of the reference adapter
 *        (src/algorithms/acquisition/adapters/glonass_l1_ca_pcps_acquisition.cc:26-45) over BasePcpsAcquisitionHip.  BUILT ONLY INSIDE A gnss-sdr TREE.
 */
#ifndef GNSS_SDR_GLONASS_L1_CA_PCPS_ACQUISITION_HIP_H
#define GNSS_SDR_GLONASS_L1_CA_PCPS_ACQUISITION_HIP_H

#include "base_pcps_acquisition_hip.h"

class GlonassL1CaPcpsAcquisitionHip : public BasePcpsAcquisitionHip
{
public:
    GlonassL1CaPcpsAcquisitionHip(const ConfigurationInterface* configuration, const std::string& role, unsigned int in_streams, unsigned int out_streams);
    ~GlonassL1CaPcpsAcquisitionHip() override = default;
    std::string implementation() override { return "GLONASS_L1_CA_PCPS_Acquisition_HIP"; }

private:
    void code_gen_complex_sampled(own::span<std::complex<float>> dest, uint32_t prn, int32_t sampling_freq) override;
};

#endif  // GNSS_SDR_GLONASS_L1_CA_PCPS_ACQUISITION_HIP_H
