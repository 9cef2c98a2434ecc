/*!
 * \file qzss_l5i_pcps_acquisition_hip.h
 * \brief AcquisitionInterface adapter "QZSS_L5i_PCPS_Acquisition_HIP" on an MI355X; the signal-specific part of the reference adapter
 *        (src/algorithms/acquisition/adapters/qzss_l5i_pcps_acquisition.cc:27-46) over BasePcpsAcquisitionHip.  BUILT ONLY INSIDE A gnss-sdr TREE.
 */
#ifndef GNSS_SDR_QZSS_L5I_PCPS_ACQUISITION_HIP_H
#define GNSS_SDR_QZSS_L5I_PCPS_ACQUISITION_HIP_H

#include "base_pcps_acquisition_hip.h"

class QzssL5iPcpsAcquisitionHip : public BasePcpsAcquisitionHip
{
public:
    QzssL5iPcpsAcquisitionHip(const ConfigurationInterface* configuration, const std::string& role, unsigned int in_streams, unsigned int out_streams);
    ~QzssL5iPcpsAcquisitionHip() override = default;
    std::string implementation() override { return "QZSS_L5i_PCPS_Acquisition_HIP"; }

private:
    void code_gen_complex_sampled(own::span<std::complex<float>> dest, uint32_t prn, int32_t sampling_freq) override;
};

#endif  // GNSS_SDR_QZSS_L5I_PCPS_ACQUISITION_HIP_H
