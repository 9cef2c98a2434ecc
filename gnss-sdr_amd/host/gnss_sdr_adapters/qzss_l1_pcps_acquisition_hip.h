/*!
 * \file qzss_l1_pcps_acquisition_hip.h
 * \brief AcquisitionInterface adapter "QZSS_L1_PCPS_Acquisition_HIP" on an MI355X; the signal-specific part of the reference adapter
 *        (src/algorithms/acquisition/adapters/qzss_l1_pcps_acquisition.cc:27-46) over BasePcpsAcquisitionHip.  BUILT ONLY INSIDE A gnss-sdr TREE.
 */
#ifndef GNSS_SDR_QZSS_L1_PCPS_ACQUISITION_HIP_H
#define GNSS_SDR_QZSS_L1_PCPS_ACQUISITION_HIP_H

#include "base_pcps_acquisition_hip.h"

class QzssL1PcpsAcquisitionHip : public BasePcpsAcquisitionHip
{
public:
    QzssL1PcpsAcquisitionHip(const ConfigurationInterface* configuration, const std::string& role, unsigned int in_streams, unsigned int out_streams);
    ~QzssL1PcpsAcquisitionHip() override = default;
    std::string implementation() override { return "QZSS_L1_PCPS_Acquisition_HIP"; }

private:
    void code_gen_complex_sampled(own::span<std::complex<float>> dest, uint32_t prn, int32_t sampling_freq) override;
};

#endif  // GNSS_SDR_QZSS_L1_PCPS_ACQUISITION_HIP_H
