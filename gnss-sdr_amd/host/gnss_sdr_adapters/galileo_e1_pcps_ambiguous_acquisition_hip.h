/*!
 * \file galileo_e1_pcps_ambiguous_acquisition_hip.h
 * \brief AcquisitionInterface adapter "Galileo_E1_PCPS_Ambiguous_Acquisition_HIP" on an MI355X; the signal-specific part of the
 *        reference adapter (src/algorithms/acquisition/adapters/galileo_e1_pcps_ambiguous_acquisition.cc:22-68: acquire_pilot,
 *        cboc) over BasePcpsAcquisitionHip.  BUILT ONLY INSIDE A gnss-sdr TREE.
 */
#ifndef GNSS_SDR_GALILEO_E1_PCPS_AMBIGUOUS_ACQUISITION_HIP_H
#define GNSS_SDR_GALILEO_E1_PCPS_AMBIGUOUS_ACQUISITION_HIP_H

#include "base_pcps_acquisition_hip.h"

class GalileoE1PcpsAmbiguousAcquisitionHip : public BasePcpsAcquisitionHip
{
public:
    GalileoE1PcpsAmbiguousAcquisitionHip(const ConfigurationInterface* configuration, const std::string& role, unsigned int in_streams, unsigned int out_streams);
    ~GalileoE1PcpsAmbiguousAcquisitionHip() override = default;
    std::string implementation() override { return "Galileo_E1_PCPS_Ambiguous_Acquisition_HIP"; }

private:
    void code_gen_complex_sampled(own::span<std::complex<float>> dest, uint32_t prn, int32_t sampling_freq) override;
    const bool acquire_pilot_;
    const bool cboc_;
};

#endif  // GNSS_SDR_GALILEO_E1_PCPS_AMBIGUOUS_ACQUISITION_HIP_H
