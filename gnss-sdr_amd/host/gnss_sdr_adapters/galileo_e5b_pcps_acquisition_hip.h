/*!
 * \file galileo_e5b_pcps_acquisition_hip.h
 * \brief AcquisitionInterface adapter "Galileo_E5b_Pcps_Acquisition_HIP" on an MI355X; the signal-specific part of the reference adapter
 *        (src/algorithms/acquisition/adapters/galileo_e5b_pcps_acquisition.cc:27-65: acquire_pilot / acquire_iq select the replica
 *        component) over BasePcpsAcquisitionHip.  BUILT ONLY INSIDE A gnss-sdr TREE.
 */
#ifndef GNSS_SDR_GALILEO_E5B_PCPS_ACQUISITION_HIP_H
#define GNSS_SDR_GALILEO_E5B_PCPS_ACQUISITION_HIP_H

#include "base_pcps_acquisition_hip.h"

class GalileoE5bPcpsAcquisitionHip : public BasePcpsAcquisitionHip
{
public:
    GalileoE5bPcpsAcquisitionHip(const ConfigurationInterface* configuration, const std::string& role, unsigned int in_streams, unsigned int out_streams);
    ~GalileoE5bPcpsAcquisitionHip() override = default;
    std::string implementation() override { return "Galileo_E5b_Pcps_Acquisition_HIP"; }

private:
    void code_gen_complex_sampled(own::span<std::complex<float>> dest, uint32_t prn, int32_t sampling_freq) override;
    const bool acq_pilot_;
    const bool acq_iq_;
};

#endif  // GNSS_SDR_GALILEO_E5B_PCPS_ACQUISITION_HIP_H
