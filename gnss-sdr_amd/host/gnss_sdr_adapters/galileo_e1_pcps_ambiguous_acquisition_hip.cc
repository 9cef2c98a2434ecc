/*!
 * \file galileo_e1_pcps_ambiguous_acquisition_hip.cc
 * \brief See the header.  BUILT ONLY INSIDE A gnss-sdr TREE.
 */
#include "galileo_e1_pcps_ambiguous_acquisition_hip.h"
#include "Galileo_E1.h"
#include "configuration_interface.h"
#include "galileo_e1_signal_replica.h"
#include <array>

GalileoE1PcpsAmbiguousAcquisitionHip::GalileoE1PcpsAmbiguousAcquisitionHip(const ConfigurationInterface* configuration, const std::string& role,
    unsigned int in_streams, unsigned int out_streams)
    : BasePcpsAcquisitionHip(configuration, role, in_streams, out_streams, GALILEO_E1_CODE_CHIP_RATE_CPS, GALILEO_E1_OPT_ACQ_FS_SPS,
          GALILEO_E1_B_CODE_LENGTH_CHIPS, GALILEO_E1_CODE_PERIOD_MS),
      acquire_pilot_(configuration->property(role + ".acquire_pilot", false)),
      cboc_(configuration->property(role + ".cboc", false))
{
}


void GalileoE1PcpsAmbiguousAcquisitionHip::code_gen_complex_sampled(own::span<std::complex<float>> dest, uint32_t prn, int32_t sampling_freq)
{
    // galileo_e1_pcps_ambiguous_acquisition.cc:51-68
    std::array<char, 3> signal = {{'1', 'C', '\0'}};  // pilot component
    if (!acquire_pilot_)
        {
            signal[0] = gnss_synchro_->Signal[0];
            signal[1] = gnss_synchro_->Signal[1];
        }
    galileo_e1_code_gen_complex_sampled(dest, signal, cboc_, prn, sampling_freq, 0, false);
}
