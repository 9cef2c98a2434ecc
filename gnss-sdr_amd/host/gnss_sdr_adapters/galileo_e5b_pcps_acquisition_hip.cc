/*!
 * \file galileo_e5b_pcps_acquisition_hip.cc
 * \brief See the header.  BUILT ONLY INSIDE A gnss-sdr TREE.
 */
#include "galileo_e5b_pcps_acquisition_hip.h"
#include "Galileo_E5b.h"
#include "configuration_interface.h"
#include "galileo_e5_signal_replica.h"
#include <array>

GalileoE5bPcpsAcquisitionHip::GalileoE5bPcpsAcquisitionHip(const ConfigurationInterface* configuration, const std::string& role, unsigned int in_streams,
    unsigned int out_streams)
    : BasePcpsAcquisitionHip(configuration, role, in_streams, out_streams, GALILEO_E5B_CODE_CHIP_RATE_CPS, GALILEO_E5B_OPT_ACQ_FS_SPS, GALILEO_E5B_CODE_LENGTH_CHIPS,
          GALILEO_E5B_CODE_PERIOD_MS),
      acq_pilot_(configuration->property(role + ".acquire_pilot", false)),
      acq_iq_(configuration->property(role + ".acquire_iq", false))
{
}


void GalileoE5bPcpsAcquisitionHip::code_gen_complex_sampled(own::span<std::complex<float>> dest, uint32_t prn, int32_t sampling_freq)
{
    // galileo_e5b_pcps_acquisition.cc:45-64: "7X" both components, "7Q" pilot, "7I" data
    std::array<char, 3> signal = {{'7', 'I', '\0'}};
    if (acq_iq_)
        {
            signal[1] = 'X';
        }
    else if (acq_pilot_)
        {
            signal[1] = 'Q';
        }
    galileo_e5_b_code_gen_complex_sampled(dest, prn, signal, sampling_freq, 0);
}
