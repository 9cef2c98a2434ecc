/*!
 * \file dll_pll_conf_hip.h
 * \brief Dll_Pll_Conf -> gsh_trk_conf: what dll_pll_veml_tracking's constructor derives per signal (trk.cc:196-700) and what
 *        start_tracking generates as local replicas (trk.cc:812-1030), for the device-closed loop.
 *
 * BUILT INSIDE A gnss-sdr TREE (or against its headers): Dll_Pll_Conf (src/algorithms/tracking/libs/dll_pll_conf.h:33-90), the signal
 * constant headers (GPS_L1_CA.h, GPS_L5.h, Galileo_E1.h, Galileo_E5a.h), the S-curve helpers of tracking_discriminators.h and the PRN
 * generators are the reference's own.
 */
#ifndef GNSS_SDR_DLL_PLL_CONF_HIP_H
#define GNSS_SDR_DLL_PLL_CONF_HIP_H

#include "dll_pll_conf.h"
#include "gnss_sdr_hip.h"
#include <cstdint>
#include <string>
#include <vector>

//! what the block needs besides the loop configuration
struct Hip_Trk_Signal
{
    std::string system_name;    //!< d_systemName
    std::string signal_type;    //!< d_signal_type ("1C", "1B", "L5", "5X")
    int32_t correlation_length_ms{1};  //!< d_correlation_length_ms (Gnss_Synchro::correlation_length_ms)
    bool interchange_iq{false};        //!< d_interchange_iq: Prompt_I / Prompt_Q swapped in the published symbol (trk.cc:2219-2228)
    bool per_prn_secondary{false};     //!< the secondary code depends on the PRN and is set at start_tracking (Galileo E5a pilot, trk.cc:857)
};

/*! Fills `out` (zero-initialised first) from the reference configuration object the way the tracking block's constructor does.
    Supported: GPS "1C", GPS "L5" (data or pilot), Galileo "1B" (data or pilot), Galileo "5X" (data or pilot).
    Returns false and says why for anything else (the factory then reports an unusable block, item_size() == 0). */
bool hip_fill_trk_conf(const Dll_Pll_Conf& p, gsh_trk_conf* out, Hip_Trk_Signal* sig, std::string* why);

/*! start_tracking's local replicas (trk.cc:812-1030) for satellite `prn`; data_code is filled only with track_pilot.  For signals whose
    secondary code depends on the PRN, conf->secondary_code is rewritten as start_tracking does. */
bool hip_make_tracking_codes(const Hip_Trk_Signal& sig, gsh_trk_conf* conf, uint32_t prn, const char signal[3], std::vector<float>* code,
    std::vector<float>* data_code, std::string* why);

#endif  // GNSS_SDR_DLL_PLL_CONF_HIP_H
