/*
 * TEST INFRASTRUCTURE ONLY -- see ref_chan_api.h.  Nothing here computes: it names the reference's adapter classes (compiled from
 * src/algorithms/acquisition/adapters and src/algorithms/tracking/adapters where they lie) for the implementation strings the channel test uses.
 */
#include "ref_chan_api.h"
#include "complex_byte_to_float_x2.h"
#include "galileo_e1_dll_pll_veml_tracking.h"
#include "galileo_e1_pcps_ambiguous_acquisition.h"
#include "gps_l1_ca_dll_pll_tracking.h"
#include "gps_l1_ca_pcps_acquisition.h"

// base_pcps_acquisition.cc:89-93 builds this converter for item_type cbyte only; the tests feed gr_complex (complex_byte_to_float_x2.cc needs upstream VOLK)
complex_byte_to_float_x2_sptr make_complex_byte_to_float_x2() { return complex_byte_to_float_x2_sptr(); }

std::shared_ptr<AcquisitionInterface> refchan_make_acquisition(const std::string& implementation, const ConfigurationInterface* configuration, const std::string& role,
    unsigned int in_streams, unsigned int out_streams)
{
    if (implementation == "GPS_L1_CA_PCPS_Acquisition") return std::make_shared<GpsL1CaPcpsAcquisition>(configuration, role, in_streams, out_streams);
    if (implementation == "Galileo_E1_PCPS_Ambiguous_Acquisition") return std::make_shared<GalileoE1PcpsAmbiguousAcquisition>(configuration, role, in_streams, out_streams);
    return nullptr;
}

std::shared_ptr<TrackingInterface> refchan_make_tracking(const std::string& implementation, const ConfigurationInterface* configuration, const std::string& role,
    unsigned int in_streams, unsigned int out_streams)
{
    if (implementation == "GPS_L1_CA_DLL_PLL_Tracking") return std::make_shared<GpsL1CaDllPllTracking>(configuration, role, in_streams, out_streams);
    if (implementation == "Galileo_E1_DLL_PLL_VEML_Tracking") return std::make_shared<GalileoE1DllPllVemlTracking>(configuration, role, in_streams, out_streams);
    return nullptr;
}
