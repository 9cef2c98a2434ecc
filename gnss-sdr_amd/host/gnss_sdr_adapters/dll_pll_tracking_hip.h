/*!
 * \file dll_pll_tracking_hip.h
 * \brief TrackingInterface adapters "GPS_L1_CA_DLL_PLL_Tracking_HIP", "Galileo_E1_DLL_PLL_VEML_Tracking_HIP",
 *        "GPS_L5_DLL_PLL_Tracking_HIP": the reference's own DLL/PLL adapters
 *        (src/algorithms/tracking/adapters/gps_l1_ca_dll_pll_tracking.h:37-58, galileo_e1_dll_pll_veml_tracking.h,
 *        gps_l5_dll_pll_tracking.h over base_dll_pll_tracking.h:39-121) with the MI355X multicorrelator inside.
 *
 * BUILT ONLY INSIDE A gnss-sdr TREE, with -DENABLE_HIP_MI355X=1, i.e. with dll_pll_veml_tracking's two correlator members
 * (dll_pll_veml_tracking.h:94-95) declared as Hip_Multicorrelator_Real_Codes.  The 2 300 lines of
 * loop logic, lock detection, bit synchronisation and telemetry hand-over of the reference block run unchanged; every
 * Carrier_wipeoff_multicorrelator_resampler call (trk.cc:1236-1256) goes to the GPU.
 * What these adapters add to the plain reference adapters: an explicit implementation name for the configuration file and the
 * per-role device choice  <role>.hip_device  (handed to the correlator through GNSS_SDR_HIP_DEVICE before the block, and with
 * it the correlator, is constructed: trk.cc:652).
 * Precedent: gps_l1_ca_dll_pll_tracking_gpu.h:37-95 (the reference's CUDA adapter is a separate class for the same reason).
 */
#ifndef GNSS_SDR_DLL_PLL_TRACKING_HIP_H
#define GNSS_SDR_DLL_PLL_TRACKING_HIP_H

#if !ENABLE_HIP_MI355X
#error "dll_pll_tracking_hip.h needs -DENABLE_HIP_MI355X=1 (the tracking block must be compiled with the HIP correlator members)"
#endif

#include "configuration_interface.h"
#include "galileo_e1_dll_pll_veml_tracking.h"
#include "gps_l1_ca_dll_pll_tracking.h"
#include "gps_l5_dll_pll_tracking.h"
#include <cstdlib>
#include <string>

namespace hip_tracking_detail
{
//! runs before the reference adapter's constructor (base-from-member): publish the device the correlators must open
struct DeviceSelector
{
    DeviceSelector(const ConfigurationInterface* configuration, const std::string& role)
    {
        const int device = configuration->property(role + ".hip_device", 0);
        setenv("GNSS_SDR_HIP_DEVICE", std::to_string(device).c_str(), 1);
    }
};
}  // namespace hip_tracking_detail

#define GSH_DECLARE_TRACKING_HIP_ADAPTER(ClassName, RefAdapter, ImplName)                                                         \
    class ClassName : private hip_tracking_detail::DeviceSelector, public RefAdapter                                              \
    {                                                                                                                             \
    public:                                                                                                                       \
        ClassName(const ConfigurationInterface* configuration, const std::string& role, unsigned int in_streams, unsigned int out_streams) \
            : hip_tracking_detail::DeviceSelector(configuration, role), RefAdapter(configuration, role, in_streams, out_streams)  \
        {                                                                                                                         \
        }                                                                                                                         \
        inline std::string implementation() override { return ImplName; }                                                         \
    }

GSH_DECLARE_TRACKING_HIP_ADAPTER(GpsL1CaDllPllTrackingHip, GpsL1CaDllPllTracking, "GPS_L1_CA_DLL_PLL_Tracking_HIP");
GSH_DECLARE_TRACKING_HIP_ADAPTER(GalileoE1DllPllVemlTrackingHip, GalileoE1DllPllVemlTracking, "Galileo_E1_DLL_PLL_VEML_Tracking_HIP");
GSH_DECLARE_TRACKING_HIP_ADAPTER(GpsL5DllPllTrackingHip, GpsL5DllPllTracking, "GPS_L5_DLL_PLL_Tracking_HIP");

#endif  // GNSS_SDR_DLL_PLL_TRACKING_HIP_H
