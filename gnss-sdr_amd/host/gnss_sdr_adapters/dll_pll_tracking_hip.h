/*!
 * \file dll_pll_tracking_hip.h
 * \brief TrackingInterface adapters "<reference name>_HIP" for every signal the reference tracks with dll_pll_veml_tracking
 *        (GPS L1 / L2C / L5, Galileo E1 / E5a / E5b / E6, GLONASS L1 / L2, BeiDou B1I / B3I, QZSS L1 / L5): the reference's own DLL/PLL
 *        adapters (src/algorithms/tracking/adapters/gps_l1_ca_dll_pll_tracking.h:37-58 and its twelve siblings over
 *        base_dll_pll_tracking.h:39-121) with the MI355X multicorrelator inside.
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

#include "beidou_b1i_dll_pll_tracking.h"
#include "beidou_b3i_dll_pll_tracking.h"
#include "configuration_interface.h"
#include "galileo_e1_dll_pll_veml_tracking.h"
#include "galileo_e5a_dll_pll_tracking.h"
#include "galileo_e5b_dll_pll_tracking.h"
#include "galileo_e6_dll_pll_tracking.h"
#include "glonass_l1_ca_dll_pll_tracking.h"
#include "glonass_l2_ca_dll_pll_tracking.h"
#include "gps_l1_ca_dll_pll_tracking.h"
#include "gps_l2_m_dll_pll_tracking.h"
#include "gps_l5_dll_pll_tracking.h"
#include "qzss_l1_dll_pll_tracking.h"
#include "qzss_l5_dll_pll_tracking.h"
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
GSH_DECLARE_TRACKING_HIP_ADAPTER(GpsL2MDllPllTrackingHip, GpsL2MDllPllTracking, "GPS_L2_M_DLL_PLL_Tracking_HIP");
GSH_DECLARE_TRACKING_HIP_ADAPTER(GalileoE5aDllPllTrackingHip, GalileoE5aDllPllTracking, "Galileo_E5a_DLL_PLL_Tracking_HIP");
GSH_DECLARE_TRACKING_HIP_ADAPTER(GalileoE5bDllPllTrackingHip, GalileoE5bDllPllTracking, "Galileo_E5b_DLL_PLL_Tracking_HIP");
GSH_DECLARE_TRACKING_HIP_ADAPTER(GalileoE6DllPllTrackingHip, GalileoE6DllPllTracking, "Galileo_E6_DLL_PLL_Tracking_HIP");
GSH_DECLARE_TRACKING_HIP_ADAPTER(GlonassL1CaDllPllTrackingHip, GlonassL1CaDllPllTracking, "GLONASS_L1_CA_DLL_PLL_Tracking_HIP");
GSH_DECLARE_TRACKING_HIP_ADAPTER(GlonassL2CaDllPllTrackingHip, GlonassL2CaDllPllTracking, "GLONASS_L2_CA_DLL_PLL_Tracking_HIP");
GSH_DECLARE_TRACKING_HIP_ADAPTER(BeidouB1iDllPllTrackingHip, BeidouB1iDllPllTracking, "BEIDOU_B1I_DLL_PLL_Tracking_HIP");
GSH_DECLARE_TRACKING_HIP_ADAPTER(BeidouB3iDllPllTrackingHip, BeidouB3iDllPllTracking, "BEIDOU_B3I_DLL_PLL_Tracking_HIP");
GSH_DECLARE_TRACKING_HIP_ADAPTER(QzssL1DllPllTrackingHip, QzssL1DllPllTracking, "QZSS_L1_CA_DLL_PLL_Tracking_HIP");
GSH_DECLARE_TRACKING_HIP_ADAPTER(QzssL5DllPllTrackingHip, QzssL5DllPllTracking, "QZSS_L5_DLL_PLL_Tracking_HIP");

#endif  // GNSS_SDR_DLL_PLL_TRACKING_HIP_H
