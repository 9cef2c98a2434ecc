/*!
 * \file dll_pll_tracking_hip.h
 * \brief TrackingInterface adapters over the MI355X device-closed DLL/PLL loop: DllPllTrackingHip and one class per signal
 *        of the reference's dll_pll_veml_tracking family (GPS L1 C/A, L2C, L5; Galileo E1, E5a, E5b, E6; BeiDou B1I, B3I; GLONASS L1, L2; QZSS L1, L5).
 *
 * These derive DIRECTLY from TrackingInterface (src/core/interfaces/tracking_interface.h:47-54): the reference's BaseDllPllTracking
 * holds a concrete dll_pll_veml_tracking_sptr (src/algorithms/tracking/adapters/base_dll_pll_tracking.h:115) and cannot carry another
 * block.  Precedent for a self-contained accelerator adapter: gps_l1_ca_dll_pll_tracking_gpu.cc:36-95.
 * Constructor signature, role(), implementation(), item_size(), connect / disconnect, get_left_block / get_right_block, set_channel,
 * set_gnss_synchro, start_tracking, stop_tracking behave as BaseDllPllTracking's (base_dll_pll_tracking.cc:25-112); configuration keys are
 * Dll_Pll_Conf's (dll_pll_conf.cc:48-158) plus
 *   <role>.hip_device            GPU index (0); pins the block to that GPU when <role>.hip_devices is given as well
 *   <role>.hip_devices           "0,1,2,...": the stream is kept resident in several GPUs of the node (ONE Hip_Sample_Ring over gsh_stream_group_*: one push over
 *                                PCIe into devices[0], RCCL replication over xGMI) and the blocks of the role are dealt to them in turn -- channel c -> GPU
 *                                c mod G; a Hip_Tracking_Runtime per device.  Default: hip_device only
 *   <role>.hip_periods_per_call  code periods one general_work call may take (1 = the reference's cadence, at most 64)
 *   <role>.hip_live              true (default): the loop kernel stays RESIDENT and follows the device sample ring, general_work reads finished records out of
 *                                page-locked memory (no launch per batch of periods); false: one launch advances every channel that has samples
 *   <role>.hip_shared_ring       which blocks share one Hip_Tracking_Runtime and, through it, one device sample ring (the stream crosses PCIe once for all of
 *                                them; one residency / launch advances them all):
 *                                  -2 (default)  every block of this role fed from the same RF chain (Channels_<signal>.RF_channel_ID) on the same device.  When
 *                                                any Channel<i>.RF_channel_ID routes a single channel to another chain the default falls back to -1: the ring
 *                                                de-duplicates pushes by absolute sample index, which only holds for blocks that see the same stream
 *                                  >= 0          every block on the device that names the same id, whatever its role (L1 C/A + E1 of one RF stream; or one id
 *                                                per RF chain in receivers with per-channel routing)
 *                                  -1            a runtime and ring of the block's own
 *   <role>.hip_periods_per_launch   launched mode: most code periods per channel one shared launch runs (max(16, hip_periods_per_call))
 *   <role>.hip_channels_per_launch  most channels of one loop configuration behind one device handle (64: one work-group each; a further handle beyond that)
 *   <role>.hip_work_groups_per_channel   launched mode (hip_live=false) only: 1 (default); 2 .. 8 compute units share every correlation window of a channel, 0 lets the
 *                                engine choose (gsh_trk_set_split) -- for long windows (Galileo E1: 4 ms) on a device with room; sums equal to rounding
 *   <role>.hip_record_timeout_ms    live mode: a channel whose next window has been resident this long without a record from the device is given up -- the block
 *                                publishes "events" 3 and the channel goes back to acquisition (1000; the device needs ~10 us per period)
 *   <role>.hip_register_input_buffer  page-lock the block's input buffer (lazily, as general_work shows it) so that pushes are DMAs without a
 *                                staging copy (true)
 * Factory registration (one `else if` per name, as gnss_block_factory.cc:657-662 does for the CUDA block): INTEGRATION.md section 2b.
 * An unusable block (unsupported item type / signal, or no GPU) is reported the reference's way: item_size() == 0
 * (gnss_block_factory.cc:1048-1052, channel.cc:96-100).
 */
#ifndef GNSS_SDR_DLL_PLL_TRACKING_HIP_H
#define GNSS_SDR_DLL_PLL_TRACKING_HIP_H

#include "dll_pll_conf.h"
#include "dll_pll_veml_tracking_hip.h"
#include "tracking_interface.h"
#include <gnuradio/top_block.h>
#include <cstddef>
#include <string>

class ConfigurationInterface;

class DllPllTrackingHip : public TrackingInterface
{
public:
    ~DllPllTrackingHip() override = default;
    inline std::string role() override { return role_; }
    inline size_t item_size() override { return item_size_; }
    void connect(gr::top_block_sptr top_block) override;
    void disconnect(gr::top_block_sptr top_block) override;
    gr::basic_block_sptr get_left_block() override;
    gr::basic_block_sptr get_right_block() override;
    void set_channel(unsigned int channel) override;
    void set_gnss_synchro(Gnss_Synchro* p_gnss_synchro) override;
    void start_tracking() override;
    void stop_tracking() override;

    //! the loop configuration the block runs with (what a Channel never needs; tests and monitors do)
    const gsh_trk_conf& trk_conf() const { return tracking_sptr_->trk_conf(); }
    const Dll_Pll_Conf& tracking_parameters() const { return trk_params_; }
    dll_pll_veml_tracking_hip_sptr block() const { return tracking_sptr_; }

protected:
    DllPllTrackingHip(const ConfigurationInterface* configuration, std::string role, unsigned int in_streams, unsigned int out_streams);
    Dll_Pll_Conf& config_params() { return trk_params_; }
    //! makes the GNU Radio block from the finished configuration (called by the signal classes' constructors)
    void create_tracking_block(const ConfigurationInterface* configuration);

private:
    Dll_Pll_Conf trk_params_;
    dll_pll_veml_tracking_hip_sptr tracking_sptr_;
    const std::string role_;
    size_t item_size_;
};

//! "GPS_L1_CA_DLL_PLL_Tracking_HIP" -- counterpart of GpsL1CaDllPllTracking (gps_l1_ca_dll_pll_tracking.cc:38-113)
class GpsL1CaDllPllTrackingHip : public DllPllTrackingHip
{
public:
    GpsL1CaDllPllTrackingHip(const ConfigurationInterface* configuration, const std::string& role, unsigned int in_streams, unsigned int out_streams);
    inline std::string implementation() override { return "GPS_L1_CA_DLL_PLL_Tracking_HIP"; }
};

//! "Galileo_E1_DLL_PLL_VEML_Tracking_HIP" -- counterpart of GalileoE1DllPllVemlTracking (galileo_e1_dll_pll_veml_tracking.cc:36-88)
class GalileoE1DllPllVemlTrackingHip : public DllPllTrackingHip
{
public:
    GalileoE1DllPllVemlTrackingHip(const ConfigurationInterface* configuration, const std::string& role, unsigned int in_streams, unsigned int out_streams);
    inline std::string implementation() override { return "Galileo_E1_DLL_PLL_VEML_Tracking_HIP"; }
};

//! "GPS_L5_DLL_PLL_Tracking_HIP" -- counterpart of GpsL5DllPllTracking (gps_l5_dll_pll_tracking.cc:36-90)
class GpsL5DllPllTrackingHip : public DllPllTrackingHip
{
public:
    GpsL5DllPllTrackingHip(const ConfigurationInterface* configuration, const std::string& role, unsigned int in_streams, unsigned int out_streams);
    inline std::string implementation() override { return "GPS_L5_DLL_PLL_Tracking_HIP"; }
};

//! "Galileo_E5a_DLL_PLL_Tracking_HIP" -- counterpart of GalileoE5aDllPllTracking (galileo_e5a_dll_pll_tracking.cc:36-85)
class GalileoE5aDllPllTrackingHip : public DllPllTrackingHip
{
public:
    GalileoE5aDllPllTrackingHip(const ConfigurationInterface* configuration, const std::string& role, unsigned int in_streams, unsigned int out_streams);
    inline std::string implementation() override { return "Galileo_E5a_DLL_PLL_Tracking_HIP"; }
};

// The remaining signals of the family: the same block, other constants (trk.cc:196-596) and replica generators (:812-1030).
#define GSH_DECLARE_TRACKING_ADAPTER(CLASS, NAME)                                                                                             \
    class CLASS : public DllPllTrackingHip                                                                                                   \
    {                                                                                                                                        \
    public:                                                                                                                                  \
        CLASS(const ConfigurationInterface* configuration, const std::string& role, unsigned int in_streams, unsigned int out_streams);     \
        inline std::string implementation() override { return NAME; }                                                                       \
    };
GSH_DECLARE_TRACKING_ADAPTER(GpsL2MDllPllTrackingHip, "GPS_L2_M_DLL_PLL_Tracking_HIP")            //!< gps_l2_m_dll_pll_tracking.cc
GSH_DECLARE_TRACKING_ADAPTER(GalileoE5bDllPllTrackingHip, "Galileo_E5b_DLL_PLL_Tracking_HIP")    //!< galileo_e5b_dll_pll_tracking.cc
GSH_DECLARE_TRACKING_ADAPTER(GalileoE6DllPllTrackingHip, "Galileo_E6_DLL_PLL_Tracking_HIP")      //!< galileo_e6_dll_pll_tracking.cc
GSH_DECLARE_TRACKING_ADAPTER(BeidouB1iDllPllTrackingHip, "BEIDOU_B1I_DLL_PLL_Tracking_HIP")      //!< beidou_b1i_dll_pll_tracking.cc
GSH_DECLARE_TRACKING_ADAPTER(BeidouB3iDllPllTrackingHip, "BEIDOU_B3I_DLL_PLL_Tracking_HIP")      //!< beidou_b3i_dll_pll_tracking.cc
GSH_DECLARE_TRACKING_ADAPTER(GlonassL1CaDllPllTrackingHip, "GLONASS_L1_CA_DLL_PLL_Tracking_HIP") //!< glonass_l1_ca_dll_pll_tracking.cc
GSH_DECLARE_TRACKING_ADAPTER(GlonassL2CaDllPllTrackingHip, "GLONASS_L2_CA_DLL_PLL_Tracking_HIP") //!< glonass_l2_ca_dll_pll_tracking.cc
GSH_DECLARE_TRACKING_ADAPTER(QzssL1DllPllTrackingHip, "QZSS_L1_CA_DLL_PLL_Tracking_HIP")         //!< qzss_l1_dll_pll_tracking.cc
GSH_DECLARE_TRACKING_ADAPTER(QzssL5DllPllTrackingHip, "QZSS_L5_DLL_PLL_Tracking_HIP")            //!< qzss_l5_dll_pll_tracking.cc
#undef GSH_DECLARE_TRACKING_ADAPTER

#endif  // GNSS_SDR_DLL_PLL_TRACKING_HIP_H
