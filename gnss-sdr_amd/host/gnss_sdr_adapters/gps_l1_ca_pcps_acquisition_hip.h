/*!
 * \file gps_l1_ca_pcps_acquisition_hip.h
 * \brief AcquisitionInterface adapter "GPS_L1_CA_PCPS_Acquisition_HIP": GPS L1 C/A PCPS acquisition on an MI355X.
 *
 * BUILT ONLY INSIDE A gnss-sdr TREE.  Derives directly from AcquisitionInterface
 * (src/core/interfaces/acquisition_interface.h:50-62) -- the reference's BasePcpsAcquisition cannot be reused because it
 * owns a concrete pcps_acquisition_sptr (base_pcps_acquisition.h:140).  Accepts the same configuration keys as
 * GPS_L1_CA_PCPS_Acquisition (acq_conf.cc:29-95) plus  <role>.hip_device  (default 0).
 * Precedent for a self-contained accelerator adapter in the reference: gps_l1_ca_dll_pll_tracking_gpu.cc:36-95.
 */
#ifndef GNSS_SDR_GPS_L1_CA_PCPS_ACQUISITION_HIP_H
#define GNSS_SDR_GPS_L1_CA_PCPS_ACQUISITION_HIP_H

#include "acquisition_interface.h"
#include "channel_fsm.h"
#include "gnss_synchro.h"
#include "pcps_acquisition_hip.h"
#include <complex>
#include <memory>
#include <string>
#include <vector>

class ConfigurationInterface;

class GpsL1CaPcpsAcquisitionHip : public AcquisitionInterface
{
public:
    GpsL1CaPcpsAcquisitionHip(const ConfigurationInterface* configuration, const std::string& role, unsigned int in_streams, unsigned int out_streams);
    ~GpsL1CaPcpsAcquisitionHip() override = default;

    std::string role() override { return role_; }
    std::string implementation() override { return "GPS_L1_CA_PCPS_Acquisition_HIP"; }
    size_t item_size() override { return sizeof(std::complex<float>); }
    void connect(gr::top_block_sptr top_block) override;
    void disconnect(gr::top_block_sptr top_block) override;
    gr::basic_block_sptr get_left_block() override { return acquisition_; }
    gr::basic_block_sptr get_right_block() override { return acquisition_; }

    void set_gnss_synchro(Gnss_Synchro* p_gnss_synchro) override
    {
        gnss_synchro_ = p_gnss_synchro;
        acquisition_->set_gnss_synchro(p_gnss_synchro);
    }
    void set_channel(unsigned int channel) override { acquisition_->set_channel(channel); }
    void set_channel_fsm(std::weak_ptr<ChannelFsm> channel_fsm) override { acquisition_->set_channel_fsm(std::move(channel_fsm)); }
    void set_doppler_center(int doppler_center) override { acquisition_->set_doppler_center(doppler_center); }
    void set_local_code() override;
    signed int mag() override { return static_cast<signed int>(acquisition_->mag()); }
    void reset() override { acquisition_->set_active(true); }
    void stop_acquisition() override { acquisition_->set_active(false); }
    void set_resampler_latency(uint32_t /*latency_samples*/) override {}
    void set_state(int state) { acquisition_->set_state(state); }

private:
    pcps_acquisition_hip_sptr acquisition_;
    Hip_Acq_Conf acq_parameters_;
    std::vector<std::complex<float>> code_;
    Gnss_Synchro* gnss_synchro_{nullptr};
    std::string role_;
    unsigned int vector_length_{0};
};

#endif  // GNSS_SDR_GPS_L1_CA_PCPS_ACQUISITION_HIP_H
