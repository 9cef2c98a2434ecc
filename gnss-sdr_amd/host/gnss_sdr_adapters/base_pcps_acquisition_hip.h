/*!
 * \file base_pcps_acquisition_hip.h
 * \brief AcquisitionInterface adapter base for PCPS acquisition on an MI355X: the counterpart of gnss-sdr's BasePcpsAcquisition
 *        (src/algorithms/acquisition/adapters/base_pcps_acquisition.h:52-141), which cannot be reused because it owns a
 *        concrete pcps_acquisition_sptr (:140).
 *
 * BUILT ONLY INSIDE A gnss-sdr TREE (GNU Radio, Gnss_Synchro, ChannelFsm, Acq_Conf); tests/host/mock_gnuradio/ lets this
 * repository compile and drive it against the reference's own interface headers (tests/test_adapters_*.py).
 * Same constructor signature, same configuration keys (Acq_Conf::SetFromConfiguration, acq_conf.cc:29-95 -- including
 * make_two_steps / second_nbins / second_doppler_step / pfa_second_step and item_type = cshort), plus
 *   <role>.hip_device                        GPU index (0); pins the block when <role>.hip_devices is given as well
 *   <role>.hip_devices                       "0,1,2,...": the blocks of the role are dealt over these GPUs in the order the factory builds them: channel c searches on
 *                                            GPU c mod G, next to its tracking block (same key of the tracking role)
 *   <role>.hip_shared_acquisition            id >= 0: channels with the same id (and device, and dwell geometry) share one Hip_Acquisition_Runtime --
 *                                            blocks that search at the same time cut the stream on a common grid and join ONE dwell batch: the
 *                                            Doppler-wiped forward transforms are computed once for all of them (-1, default: every block on its own)
 *   <role>.hip_shared_acquisition_channels   slots of the shared handle (64)
 *   <role>.hip_shared_acquisition_wait_us    how long the first channel of a batch waits for the others that announced the same window (2000)
 * Signal-specific adapters only provide code_gen_complex_sampled(), exactly as in the reference (:133).
 * Precedent for a self-contained accelerator adapter in the reference: gps_l1_ca_dll_pll_tracking_gpu.cc:36-95.
 */
#ifndef GNSS_SDR_BASE_PCPS_ACQUISITION_HIP_H
#define GNSS_SDR_BASE_PCPS_ACQUISITION_HIP_H

#include "acq_conf.h"
#include "acquisition_interface.h"
#include "channel_fsm.h"
#include "gnss_synchro.h"
#include "pcps_acquisition_hip.h"
#include <complex>
#include <cstdint>
#include <memory>
#include <string>
#include <utility>
#include <vector>
#if HAS_STD_SPAN
#include <span>
namespace own = std;
#else
#include <gsl-lite/gsl-lite.hpp>
namespace own = gsl_lite;
#endif

class ConfigurationInterface;

class BasePcpsAcquisitionHip : public AcquisitionInterface
{
public:
    BasePcpsAcquisitionHip(const ConfigurationInterface* configuration, const std::string& role, unsigned int in_streams, unsigned int out_streams,
        double chip_rate, double opt_freq, double code_length_chips, uint32_t ms_per_code);
    ~BasePcpsAcquisitionHip() override = default;

    std::string role() override { return role_; }
    size_t item_size() override { return acquisition_ ? acq_parameters_.it_size : 0; }  // 0 = unusable block (gnss_block_factory.cc:1048-1052)
    void connect(gr::top_block_sptr top_block) override;
    void disconnect(gr::top_block_sptr top_block) override;
    gr::basic_block_sptr get_left_block() override { return acquisition_; }
    gr::basic_block_sptr get_right_block() override { return acquisition_; }
    //! the block itself (tests and monitors; a Channel never needs it)
    pcps_acquisition_hip_sptr block() const { return acquisition_; }
    //! the GPU this block searches on (<role>.hip_device, or its turn of <role>.hip_devices)
    int device() const { return device_; }

    void set_gnss_synchro(Gnss_Synchro* p_gnss_synchro) override
    {
        gnss_synchro_ = p_gnss_synchro;
        if (acquisition_) acquisition_->set_gnss_synchro(p_gnss_synchro);
    }
    // (an unusable adapter -- item_size() == 0: no GPU, unsupported item type -- has no block; a Channel calls these from its constructor all the same, channel.cc:53-61)
    void set_channel(unsigned int channel) override
    {
        if (acquisition_) acquisition_->set_channel(channel);
    }
    void set_channel_fsm(std::weak_ptr<ChannelFsm> channel_fsm) override
    {
        if (acquisition_) acquisition_->set_channel_fsm(std::move(channel_fsm));
    }
    void set_doppler_center(int doppler_center) override
    {
        if (acquisition_) acquisition_->set_doppler_center(doppler_center);
    }
    signed int mag() override { return acquisition_ ? static_cast<signed int>(acquisition_->mag()) : 0; }
    void reset() override
    {
        if (acquisition_) acquisition_->set_active(true);
    }
    void stop_acquisition() override
    {
        if (acquisition_) acquisition_->set_active(false);
    }
    void set_resampler_latency(uint32_t latency_samples) override
    {
        if (acquisition_) acquisition_->set_resampler_latency(latency_samples);
    }
    void set_local_code() override;
    void set_state(int state)
    {
        if (acquisition_) acquisition_->set_state(state);
    }

protected:
    Gnss_Synchro* gnss_synchro_{nullptr};

private:
    virtual void code_gen_complex_sampled(own::span<std::complex<float>> dest, uint32_t prn, int32_t sampling_freq) = 0;

    const Acq_Conf acq_parameters_;
    const std::string role_;
    const unsigned int vector_length_;
    const unsigned int code_length_;
    std::vector<std::complex<float>> code_;
    pcps_acquisition_hip_sptr acquisition_;
    int device_{0};
};

#endif  // GNSS_SDR_BASE_PCPS_ACQUISITION_HIP_H
