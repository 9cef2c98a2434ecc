/*!
 * \file galileo_e5a_noncoherent_iq_acquisition_caf_hip.h
 * \brief MI355X counterparts of gnss-sdr's Galileo E5a non-coherent I + Q acquisition:
 *
 *   galileo_e5a_noncoherentIQ_acquisition_caf_hip   the GNU Radio block -- gnuradio_blocks/galileo_e5a_noncoherent_iq_acquisition_caf_cc.{h,cc}
 *   GalileoE5aNoncoherentIQAcquisitionCafHip        the adapter        -- adapters/galileo_e5a_noncoherent_iq_acquisition_caf.{h,cc}
 *                                                                         over adapters/base_pcps_acquisition_custom.{h,cc}
 *
 * BUILT ONLY INSIDE A gnss-sdr TREE (GNU Radio, pmt, Gnss_Synchro, ChannelFsm, the reference's Acq_Conf and signal generators).
 * The block keeps the reference block's stream contract and states: 1 input of gr_complex SAMPLES (it buffers a block itself, states 1 / 2,
 * e5a.cc:276-315), 0..1 output of Gnss_Synchro, message port "events" carrying 1 / 2; state 0 restarts, 3 / 4 report.  The search of
 * state 2 is Hip_Galileo_E5a_Noncoherent_Iq_Core (host/hip_pcps_detectors.{h,cc}) on the GPU.  Implementation name
 * "Galileo_E5a_Noncoherent_IQ_Acquisition_CAF_HIP"; same configuration keys as the reference adapter (Zero_padding, CAF_window_hz, Channel.signal,
 * coherent_integration_time_ms capped at 3, or 2 with zero padding) plus hip_device.
 */
#ifndef GNSS_SDR_GALILEO_E5A_NONCOHERENT_IQ_ACQUISITION_CAF_HIP_H
#define GNSS_SDR_GALILEO_E5A_NONCOHERENT_IQ_ACQUISITION_CAF_HIP_H

#include "acq_conf.h"
#include "acquisition_impl_interface.h"
#include "acquisition_interface.h"
#include "channel_fsm.h"
#include "gnss_synchro.h"
#include "hip_pcps_detectors.h"
#include <gnuradio/block.h>
#include <complex>
#include <memory>
#include <string>
#include <vector>

class ConfigurationInterface;
class galileo_e5a_noncoherentIQ_acquisition_caf_hip;
using galileo_e5a_noncoherentIQ_acquisition_caf_hip_sptr = gnss_shared_ptr<galileo_e5a_noncoherentIQ_acquisition_caf_hip>;

galileo_e5a_noncoherentIQ_acquisition_caf_hip_sptr galileo_e5a_noncoherentIQ_make_acquisition_caf_hip(const Hip_Acq_Conf& conf, bool enable_monitor_output,
    bool both_signal_components_, int CAF_window_hz_, int Zero_padding_, int device);

class galileo_e5a_noncoherentIQ_acquisition_caf_hip : public acquisition_impl_interface
{
public:
    ~galileo_e5a_noncoherentIQ_acquisition_caf_hip() override = default;

    void set_gnss_synchro(Gnss_Synchro* p_gnss_synchro) override { d_gnss_synchro = p_gnss_synchro; }  // e5a.h:84-87
    void set_channel(uint32_t channel_id) override { d_channel = channel_id; }                           // e5a.h:122-125
    void set_channel_fsm(std::weak_ptr<ChannelFsm> channel_fsm) override { d_channel_fsm = std::move(channel_fsm); }
    void set_local_code(std::complex<float>* code) override { set_local_code(code, code); }
    void set_local_code(std::complex<float>* codeI, std::complex<float>* codeQ) override;               // e5a.cc:162-222
    uint32_t mag() const override { return static_cast<uint32_t>(d_core.mag()); }                       // e5a.h:92-95
    void set_active(bool active) override { d_active = active; }                                          // e5a.h:109-112
    bool ok() const { return d_core.ok(); }

    int general_work(int noutput_items, gr_vector_int& ninput_items, gr_vector_const_void_star& input_items, gr_vector_void_star& output_items) override;

private:
    friend galileo_e5a_noncoherentIQ_acquisition_caf_hip_sptr galileo_e5a_noncoherentIQ_make_acquisition_caf_hip(const Hip_Acq_Conf& conf, bool enable_monitor_output,
        bool both_signal_components_, int CAF_window_hz_, int Zero_padding_, int device);
    galileo_e5a_noncoherentIQ_acquisition_caf_hip(const Hip_Acq_Conf& conf, bool enable_monitor_output, bool both_signal_components_, int CAF_window_hz_,
        int Zero_padding_, int device);

    // the steps of general_work (states of e5a.cc:240-733)
    void pass_by(int offered);
    void begin_search();
    void fill_block(const gr_complex* in, int offered);
    void search_block(const gr_complex* in);
    int report(bool positive, int offered, gr_vector_void_star& output_items);

    Hip_Galileo_E5a_Noncoherent_Iq_Core d_core;
    std::vector<std::complex<float>> d_inbuffer;
    std::weak_ptr<ChannelFsm> d_channel_fsm;
    Gnss_Synchro* d_gnss_synchro{nullptr};
    uint64_t d_sample_counter{0ULL};
    int d_state{0};
    int d_buffer_count{0};
    int d_fft_size{0};
    int d_gr_stream_buffer{0};
    uint32_t d_channel{0};
    bool d_active{false};
    bool d_enable_monitor_output{false};
};


/*! ThresholdComputeDoppler / get_acq_conf of adapters/base_pcps_acquisition_custom.cc:31-112 for this adapter */
Acq_Conf hip_e5a_caf_acq_conf(const ConfigurationInterface* configuration, const std::string& role);

class GalileoE5aNoncoherentIQAcquisitionCafHip : public AcquisitionInterface
{
public:
    GalileoE5aNoncoherentIQAcquisitionCafHip(const ConfigurationInterface* configuration, const std::string& role, unsigned int in_streams, unsigned int out_streams);
    ~GalileoE5aNoncoherentIQAcquisitionCafHip() override = default;

    inline std::string role() override { return role_; }
    inline std::string implementation() override { return "Galileo_E5a_Noncoherent_IQ_Acquisition_CAF_HIP"; }
    /*! 0 when the engine could not be created (no GPU, unsupported length) or for item types the block does not take: the factory rejects the block */
    inline size_t item_size() override { return acquisition_cc_ ? sizeof(gr_complex) : 0; }

    void connect(gr::top_block_sptr top_block) override;
    void disconnect(gr::top_block_sptr top_block) override;
    gr::basic_block_sptr get_left_block() override { return acquisition_cc_; }
    gr::basic_block_sptr get_right_block() override { return acquisition_cc_; }

    // adapters/base_pcps_acquisition_custom.cc:205-260
    void set_gnss_synchro(Gnss_Synchro* p_gnss_synchro) override;
    void set_channel(unsigned int channel) override;
    void set_channel_fsm(std::weak_ptr<ChannelFsm> channel_fsm) override;
    void set_local_code() override;  //!< adapters/galileo_e5a_noncoherent_iq_acquisition_caf.cc:94-141
    signed int mag() override;
    void reset() override;
    void stop_acquisition() override;
    void set_resampler_latency(uint32_t /*latency_samples*/) override {}

    const Acq_Conf& acq_parameters() const { return acq_parameters_; }

private:
    const Acq_Conf acq_parameters_;
    galileo_e5a_noncoherentIQ_acquisition_caf_hip_sptr acquisition_cc_;
    Gnss_Synchro* gnss_synchro_{nullptr};
    std::vector<std::complex<float>> codeI_, codeQ_;
    const std::string role_;
    unsigned int channel_{0};
    const int zero_padding_;
    const int caf_window_hz_;
    bool both_signal_components_{false};
};

#endif  // GNSS_SDR_GALILEO_E5A_NONCOHERENT_IQ_ACQUISITION_CAF_HIP_H
