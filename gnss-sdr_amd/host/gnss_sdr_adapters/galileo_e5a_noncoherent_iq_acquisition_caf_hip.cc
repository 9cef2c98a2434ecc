/*!
 * \file galileo_e5a_noncoherent_iq_acquisition_caf_hip.cc
 * \brief See the header.  BUILT ONLY INSIDE A gnss-sdr TREE.
 */
#include "galileo_e5a_noncoherent_iq_acquisition_caf_hip.h"
#include "Galileo_E5a.h"
#include "configuration_interface.h"
#include "galileo_e5_signal_replica.h"
#include <gnuradio/io_signature.h>
#include <pmt/pmt.h>
#include <algorithm>
#include <array>
#include <cmath>
#include <iostream>

namespace
{
Hip_Acq_Conf to_hip_conf(const Acq_Conf& a)
{
    Hip_Acq_Conf h;
    h.fs_in = a.fs_in;
    h.resampled_fs = a.resampled_fs;
    h.samples_per_ms = a.samples_per_ms;
    h.threshold = a.threshold;
    h.pfa = a.pfa;
    h.samples_per_code = a.samples_per_code;
    h.sampled_ms = a.sampled_ms;
    h.ms_per_code = a.ms_per_code;
    h.samples_per_chip = a.samples_per_chip;
    h.chips_per_second = a.chips_per_second;
    h.max_dwells = a.max_dwells;
    h.doppler_max = a.doppler_max;
    h.doppler_step = a.doppler_step;
    h.bit_transition_flag = a.bit_transition_flag;
    h.dump = a.dump;
    h.dump_filename = a.dump_filename;
    h.dump_channel = a.dump_channel;
    return h;
}
}  // namespace


// ------------------------------------------------------------------------------------------------------------------ the block
galileo_e5a_noncoherentIQ_acquisition_caf_hip_sptr galileo_e5a_noncoherentIQ_make_acquisition_caf_hip(const Hip_Acq_Conf& conf, bool enable_monitor_output,
    bool both_signal_components_, int CAF_window_hz_, int Zero_padding_, int device)
{
    return galileo_e5a_noncoherentIQ_acquisition_caf_hip_sptr(
        new galileo_e5a_noncoherentIQ_acquisition_caf_hip(conf, enable_monitor_output, both_signal_components_, CAF_window_hz_, Zero_padding_, device));
}


galileo_e5a_noncoherentIQ_acquisition_caf_hip::galileo_e5a_noncoherentIQ_acquisition_caf_hip(const Hip_Acq_Conf& conf, bool enable_monitor_output,
    bool both_signal_components_, int CAF_window_hz_, int Zero_padding_, int device)
    : acquisition_impl_interface("galileo_e5a_noncoherentIQ_acquisition_caf_hip", gr::io_signature::make(1, 1, sizeof(gr_complex)),
          gr::io_signature::make(0, 1, sizeof(Gnss_Synchro))),
      d_core(conf, both_signal_components_, CAF_window_hz_, Zero_padding_, device),
      d_fft_size(static_cast<int>(conf.sampled_ms) * static_cast<int>(conf.samples_per_ms)),
      d_enable_monitor_output(enable_monitor_output)
{
    this->message_port_register_out(pmt::mp("events"));
    d_inbuffer.resize(static_cast<size_t>(d_fft_size));
}


void galileo_e5a_noncoherentIQ_acquisition_caf_hip::set_local_code(std::complex<float>* codeI, std::complex<float>* codeQ)
{
    d_core.set_local_code(codeI, codeQ);
}


int galileo_e5a_noncoherentIQ_acquisition_caf_hip::general_work(int /*noutput_items*/, gr_vector_int& ninput_items, gr_vector_const_void_star& input_items,
    gr_vector_void_star& output_items)
{
    int acquisition_message = -1;  // 1 = ACQ_SUCCESS, 2 = ACQ_FAIL
    int return_value = 0;          // number of Gnss_Synchro objects produced

    if (!d_active)  // e5a.cc:249-254
        {
            d_sample_counter += static_cast<uint64_t>(ninput_items[0]);
            consume_each(ninput_items[0]);
            return 0;
        }

    switch (d_state)
        {
        case 0:  // restart (e5a.cc:262-274)
            {
                d_gnss_synchro->Acq_delay_samples = 0.0;
                d_gnss_synchro->Acq_doppler_hz = 0.0;
                d_gnss_synchro->Acq_samplestamp_samples = 0ULL;
                d_gnss_synchro->Acq_doppler_step = 0U;
                d_core.init();
                d_state = 1;
                break;
            }
        case 1:  // load the buffer until it holds a block (e5a.cc:276-298)
            {
                const auto* in = reinterpret_cast<const gr_complex*>(input_items[0]);
                int buff_increment;
                if ((ninput_items[0] + d_buffer_count) <= d_fft_size)
                    {
                        buff_increment = ninput_items[0];
                    }
                else
                    {
                        buff_increment = d_fft_size - d_buffer_count;
                    }
                std::copy(in, in + buff_increment, d_inbuffer.begin() + d_buffer_count);
                // if the buffer will be full in the next iteration
                if (d_buffer_count >= static_cast<int>(d_fft_size - d_gr_stream_buffer))
                    {
                        d_state = 2;
                    }
                d_buffer_count += buff_increment;
                d_sample_counter += static_cast<uint64_t>(buff_increment);
                consume_each(buff_increment);
                break;
            }
        case 2:  // the search (e5a.cc:300-670), on the GPU
            {
                const auto* in = reinterpret_cast<const gr_complex*>(input_items[0]);
                if (d_buffer_count < d_fft_size)
                    {
                        std::copy(in, in + (d_fft_size - d_buffer_count), d_inbuffer.begin() + d_buffer_count);
                    }
                d_sample_counter += static_cast<uint64_t>(d_fft_size - d_buffer_count);

                const int next = d_core.work(d_sample_counter, d_inbuffer.data());
                if (next < 0)
                    {
                        // the engine failed (the error is in d_core.last_error()): report a negative acquisition instead of running on stale values
                        std::cerr << "galileo_e5a_noncoherentIQ_acquisition_caf_hip: " << d_core.last_error() << '\n';
                        d_state = 4;
                    }
                else
                    {
                        // the core fills its own result record whenever the block would have written the Gnss_Synchro (e5a.cc:506-514, :634-636)
                        const Hip_Detector_Result& r = d_core.result();
                        d_gnss_synchro->Acq_delay_samples = r.Acq_delay_samples;
                        d_gnss_synchro->Acq_doppler_hz = r.Acq_doppler_hz;
                        d_gnss_synchro->Acq_samplestamp_samples = r.Acq_samplestamp_samples;
                        d_gnss_synchro->Acq_doppler_step = r.Acq_doppler_step;
                        d_state = next;
                    }
                consume_each(d_fft_size - d_buffer_count);
                d_buffer_count = 0;
                break;
            }
        case 3:  // positive acquisition (e5a.cc:672-707)
            {
                d_active = false;
                d_state = 0;
                acquisition_message = 1;
                this->message_port_pub(pmt::mp("events"), pmt::from_long(acquisition_message));
                d_sample_counter += static_cast<uint64_t>(ninput_items[0]);
                consume_each(ninput_items[0]);
                if (d_enable_monitor_output)
                    {
                        auto** out = reinterpret_cast<Gnss_Synchro**>(&output_items[0]);
                        Gnss_Synchro current_synchro_data = Gnss_Synchro();
                        current_synchro_data = *d_gnss_synchro;
                        *out[0] = std::move(current_synchro_data);
                        return_value = 1;
                    }
                break;
            }
        case 4:  // negative acquisition (e5a.cc:709-731)
            {
                d_active = false;
                d_state = 0;
                d_sample_counter += static_cast<uint64_t>(ninput_items[0]);
                consume_each(ninput_items[0]);
                acquisition_message = 2;
                this->message_port_pub(pmt::mp("events"), pmt::from_long(acquisition_message));
                break;
            }
        }
    return return_value;
}


// ------------------------------------------------------------------------------------------------------------------ the adapter
Acq_Conf hip_e5a_caf_acq_conf(const ConfigurationInterface* configuration, const std::string& role)
{
    // adapters/base_pcps_acquisition_custom.cc:35-86 with the arguments galileo_e5a_noncoherent_iq_acquisition_caf.cc:73-83 passes
    // (the command-line flag overrides of :53-72 belong to the receiver's main program)
    const uint32_t ms_per_code = GALILEO_E5A_CODE_PERIOD_MS;
    const double chip_rate = GALILEO_E5A_CODE_CHIP_RATE_CPS;
    const double code_length_chips = GALILEO_E5A_CODE_LENGTH_CHIPS;
    Acq_Conf acq_parameters;
    acq_parameters.ms_per_code = ms_per_code;
    acq_parameters.sampled_ms = ms_per_code;
    acq_parameters.dump_filename = "./acquisition.dat";
    acq_parameters.SetFromConfiguration(configuration, role, chip_rate, 0);
    const uint32_t max_sampled_ms = configuration->property(role + ".Zero_padding", 0) > 0 ? 2U : 3U;  // :56-66
    if (acq_parameters.sampled_ms > max_sampled_ms)
        {
            acq_parameters.sampled_ms = max_sampled_ms;
            std::cout << "Too high coherent integration time. Changing to " << max_sampled_ms << "ms\n";
        }
    acq_parameters.num_codes = acq_parameters.sampled_ms / ms_per_code;
    acq_parameters.code_length = static_cast<unsigned int>(round(acq_parameters.fs_in / (chip_rate / code_length_chips)));
    acq_parameters.vector_length = acq_parameters.code_length * acq_parameters.num_codes;
    if (acq_parameters.pfa != 0)  // ThresholdComputeDoppler::calculate_threshold, :89-112
        {
            acq_parameters.threshold = hip_threshold_compute_doppler(acq_parameters.pfa, acq_parameters.vector_length, acq_parameters.doppler_max, acq_parameters.doppler_step);
        }
    return acq_parameters;
}


GalileoE5aNoncoherentIQAcquisitionCafHip::GalileoE5aNoncoherentIQAcquisitionCafHip(const ConfigurationInterface* configuration, const std::string& role,
    unsigned int /*in_streams*/, unsigned int /*out_streams*/)
    : acq_parameters_(hip_e5a_caf_acq_conf(configuration, role)),
      codeI_(acq_parameters_.vector_length),
      codeQ_(acq_parameters_.vector_length),
      role_(role),
      zero_padding_(configuration->property(role + ".Zero_padding", 0)),
      caf_window_hz_(configuration->property(role + ".CAF_window_hz", 0))
{
    if (acq_parameters_.item_type == "gr_complex")
        {
            const auto sig = configuration->property("Channel.signal", std::string("5X"));
            both_signal_components_ = (sig.at(0) == '5' && sig.at(1) == 'X');  // galileo_e5a_noncoherent_iq_acquisition_caf.cc:86-87
            const int device = configuration->property(role + ".hip_device", 0);
            acquisition_cc_ = galileo_e5a_noncoherentIQ_make_acquisition_caf_hip(to_hip_conf(acq_parameters_), acq_parameters_.enable_monitor_output,
                both_signal_components_, caf_window_hz_, zero_padding_, device);
            if (!acquisition_cc_->ok()) acquisition_cc_.reset();  // item_size() == 0: the factory rejects the block instead of running without a GPU
        }
}


void GalileoE5aNoncoherentIQAcquisitionCafHip::connect(gr::top_block_sptr /*top_block*/) {}
void GalileoE5aNoncoherentIQAcquisitionCafHip::disconnect(gr::top_block_sptr /*top_block*/) {}


void GalileoE5aNoncoherentIQAcquisitionCafHip::set_gnss_synchro(Gnss_Synchro* p_gnss_synchro)
{
    gnss_synchro_ = p_gnss_synchro;
    if (acquisition_cc_) acquisition_cc_->set_gnss_synchro(p_gnss_synchro);
}


void GalileoE5aNoncoherentIQAcquisitionCafHip::set_channel(unsigned int channel)
{
    channel_ = channel;
    if (acquisition_cc_) acquisition_cc_->set_channel(channel);
}


void GalileoE5aNoncoherentIQAcquisitionCafHip::set_channel_fsm(std::weak_ptr<ChannelFsm> channel_fsm)
{
    if (acquisition_cc_) acquisition_cc_->set_channel_fsm(std::move(channel_fsm));
}


signed int GalileoE5aNoncoherentIQAcquisitionCafHip::mag() { return acquisition_cc_ ? static_cast<signed int>(acquisition_cc_->mag()) : 0; }


void GalileoE5aNoncoherentIQAcquisitionCafHip::reset()
{
    if (acquisition_cc_) acquisition_cc_->set_active(true);
}


void GalileoE5aNoncoherentIQAcquisitionCafHip::stop_acquisition()
{
    if (acquisition_cc_) acquisition_cc_->set_active(false);
}


void GalileoE5aNoncoherentIQAcquisitionCafHip::set_local_code()
{
    if (!acquisition_cc_) return;
    // galileo_e5a_noncoherent_iq_acquisition_caf.cc:94-141
    const auto code_length = acq_parameters_.code_length;
    std::vector<std::complex<float>> codeI(code_length);
    std::vector<std::complex<float>> codeQ(code_length);
    const bool both = gnss_synchro_->Signal[0] == '5' && gnss_synchro_->Signal[1] == 'X';
    if (both)
        {
            std::array<char, 3> a = {{'5', 'I', '\0'}};
            galileo_e5_a_code_gen_complex_sampled(codeI, gnss_synchro_->PRN, a, acq_parameters_.fs_in, 0);
            std::array<char, 3> b = {{'5', 'Q', '\0'}};
            galileo_e5_a_code_gen_complex_sampled(codeQ, gnss_synchro_->PRN, b, acq_parameters_.fs_in, 0);
        }
    else
        {
            std::array<char, 3> signal_type_ = {{'5', 'X', '\0'}};
            galileo_e5_a_code_gen_complex_sampled(codeI, gnss_synchro_->PRN, signal_type_, acq_parameters_.fs_in, 0);
        }
    // sampled_ms code periods (the secondary sequence (1,1,1); the block forms the other combination itself), or one period + zeros
    std::fill(codeI_.begin(), codeI_.end(), std::complex<float>(0.0F, 0.0F));
    std::fill(codeQ_.begin(), codeQ_.end(), std::complex<float>(0.0F, 0.0F));
    const unsigned int periods = zero_padding_ == 0 ? acq_parameters_.sampled_ms : 1U;
    for (unsigned int i = 0; i < periods; i++)
        {
            std::copy_n(codeI.data(), code_length, codeI_.data() + static_cast<size_t>(i) * code_length);
            if (both) std::copy_n(codeQ.data(), code_length, codeQ_.data() + static_cast<size_t>(i) * code_length);
        }
    acquisition_cc_->set_local_code(codeI_.data(), codeQ_.data());
}
