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


// The block's scheduler entry: the five states of galileo_e5a_noncoherent_iq_acquisition_caf_cc::general_work (e5a.cc:240-733) -- idle, restart, fill the block
// buffer, search, report -- as one small dispatcher over four steps.  What the steps consume and when they move on is the reference block's, call for call (the
// side-by-side runs in tests/host/test_adapters.cc compare consumed counts and events with it); the search itself runs on the GPU (d_core).
int galileo_e5a_noncoherentIQ_acquisition_caf_hip::general_work(int /*noutput_items*/, gr_vector_int& ninput_items, gr_vector_const_void_star& input_items,
    gr_vector_void_star& output_items)
{
    const int offered = ninput_items[0];
    const auto* in = reinterpret_cast<const gr_complex*>(input_items[0]);
    if (!d_active)  // not asked to search: the stream passes by (e5a.cc:249-254)
        {
            pass_by(offered);
            return 0;
        }
    switch (d_state)
        {
        case 0:
            begin_search();
            return 0;
        case 1:
            fill_block(in, offered);
            return 0;
        case 2:
            search_block(in);
            return 0;
        case 3:
            return report(true, offered, output_items);
        case 4:
            return report(false, offered, output_items);
        default:
            return 0;
        }
}


void galileo_e5a_noncoherentIQ_acquisition_caf_hip::pass_by(int offered)
{
    d_sample_counter += static_cast<uint64_t>(offered);
    consume_each(offered);
}


// state 0 (e5a.cc:262-274): a fresh search -- the synchro's acquisition fields cleared, the detector's dwell counters reset; nothing is consumed in this call
void galileo_e5a_noncoherentIQ_acquisition_caf_hip::begin_search()
{
    d_gnss_synchro->Acq_delay_samples = 0.0;
    d_gnss_synchro->Acq_doppler_hz = 0.0;
    d_gnss_synchro->Acq_samplestamp_samples = 0ULL;
    d_gnss_synchro->Acq_doppler_step = 0U;
    d_core.init();
    d_state = 1;
}


// state 1 (e5a.cc:276-298): samples go into the block buffer, as many as fit.  The hand-over to the search is decided on the fill level the buffer had BEFORE
// this call: once less than one scheduler buffer (d_gr_stream_buffer) was missing, the next call is the search, which tops the block up from its own input.
void galileo_e5a_noncoherentIQ_acquisition_caf_hip::fill_block(const gr_complex* in, int offered)
{
    const int taken = std::min(offered, d_fft_size - d_buffer_count);
    const bool search_next = d_buffer_count >= static_cast<int>(d_fft_size - d_gr_stream_buffer);
    std::copy_n(in, taken, d_inbuffer.begin() + d_buffer_count);
    d_buffer_count += taken;
    d_sample_counter += static_cast<uint64_t>(taken);
    consume_each(taken);
    if (search_next) d_state = 2;
}


// state 2 (e5a.cc:300-670): the block is completed from the head of the input and searched -- on the GPU.  The core answers with the next state (1: another
// dwell, 3 / 4: positive / negative) and has filled its result record wherever the reference block writes the Gnss_Synchro (e5a.cc:506-514, :634-636).
void galileo_e5a_noncoherentIQ_acquisition_caf_hip::search_block(const gr_complex* in)
{
    const int missing = d_fft_size - d_buffer_count;
    if (missing > 0) std::copy_n(in, missing, d_inbuffer.begin() + d_buffer_count);
    d_sample_counter += static_cast<uint64_t>(missing);
    const int next = d_core.work(d_sample_counter, d_inbuffer.data());
    if (next < 0)
        {
            // the engine failed (the error is in d_core.last_error()): report a negative acquisition instead of running on stale values
            std::cerr << "galileo_e5a_noncoherentIQ_acquisition_caf_hip: " << d_core.last_error() << '\n';
            d_state = 4;
        }
    else
        {
            const Hip_Detector_Result& r = d_core.result();
            d_gnss_synchro->Acq_delay_samples = r.Acq_delay_samples;
            d_gnss_synchro->Acq_doppler_hz = r.Acq_doppler_hz;
            d_gnss_synchro->Acq_samplestamp_samples = r.Acq_samplestamp_samples;
            d_gnss_synchro->Acq_doppler_step = r.Acq_doppler_step;
            d_state = next;
        }
    consume_each(missing);
    d_buffer_count = 0;
}


// states 3 and 4 (e5a.cc:672-731): the verdict goes out on "events" (1 = ACQ_SUCCESS, 2 = ACQ_FAIL), the block goes idle, the call's whole input is consumed;
// a positive verdict also leaves the synchro on the monitor output when that is enabled.  Returns the items produced.
int galileo_e5a_noncoherentIQ_acquisition_caf_hip::report(bool positive, int offered, gr_vector_void_star& output_items)
{
    d_active = false;
    d_state = 0;
    this->message_port_pub(pmt::mp("events"), pmt::from_long(positive ? 1 : 2));
    pass_by(offered);
    if (!positive || !d_enable_monitor_output) return 0;
    auto** out = reinterpret_cast<Gnss_Synchro**>(&output_items[0]);
    *out[0] = *d_gnss_synchro;
    return 1;
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
