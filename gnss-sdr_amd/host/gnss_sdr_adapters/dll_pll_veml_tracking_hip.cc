/*!
 * \file dll_pll_veml_tracking_hip.cc
 * \brief GNU Radio block with dll_pll_veml_tracking's contract over the MI355X device-closed loop; see the header.
 */
#include "dll_pll_veml_tracking_hip.h"
#include "gnss_synchro.h"
#include <gnuradio/io_signature.h>
#include <gnuradio/thread/thread.h>
#include <pmt/pmt_sugar.h>
#include <algorithm>
#include <any>
#include <cmath>
#include <cstring>
#include <iostream>
#include <utility>

#if USE_GLOG_AND_GFLAGS
#include <glog/logging.h>
#else
#include <absl/log/log.h>
#endif

dll_pll_veml_tracking_hip_sptr dll_pll_veml_make_tracking_hip(const Dll_Pll_Conf& conf_, int hip_device, int hip_periods_per_call,
    std::shared_ptr<Hip_Sample_Ring> shared_ring)
{
    return dll_pll_veml_tracking_hip_sptr(new dll_pll_veml_tracking_hip(conf_, hip_device, hip_periods_per_call, std::move(shared_ring)));
}


dll_pll_veml_tracking_hip::dll_pll_veml_tracking_hip(const Dll_Pll_Conf& conf_, int hip_device, int hip_periods_per_call,
    std::shared_ptr<Hip_Sample_Ring> shared_ring)
    : gr::block("dll_pll_veml_tracking_hip", gr::io_signature::make(1, 1, sizeof(gr_complex)), gr::io_signature::make(1, 1, sizeof(Gnss_Synchro))),
      d_trk_parameters(conf_),
      d_shared_ring(std::move(shared_ring)),
      d_device(hip_device),
      d_periods_per_call(std::max(1, hip_periods_per_call))
{
    // trk.cc:143-166
#if GNURADIO_GREATER_THAN_38
    this->set_relative_rate(1, static_cast<uint64_t>(std::max<uint32_t>(d_trk_parameters.vector_length, 1U)));
#else
    this->set_relative_rate(1.0 / static_cast<double>(std::max<uint32_t>(d_trk_parameters.vector_length, 1U)));
#endif
    this->set_max_noutput_items(d_periods_per_call);  // 1: prevent telemetry symbols accumulating in the output buffers (trk.cc:149)
    this->message_port_register_out(pmt::mp("events"));
    this->message_port_register_in(pmt::mp("telemetry_to_trk"));
    this->set_msg_handler(pmt::mp("telemetry_to_trk"), [this](auto&& PH1) { msg_handler_telemetry_to_trk(PH1); });
    this->set_tag_propagation_policy(TPP_DONT);

    std::string why;
    if (!hip_fill_trk_conf(d_trk_parameters, &d_conf, &d_signal, &why))
        {
            d_error = why;
            LOG(WARNING) << "dll_pll_veml_tracking_hip: " << why;
            return;
        }
    d_records.resize(d_periods_per_call);
    d_usable = true;
}


void dll_pll_veml_tracking_hip::forecast(int noutput_items, gr_vector_int& ninput_items_required)
{
    if (noutput_items != 0)  // trk.cc:747-754; several periods per call need their samples at once
        {
            ninput_items_required[0] = static_cast<int32_t>(d_trk_parameters.vector_length) * (1 + std::min(std::max(noutput_items, 1), d_periods_per_call));
        }
}


void dll_pll_veml_tracking_hip::msg_handler_telemetry_to_trk(const pmt::pmt_t& msg)
{
    try
        {
            if (pmt::any_ref(msg).type().hash_code() == typeid(int).hash_code())
                {
                    const int tlm_event = std::any_cast<int>(pmt::any_ref(msg));
                    if (tlm_event == 1)  // telemetry fault: force the loss-of-lock condition (trk.cc:763-768)
                        {
                            gr::thread::scoped_lock lock(d_setlock);
                            d_force_loss_of_lock = true;
                        }
                }
        }
    catch (const std::exception& ex)
        {
            LOG(WARNING) << "msg_handler_telemetry_to_trk Bad any_cast: " << ex.what();
        }
}


void dll_pll_veml_tracking_hip::set_channel(uint32_t channel)
{
    gr::thread::scoped_lock l(d_setlock);
    d_channel = channel;
}


void dll_pll_veml_tracking_hip::set_gnss_synchro(Gnss_Synchro* p_gnss_synchro)
{
    gr::thread::scoped_lock l(d_setlock);
    d_acquisition_gnss_synchro = p_gnss_synchro;
}


void dll_pll_veml_tracking_hip::start_tracking()
{
    gr::thread::scoped_lock l(d_setlock);
    if (!d_usable || d_acquisition_gnss_synchro == nullptr) return;
    // local replica(s) for the satellite acquisition has found (trk.cc:812-1030); E5a's pilot secondary code comes with the PRN
    std::string why;
    gsh_trk_conf conf = d_conf;
    if (!hip_make_tracking_codes(d_signal, &conf, d_acquisition_gnss_synchro->PRN, d_acquisition_gnss_synchro->Signal, &d_code, &d_data_code, &why))
        {
            d_error = why;
            LOG(WARNING) << "dll_pll_veml_tracking_hip: " << why;
            return;
        }
    const bool conf_changed = std::memcmp(&conf, &d_conf, sizeof(conf)) != 0;
    d_conf = conf;
    if (!d_loop || conf_changed)
        {
            d_loop = std::make_unique<Hip_Tracking_Loop>(d_device, d_conf, static_cast<int>(d_code.size()), d_shared_ring);
            if (!d_loop->ok())
                {
                    d_error = d_loop->last_error();
                    LOG(ERROR) << "dll_pll_veml_tracking_hip: " << d_error;
                    d_loop.reset();
                    return;
                }
        }
    d_force_loss_of_lock = false;
    d_state = 1;  // pull-in at the next general_work, where the read pointer is known (trk.cc:1107)
    LOG(INFO) << "Tracking of " << d_signal.system_name << " " << d_signal.signal_type << " signal started on channel " << d_channel << " for satellite PRN "
              << d_acquisition_gnss_synchro->PRN << " (MI355X loop, device " << d_device << ")";
}


void dll_pll_veml_tracking_hip::stop_tracking()
{
    gr::thread::scoped_lock l(d_setlock);
    d_state = 0;  // trk.cc:1113-1116
    if (d_loop) d_loop->stop();
}


void dll_pll_veml_tracking_hip::fill_symbol(Gnss_Synchro* out, const gsh_trk_epoch& r, bool loss_of_lock) const
{
    Gnss_Synchro current_synchro_data = *d_acquisition_gnss_synchro;  // trk.cc:2012, 2217
    if (!loss_of_lock)
        {
            if (d_signal.interchange_iq)  // trk.cc:2219-2228
                {
                    current_synchro_data.Prompt_I = static_cast<double>(r.p_data_accu[1]);
                    current_synchro_data.Prompt_Q = static_cast<double>(r.p_data_accu[0]);
                }
            else
                {
                    current_synchro_data.Prompt_I = static_cast<double>(r.p_data_accu[0]);
                    current_synchro_data.Prompt_Q = static_cast<double>(r.p_data_accu[1]);
                }
            current_synchro_data.Code_phase_samples = r.rem_code_phase_samples;
            current_synchro_data.Carrier_phase_rads = r.acc_carrier_phase_rad;
            current_synchro_data.Carrier_Doppler_hz = r.carrier_doppler_hz;
            current_synchro_data.CN0_dB_hz = static_cast<double>(r.cn0_db_hz);
            current_synchro_data.correlation_length_ms = d_signal.correlation_length_ms;
        }
    current_synchro_data.TOW_at_current_symbol_ms = 0U;  // no TOW hand-back (tow_to_trk is not carried over)
    current_synchro_data.fs = static_cast<int64_t>(d_trk_parameters.fs_in);  // trk.cc:2287-2294
    current_synchro_data.Tracking_sample_counter = r.sample_counter;      // nitems_read(0) during the call = first sample of the period
    current_synchro_data.Flag_valid_symbol_output = !loss_of_lock;
    current_synchro_data.Flag_PLL_180_deg_phase_locked = (r.symbol_flags & 2) != 0;
    *out = current_synchro_data;
}


int dll_pll_veml_tracking_hip::general_work(int noutput_items, gr_vector_int& ninput_items, gr_vector_const_void_star& input_items,
    gr_vector_void_star& output_items)
{
    gr::thread::scoped_lock l(d_setlock);
    const auto* in = reinterpret_cast<const gr_complex*>(input_items[0]);
    auto* out = reinterpret_cast<Gnss_Synchro*>(output_items[0]);
    const uint64_t read_pos = this->nitems_read(0);

    switch (d_state)
        {
        case 0:  // standby: consume at full throttle (trk.cc:1941-1947)
            consume_each(ninput_items[0]);
            return 0;
        case 1:  // pull-in: skip samples until the incoming signal is aligned with the local replica (trk.cc:1949-1978)
            {
                int32_t samples_offset = 0;
                if (!d_loop || !d_loop->start(d_code.data(), d_conf.track_pilot ? d_data_code.data() : nullptr, static_cast<int>(d_code.size()), read_pos,
                                  d_acquisition_gnss_synchro->Acq_delay_samples, d_acquisition_gnss_synchro->Acq_doppler_hz,
                                  d_acquisition_gnss_synchro->Acq_samplestamp_samples, &samples_offset))
                    {
                        d_error = d_loop ? d_loop->last_error() : std::string("no device loop");
                        LOG(ERROR) << "dll_pll_veml_tracking_hip: " << d_error;
                        d_state = 0;
                        this->message_port_pub(pmt::mp("events"), pmt::from_long(3));  // the channel goes back to acquisition
                        consume_each(ninput_items[0]);
                        return 0;
                    }
                d_state = 2;
                consume_each(samples_offset);
                return 0;
            }
        default:
            break;
        }

    // ---- tracking: one code period per output item (states 2 / 3 / 4 live on the device)
    if (d_force_loss_of_lock)  // the reference raises the fail counter and the next lock test drops the channel (trk.cc:767, 1208-1221)
        {
            d_force_loss_of_lock = false;
            gsh_trk_epoch r{};
            r.sample_counter = read_pos;
            d_loop->stop();
            d_state = 0;
            std::cout << "Loss of lock in channel " << d_channel << " (telemetry fault)!\n";
            this->message_port_pub(pmt::mp("events"), pmt::from_long(3));
            fill_symbol(&out[0], r, true);
            consume_each(static_cast<int>(d_trk_parameters.vector_length));
            return 1;
        }
    if (!d_loop->push(in, read_pos, static_cast<uint64_t>(ninput_items[0])))
        {
            d_error = d_loop->last_error();
            LOG(ERROR) << "dll_pll_veml_tracking_hip: " << d_error;
            d_state = 0;
            this->message_port_pub(pmt::mp("events"), pmt::from_long(3));
            consume_each(ninput_items[0]);
            return 0;
        }
    const int want = std::min(std::max(noutput_items, 1), d_periods_per_call);
    const int done = d_loop->run(want, d_records.data());
    if (done < 0)
        {
            d_error = d_loop->last_error();
            LOG(ERROR) << "dll_pll_veml_tracking_hip: " << d_error;
            d_state = 0;
            this->message_port_pub(pmt::mp("events"), pmt::from_long(3));
            consume_each(ninput_items[0]);
            return 0;
        }
    int produced = 0;
    int64_t consumed = 0;
    for (int e = 0; e < done; e++)
        {
            const gsh_trk_epoch& r = d_records[e];
            d_last = r;
            if (r.flags & 2)  // loss of lock declared by the device's lock detectors: "events" 3 and an invalid symbol (trk.cc:1208-1221, 2009-2014)
                {
                    std::cout << "Loss of lock in channel " << d_channel << "!\n";
                    this->message_port_pub(pmt::mp("events"), pmt::from_long(3));
                    d_state = 0;
                    fill_symbol(&out[produced++], r, true);
                    // clear_tracking_vars leaves d_current_prn_length_samples at its last value; the block consumes it (trk.cc:2283)
                    consumed += static_cast<int64_t>(d_trk_parameters.vector_length);
                    break;
                }
            consumed += r.prn_length_samples;
            if (r.symbol_flags & 1) fill_symbol(&out[produced++], r, false);
        }
    consume_each(static_cast<int>(consumed));
    return produced;
}
