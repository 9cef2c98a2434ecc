/*!
 * \file dll_pll_veml_tracking_hip.cc
 * \brief GNU Radio block with dll_pll_veml_tracking's contract over the MI355X device-closed loop; see the header.
 */
#include "dll_pll_veml_tracking_hip.h"
#include "hip_mat5_writer.h"
#include <iostream>
#include "gnss_synchro.h"
#include <gnuradio/io_signature.h>
#include <gnuradio/thread/thread.h>
#include <pmt/pmt_sugar.h>
#include <algorithm>
#include <any>
#include <cmath>
#include <cstring>
#include <filesystem>
#include <iostream>
#include <utility>

#if USE_GLOG_AND_GFLAGS
#include <glog/logging.h>
#else
#include <absl/log/log.h>
#endif

dll_pll_veml_tracking_hip_sptr dll_pll_veml_make_tracking_hip(const Dll_Pll_Conf& conf_, int hip_periods_per_call, std::shared_ptr<Hip_Tracking_Runtime> runtime)
{
    return dll_pll_veml_tracking_hip_sptr(new dll_pll_veml_tracking_hip(conf_, hip_periods_per_call, std::move(runtime)));
}


dll_pll_veml_tracking_hip::dll_pll_veml_tracking_hip(const Dll_Pll_Conf& conf_, int hip_periods_per_call, std::shared_ptr<Hip_Tracking_Runtime> runtime)
    : gr::block("dll_pll_veml_tracking_hip", gr::io_signature::make(1, 1, sizeof(gr_complex)), gr::io_signature::make(1, 1, sizeof(Gnss_Synchro))),
      d_trk_parameters(conf_),
      d_runtime(std::move(runtime)),
      d_periods_per_call(std::min(std::max(1, hip_periods_per_call), 64)),
      d_last_tow_received(std::make_shared<TOW_to_trk>()),
      d_signal_type(conf_.signal)
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
    this->set_tag_propagation_policy(TPP_DONT);  // the time tag is adjusted and regenerated in general_work (trk.cc:742)

    std::string why;
    if (!hip_fill_trk_conf(d_trk_parameters, &d_conf, &d_signal, &why))
        {
            d_error = why;
            LOG(WARNING) << "dll_pll_veml_tracking_hip: " << why;
            return;
        }
    if (!d_runtime || !d_runtime->ok())
        {
            d_error = d_runtime ? d_runtime->last_error(-1) : std::string("no tracking runtime");
            LOG(ERROR) << "dll_pll_veml_tracking_hip: " << d_error;
            return;
        }
    d_records.resize(d_periods_per_call);

    // dump file name, trk.cc:707-735: <dir>/<base name without extension>, the channel number and ".dat" are added in set_channel
    d_dump = d_trk_parameters.dump;
    if (d_dump)
        {
            d_dump_filename = d_trk_parameters.dump_filename;
            std::string dump_path;
            if (d_dump_filename.find_last_of('/') != std::string::npos)
                {
                    const std::string base = d_dump_filename.substr(d_dump_filename.find_last_of('/') + 1);
                    dump_path = d_dump_filename.substr(0, d_dump_filename.find_last_of('/'));
                    d_dump_filename = base;
                }
            else
                {
                    dump_path = std::string(".");
                }
            if (d_dump_filename.empty()) d_dump_filename = "trk_channel_";
            if (d_dump_filename.substr(1).find_last_of('.') != std::string::npos) d_dump_filename = d_dump_filename.substr(0, d_dump_filename.find_last_of('.'));
            d_dump_filename = dump_path + static_cast<char>(std::filesystem::path::preferred_separator) + d_dump_filename;
            std::error_code ec;
            std::filesystem::create_directories(dump_path, ec);
            if (ec)
                {
                    std::cerr << "GNSS-SDR cannot create dump files for the tracking block. Wrong permissions?\n";
                    d_dump = false;
                }
        }
    d_usable = true;
}


dll_pll_veml_tracking_hip::~dll_pll_veml_tracking_hip()
{
    try
        {
            flush_dump();
            // trk.cc:175-193 (destructor): with dump_mat the binary dump is turned into <dump_filename><channel>.mat (save_matfile, :1706-1890)
            if (d_dump && d_trk_parameters.dump_mat && !d_dump_path.empty())
                {
                    const long epochs = hip_tracking_dump_to_mat(d_dump_path);
                    if (epochs < 0) std::cerr << "Problem generating the .mat file of " << d_dump_path << '\n';
                }
            if (d_runtime && d_slot >= 0) d_runtime->detach(d_slot);
        }
    catch (...)
        {
        }
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
            if (d_trk_parameters.tow_to_trk && pmt::any_ref(msg).type().hash_code() == typeid(const std::shared_ptr<TOW_to_trk>).hash_code())  // trk.cc:771-779
                {
                    const auto tow_event = std::any_cast<const std::shared_ptr<TOW_to_trk>>(pmt::any_ref(msg));
                    gr::thread::scoped_lock lock(d_setlock);
                    if (d_acquisition_gnss_synchro != nullptr && tow_event->signal == d_signal_type && tow_event->channel == static_cast<int32_t>(d_channel) &&
                        tow_event->prn == d_acquisition_gnss_synchro->PRN)
                        {
                            d_last_tow_received = tow_event;
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
    if (d_dump && d_dump_path.empty())  // trk.cc:1851-1873: <name><channel>.dat, created here
        {
            d_dump_path = d_dump_filename + std::to_string(d_channel) + ".dat";
            if (gsh_trk_write_dump(d_dump_path.c_str(), 0, &d_conf, 0, nullptr, 0, nullptr, nullptr) != GSH_OK)
                {
                    LOG(WARNING) << "channel " << d_channel << " Exception opening trk dump file " << gsh_last_error();
                    d_dump_path.clear();
                }
            else
                {
                    LOG(INFO) << "Tracking dump enabled on channel " << d_channel << " Log file: " << d_dump_path;
                }
        }
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
    if (d_slot < 0 || conf_changed)
        {
            // the slot belongs to the group of channels with exactly this loop configuration (signals whose configuration depends on the
            // satellite -- per-PRN secondary codes, BeiDou GEO / MEO, the GLONASS frequency channel -- change group with the satellite)
            if (d_slot >= 0) d_runtime->detach(d_slot);
            d_slot = d_runtime->attach(d_conf, static_cast<int>(d_code.size()));
            if (d_slot < 0)
                {
                    d_error = d_runtime->last_error(-1);
                    LOG(ERROR) << "dll_pll_veml_tracking_hip: " << d_error;
                    return;
                }
        }
    else
        {
            d_runtime->stop(d_slot);  // a channel that is started again while it still runs (trk.cc:793-1110 re-initialises everything)
        }
    d_force_loss_of_lock = false;
    d_timetag_waiting = false;
    d_last_timetag_samplecounter = 0;
    d_state = 1;  // pull-in at the next general_work, where the read pointer is known (trk.cc:1107)
    LOG(INFO) << "Tracking of " << d_signal.system_name << " " << d_signal.signal_type << " signal started on channel " << d_channel << " for satellite PRN "
              << d_acquisition_gnss_synchro->PRN << " (MI355X loop, device " << d_runtime->device() << ")";
}


void dll_pll_veml_tracking_hip::stop_tracking()
{
    gr::thread::scoped_lock l(d_setlock);
    d_state = 0;  // trk.cc:1113-1116
    if (d_runtime && d_slot >= 0) d_runtime->stop(d_slot);
}


// trk.cc:1921-1935: TOW of the period that starts at `period_start`, from the last telemetry hand-back
void dll_pll_veml_tracking_hip::estimate_tow(uint64_t period_start, int32_t prn_length_before, uint64_t* tow_ms, uint32_t* wn) const
{
    *tow_ms = 0ULL;
    *wn = 0U;
    if (d_trk_parameters.tow_to_trk && (d_last_tow_received->prn == d_acquisition_gnss_synchro->PRN))
        {
            const double time_diff_s = (static_cast<double>(period_start) + prn_length_before - static_cast<double>(d_last_tow_received->sample_stamp)) / d_trk_parameters.fs_in;
            const auto time_diff_ms = static_cast<uint64_t>((time_diff_s * 1000.0));
            *tow_ms = (d_last_tow_received->tow + time_diff_ms) % static_cast<uint64_t>(604800000);
            *wn = (*tow_ms < d_last_tow_received->tow) ? d_last_tow_received->wn + 1 : d_last_tow_received->wn;
        }
}


// trk.cc:2256-2283: remember the newest GnssTime tag of the period's samples
void dll_pll_veml_tracking_hip::collect_time_tags(uint64_t from, uint64_t to)
{
    std::vector<gr::tag_t> tags_vec;
    this->get_tags_in_range(tags_vec, 0, from, to);
    for (const auto& it : tags_vec)
        {
            try
                {
                    if (pmt::any_ref(it.value).type().hash_code() == typeid(const std::shared_ptr<GnssTime>).hash_code())
                        {
                            const auto last_timetag = std::any_cast<const std::shared_ptr<GnssTime>>(pmt::any_ref(it.value));
                            d_last_timetag = *last_timetag;
                            d_last_timetag_samplecounter = it.offset;
                            d_timetag_waiting = true;
                        }
                }
            catch (const std::exception& ex)
                {
                    LOG(WARNING) << "Bad any_cast: " << ex.what();
                }
        }
}


// trk.cc:2295-2324: the tags that travel with an output item
void dll_pll_veml_tracking_hip::emit_tags(uint64_t out_item, uint64_t tracking_sample_counter, uint64_t period_start)
{
    if (d_timetag_waiting)
        {
            const int64_t diff_samplecount = tracking_sample_counter >= d_last_timetag_samplecounter
                                                 ? static_cast<int64_t>(tracking_sample_counter - d_last_timetag_samplecounter)
                                                 : -static_cast<int64_t>(d_last_timetag_samplecounter - tracking_sample_counter);
            double intpart;
            d_last_timetag.tow_ms_fraction = d_last_timetag.tow_ms_fraction + std::modf(1000.0 * static_cast<double>(diff_samplecount) / d_trk_parameters.fs_in, &intpart);
            const std::shared_ptr<GnssTime> tmp_obj = std::make_shared<GnssTime>(GnssTime());
            tmp_obj->week = d_last_timetag.week;
            tmp_obj->tow_ms = d_last_timetag.tow_ms + static_cast<int>(intpart);
            tmp_obj->tow_ms_fraction = d_last_timetag.tow_ms_fraction;
            tmp_obj->rx_time = static_cast<double>(tracking_sample_counter) / d_trk_parameters.fs_in;
            add_item_tag(0, out_item + 1, pmt::mp("timetag"), pmt::make_any(tmp_obj));
            d_timetag_waiting = false;
        }
    std::vector<gr::tag_t> tags{};
    const uint64_t len = static_cast<uint64_t>(std::max(d_current_prn_length_samples, 0));
    get_tags_in_range(tags, 0, period_start >= len ? period_start - len : 0, period_start, pmt::mp("sensor_data"));
    for (const auto& tag : tags) add_item_tag(0, out_item + 1, tag.key, tag.value);
}


void dll_pll_veml_tracking_hip::dump_record(const gsh_trk_epoch& r, uint64_t tow_ms, uint32_t wn)
{
    if (!d_dump || d_dump_path.empty()) return;
    gsh_trk_epoch q = r;
    if (r.state == 3)
        {
            // coherent integration runs no loop update: log_data prints the members run_dll_pll left in the last period that ran it
            q.carr_phase_error_hz = d_loop_fields.carr_phase_error_hz;
            q.carr_freq_error_hz = d_loop_fields.carr_freq_error_hz;
            q.carr_error_filt_hz = d_loop_fields.carr_error_filt_hz;
            q.code_error_chips = d_loop_fields.code_error_chips;
            q.code_error_filt_chips = d_loop_fields.code_error_filt_chips;
        }
    else
        {
            d_loop_fields = r;
        }
    d_dump_records.push_back(q);
    d_dump_tow.push_back(tow_ms);
    d_dump_wn.push_back(wn);
}


void dll_pll_veml_tracking_hip::flush_dump()
{
    if (d_dump_records.empty() || d_dump_path.empty()) return;
    const uint32_t prn = d_acquisition_gnss_synchro != nullptr ? d_acquisition_gnss_synchro->PRN : 0U;
    if (gsh_trk_write_dump(d_dump_path.c_str(), 1, &d_conf, prn, d_dump_records.data(), static_cast<int>(d_dump_records.size()), d_dump_tow.data(), d_dump_wn.data()) != GSH_OK)
        LOG(WARNING) << "Exception writing trk dump file " << gsh_last_error();
    d_dump_records.clear();
    d_dump_tow.clear();
    d_dump_wn.clear();
}


void dll_pll_veml_tracking_hip::fill_symbol(Gnss_Synchro* out, const gsh_trk_epoch& r, bool loss_of_lock, uint64_t tow_ms) const
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
    current_synchro_data.TOW_at_current_symbol_ms = tow_ms;               // trk.cc:2255
    current_synchro_data.fs = static_cast<int64_t>(d_trk_parameters.fs_in);  // trk.cc:2287-2294
    current_synchro_data.Tracking_sample_counter = r.sample_counter;      // nitems_read(0) during the call = first sample of the period
    current_synchro_data.Flag_valid_symbol_output = !loss_of_lock;
    current_synchro_data.Flag_PLL_180_deg_phase_locked = (r.symbol_flags & 2) != 0;
    *out = current_synchro_data;
}


// consume_each, after making sure that what goes back to the scheduler is no longer being read by a DMA this block (or a sibling on the same buffer) queued:
// push() does not wait for its copies -- the look-ahead a block is offered beyond what it consumes stays valid, and is copied while the blocks work
void dll_pll_veml_tracking_hip::give_back(int n_items)
{
    if (d_usable && n_items > 0 && d_runtime) (void)d_runtime->release_input(this->nitems_read(0) + static_cast<uint64_t>(n_items));
    consume_each(n_items);
}


// an engine failure: the channel goes back to acquisition the reference's way ("events" 3), the block to standby
void dll_pll_veml_tracking_hip::drop_channel(int ninput)
{
    LOG(ERROR) << "dll_pll_veml_tracking_hip: " << d_error;
    d_state = 0;
    if (d_slot >= 0) d_runtime->stop(d_slot);
    this->message_port_pub(pmt::mp("events"), pmt::from_long(3));
    give_back(ninput);
}


int dll_pll_veml_tracking_hip::general_work(int noutput_items, gr_vector_int& ninput_items, gr_vector_const_void_star& input_items,
    gr_vector_void_star& output_items)
{
    gr::thread::scoped_lock l(d_setlock);
    const auto* in = reinterpret_cast<const gr_complex*>(input_items[0]);
    auto* out = reinterpret_cast<Gnss_Synchro*>(output_items[0]);
    const uint64_t read_pos = this->nitems_read(0);

    // What this block sees of the stream goes to the device ring in every call, whatever the state (for samples that are already there -- another
    // channel of the stream was first -- a comparison of two indices): the ring then always holds the stream up to the front-runner's read
    // pointer, and a channel that starts anywhere behind it finds its samples resident (the reference's channels read one shared buffer,
    // gnss_flowgraph.cc:1227-1231).
    bool pushed = true;
    if (d_usable) pushed = d_runtime->push(in, read_pos, static_cast<uint64_t>(ninput_items[0]), d_state >= 2);

    switch (d_state)
        {
        case 0:  // standby: consume at full throttle (trk.cc:1941-1947)
            give_back(ninput_items[0]);
            return 0;
        case 1:  // pull-in: skip samples until the incoming signal is aligned with the local replica (trk.cc:1949-1978)
            {
                int32_t samples_offset = 0, first_len = 0;
                if (d_slot < 0 || !d_runtime->start(d_slot, d_code.data(), d_conf.track_pilot ? d_data_code.data() : nullptr, static_cast<int>(d_code.size()), read_pos,
                                      d_acquisition_gnss_synchro->Acq_delay_samples, d_acquisition_gnss_synchro->Acq_doppler_hz,
                                      d_acquisition_gnss_synchro->Acq_samplestamp_samples, &samples_offset, &first_len))
                    {
                        d_error = d_slot >= 0 ? d_runtime->last_error(d_slot) : std::string("no device loop");
                        drop_channel(ninput_items[0]);
                        return 0;
                    }
                d_current_prn_length_samples = first_len;  // trk.cc:1964
                d_loop_fields = gsh_trk_epoch{};
                d_state = 2;
                give_back(samples_offset);
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
            d_runtime->stop(d_slot);
            d_state = 0;
            std::cout << "Loss of lock in channel " << d_channel << " (telemetry fault)!\n";
            this->message_port_pub(pmt::mp("events"), pmt::from_long(3));
            uint64_t tow_ms = 0;
            uint32_t wn = 0;
            estimate_tow(read_pos, d_current_prn_length_samples, &tow_ms, &wn);
            fill_symbol(&out[0], r, true, tow_ms);
            give_back(d_current_prn_length_samples);
            return 1;
        }
    if (!pushed)
        {
            d_error = d_runtime->last_error(d_slot);
            drop_channel(ninput_items[0]);
            return 0;
        }
    const int want = std::min(std::max(noutput_items, 1), d_periods_per_call);
    const int done = d_runtime->take(d_slot, read_pos + static_cast<uint64_t>(ninput_items[0]), want, d_records.data());
    if (done < 0)
        {
            d_error = d_runtime->last_error(d_slot);
            drop_channel(ninput_items[0]);
            return 0;
        }
    int produced = 0;
    int64_t consumed = 0;
    for (int e = 0; e < done; e++)
        {
            const gsh_trk_epoch& r = d_records[e];
            d_last = r;
            uint64_t tow_ms = 0;
            uint32_t wn = 0;
            estimate_tow(r.sample_counter, d_current_prn_length_samples, &tow_ms, &wn);  // with the length the PREVIOUS period left, as the top of general_work sees it
            if (r.flags & 2)  // loss of lock declared by the device's lock detectors: "events" 3 and an invalid symbol (trk.cc:1208-1221, 2009-2014)
                {
                    std::cout << "Loss of lock in channel " << d_channel << "!\n";
                    this->message_port_pub(pmt::mp("events"), pmt::from_long(3));
                    d_state = 0;
                    collect_time_tags(r.sample_counter, r.sample_counter + static_cast<uint64_t>(std::max(d_current_prn_length_samples, 0)));
                    emit_tags(this->nitems_written(0) + static_cast<uint64_t>(produced), r.sample_counter, r.sample_counter);
                    fill_symbol(&out[produced++], r, true, tow_ms);
                    // clear_tracking_vars leaves d_current_prn_length_samples at its last value; the block consumes it (trk.cc:2283)
                    consumed += static_cast<int64_t>(d_current_prn_length_samples);
                    break;
                }
            d_current_prn_length_samples = r.prn_length_samples;  // update_tracking_vars, trk.cc:1409-1421
            dump_record(r, tow_ms, wn);
            collect_time_tags(r.sample_counter, r.sample_counter + static_cast<uint64_t>(std::max(r.prn_length_samples, 0)));
            consumed += r.prn_length_samples;
            if (r.symbol_flags & 1)
                {
                    emit_tags(this->nitems_written(0) + static_cast<uint64_t>(produced), r.sample_counter, r.sample_counter);
                    fill_symbol(&out[produced++], r, false, tow_ms);
                }
        }
    flush_dump();
    give_back(static_cast<int>(consumed));
    return produced;
}
