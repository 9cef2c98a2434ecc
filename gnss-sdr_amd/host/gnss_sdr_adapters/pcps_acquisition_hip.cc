/*!
 * \file pcps_acquisition_hip.cc
 * \brief See the header.  State machine: 0 = (re)start, 1 = fill the dwell buffer, 2 = run one dwell on the GPU
 *        -- the same three states the reference block steps through in general_work (acq.cc:749-853); the dwell itself is
 *        blocking, like the reference's default (acq_conf.h:72).  BUILT ONLY INSIDE A gnss-sdr TREE.
 */
#include "pcps_acquisition_hip.h"
#include "GLONASS_L1_L2_CA.h"
#include "hip_mat5_writer.h"
#include <cstring>
#include <gnuradio/io_signature.h>
#if USE_GLOG_AND_GFLAGS
#include <glog/logging.h>
#else
#include <absl/log/log.h>
#endif
#include <pmt/pmt.h>
#include <algorithm>
#include <cmath>
#include <filesystem>
#include <iostream>
#include <utility>
#include <vector>

namespace
{
// get_dump_filename, acq.cc:58-92: directory and base name apart, the extension dropped, the directory created; empty when it cannot be
std::string hip_acq_dump_filename(std::string dump_filename)
{
    std::string dump_path;
    if (dump_filename.find_last_of('/') != std::string::npos)
        {
            const auto last_slash_index = dump_filename.find_last_of('/');
            dump_path = dump_filename.substr(0, last_slash_index);
            dump_filename = dump_filename.substr(last_slash_index + 1);
        }
    else
        {
            dump_path = std::string(".");
        }
    if (dump_filename.empty()) dump_filename = "acquisition";
    if (dump_filename.substr(1).find_last_of('.') != std::string::npos) dump_filename = dump_filename.substr(0, dump_filename.find_last_of('.'));
    std::error_code ec;
    std::filesystem::create_directories(dump_path, ec);
    if (ec)
        {
            std::cerr << "GNSS-SDR cannot create dump file for the Acquisition block. The dump path is: " << dump_path << '\n';
            return std::string{};
        }
    return dump_path + static_cast<char>(std::filesystem::path::preferred_separator) + dump_filename;
}
}  // namespace

pcps_acquisition_hip_sptr pcps_make_acquisition_hip(const Hip_Acq_Conf& conf, int device, bool blocking_on_standby, std::shared_ptr<Hip_Acquisition_Runtime> runtime)
{
    return pcps_acquisition_hip_sptr(new pcps_acquisition_hip(conf, device, blocking_on_standby, std::move(runtime)));
}


pcps_acquisition_hip::pcps_acquisition_hip(const Hip_Acq_Conf& conf, int device, bool blocking_on_standby, std::shared_ptr<Hip_Acquisition_Runtime> runtime)
    : acquisition_impl_interface("pcps_acquisition_hip",
          gr::io_signature::make(1, 1, conf.cshort ? sizeof(std::complex<int16_t>) : sizeof(gr_complex)),  // acq.cc:102-103 (it_size)
          gr::io_signature::make(0, 1, sizeof(Gnss_Synchro))),
      d_core(conf, device),
      d_data_buffer(conf.cshort ? 0U : d_core.consumed_samples()),
      d_data_buffer_sc(conf.cshort ? d_core.consumed_samples() : 0U),
      d_cshort(conf.cshort),
      d_blocking_on_standby(blocking_on_standby),
      d_dump_filename(conf.dump ? hip_acq_dump_filename(conf.dump_filename) : std::string{}),
      d_dump_channel(conf.dump_channel)
{
    d_dump = !d_dump_filename.empty();  // acq.cc:106, 120
    this->message_port_register_out(pmt::mp("events"));
    // a shared runtime is only of use to a block whose dwell is the runtime's dwell (same transform, same Doppler grid, same statistic)
    if (runtime && runtime->ok() && d_core.ok() && runtime->same_geometry(d_core.engine_conf()))
        {
            d_runtime = std::move(runtime);
            d_slot = d_runtime->attach();
            if (d_slot < 0) d_runtime.reset();
        }
}


pcps_acquisition_hip::~pcps_acquisition_hip()
{
    if (d_runtime && d_slot >= 0) d_runtime->detach(d_slot);
}


// the block stops caring about the window it had announced (deactivated, restarted): a batch must not wait for it
void pcps_acquisition_hip::leave_shared_window()
{
    if (d_runtime) d_runtime->withdraw(d_slot);
    d_shared_dwell = false;
    d_skip = 0;
}


void pcps_acquisition_hip::set_local_code(std::complex<float>* code)
{
    // acq.cc:220-223 + is_fdma (:252-272): a GLONASS satellite sits on its own FDMA carrier; the offset enters the wipe-off only
    int32_t doppler_bias = 0;
    if (d_gnss_synchro != nullptr)
        {
            const auto is_1G = strcmp(d_gnss_synchro->Signal, "1G") == 0;
            const auto is_2G = strcmp(d_gnss_synchro->Signal, "2G") == 0;
            if (is_1G || is_2G)
                {
                    const auto freq = is_1G ? DFRQ1_GLO : DFRQ2_GLO;
                    doppler_bias = static_cast<int32_t>(freq * GLONASS_PRN.at(d_gnss_synchro->PRN));
                }
        }
    gr::thread::scoped_lock lock(d_setlock);
    d_core.set_doppler_bias(doppler_bias);
    d_core.set_local_code(code);
    if (d_runtime && !d_runtime->set_local_code(d_slot, code))  // the same replica in this block's slot of the shared handle
        {
            d_runtime->detach(d_slot);
            d_runtime.reset();
        }
}


void pcps_acquisition_hip::set_active(bool active)
{
    gr::thread::scoped_lock lock(d_setlock);
    d_active = active;
    if (!active)
        leave_shared_window();
    else if (d_runtime && d_state == 0 && d_core.next_dwell_is_shareable())
        d_runtime->searching(d_slot);  // batches about to close wait (bounded) for this channel's first window
}


void pcps_acquisition_hip::set_doppler_center(int32_t doppler_center)
{
    gr::thread::scoped_lock lock(d_setlock);
    d_core.set_doppler_center(doppler_center);
}


// pcps_acquisition::dump_results, acq.cc:354-406: file name, variables, classes and dimensions as there.  The container is MAT-file level 5 (written here,
// host/hip_mat5_writer.h) where the reference's matio writes 7.3.  make_two_steps searches: acq_grid is the wide grid of step one, acq_grid_narrow the narrow one of
// step two -- the core keeps host copies of both, since on the device the narrow grid takes the place of the wide grid's first rows (Hip_Pcps_Acquisition_Core::read_grid).
void pcps_acquisition_hip::dump_results(const Hip_Pcps_Acquisition_Core::AcquisitionResult& result)
{
    d_dump_number++;
    std::string filename = d_dump_filename;
    filename.append("_");
    filename.append(1, d_gnss_synchro->System);
    filename.append("_");
    filename.append(1, d_gnss_synchro->Signal[0]);
    filename.append(1, d_gnss_synchro->Signal[1]);
    filename.append("_ch_");
    filename.append(std::to_string(d_channel));
    filename.append("_");
    filename.append(std::to_string(d_dump_number));
    filename.append("_sat_");
    filename.append(std::to_string(d_gnss_synchro->PRN));
    filename.append(".mat");
    const size_t eff = d_core.effective_fft_size(), bins = d_core.num_doppler_bins();
    std::vector<float> grid(eff * bins);
    if (!d_core.read_grid(grid.data()))
        {
            std::cout << "Acquisition dump: the search grid could not be read back: " << d_core.last_error() << '\n';
            return;
        }
    Hip_Mat5_Writer w(filename);
    if (!w.ok())
        {
            std::cout << "Unable to create or open Acquisition dump file\n";
            return;
        }
    w.matrix("acq_grid", grid.data(), eff, bins);  // arma::fmat(d_effective_fft_size, d_num_doppler_bins): one column per Doppler bin
    w.scalar<int32_t>("doppler_max", static_cast<int32_t>(d_core.doppler_max()));
    w.scalar<int32_t>("doppler_step", static_cast<int32_t>(d_core.doppler_step()));
    w.scalar<int32_t>("positive_acq", result.positive_acq ? 1 : 0);
    w.scalar<float>("acq_doppler_hz", static_cast<float>(d_gnss_synchro->Acq_doppler_hz));
    w.scalar<float>("acq_delay_samples", static_cast<float>(d_gnss_synchro->Acq_delay_samples));
    w.scalar<float>("test_statistic", result.test_statistics);
    w.scalar<float>("threshold", d_core.get_threshold());
    w.scalar<float>("input_power", d_core.input_power());
    w.scalar<uint64_t>("sample_counter", result.sample_count);
    w.scalar<uint32_t>("PRN", d_gnss_synchro->PRN);
    w.scalar<int32_t>("num_dwells", static_cast<int32_t>(result.num_dwells));
    if (d_core.make_two_steps())  // acq.cc:392-400
        {
            const size_t bins2 = d_core.num_doppler_bins_step2();
            std::vector<float> narrow(eff * bins2);
            d_core.read_narrow_grid(narrow.data());
            const float doppler_grid_narrow_min = d_core.doppler_center_step_two() - static_cast<float>(std::floor(static_cast<double>(bins2) / 2.0)) * d_core.doppler_step2();
            w.matrix("acq_grid_narrow", narrow.data(), eff, bins2);
            w.scalar<float>("doppler_step_narrow", d_core.doppler_step2());
            w.scalar<float>("doppler_grid_narrow_min", doppler_grid_narrow_min);
        }
    w.close();
}


void pcps_acquisition_hip::set_threshold(float threshold)
{
    gr::thread::scoped_lock lock(d_setlock);
    d_core.set_threshold(threshold);
}


void pcps_acquisition_hip::set_state(int32_t state)
{
    gr::thread::scoped_lock lock(d_setlock);
    d_state = state;
    if (state == 0)
        {
            d_core.reset();
            leave_shared_window();
        }
}


// The dwell runs with d_setlock released (it waits for the GPU; setters from the control thread must get through meanwhile), so everything it needs of the
// block's shared-window state is handed over as it stood under the lock: the runtime is held by the caller's own reference for the duration (set_local_code may
// drop the block's), and the slot / window / "this window is a shared one" values cannot change under its feet.
void pcps_acquisition_hip::run_dwell(uint64_t sample_count, const std::shared_ptr<Hip_Acquisition_Runtime>& runtime, int slot, uint64_t window, bool shared_dwell)
{
    Hip_Pcps_Acquisition_Core::AcquisitionResult result;
    const bool was_step_two = d_core.step_two();
    Hip_Pcps_Acquisition_Core::Outcome outcome;
    if (shared_dwell && runtime)
        {
            // every channel that buffered this window joins one batch: the Doppler-wiped forward transforms are computed once for all of them
            gsh_acq_result r{};
            const bool ok = runtime->dwell(slot, window, d_data_buffer.data(), &r);
            outcome = d_core.acquisition_core_shared(sample_count, ok, r, &result);
        }
    else
        {
            outcome = d_cshort ? d_core.acquisition_core(sample_count, d_data_buffer_sc.data(), &result)
                               : d_core.acquisition_core(sample_count, d_data_buffer.data(), &result);
        }
    // What the search decided is filed under the lock; the channel is TOLD with the lock released.  The reference calls ChannelFsm::Event_valid_acquisition with d_setlock
    // held (acquisition_core locks it for the whole dwell, acq.cc:650, 318-326) while ChannelFsm::Event_start_acquisition, holding the FSM's mutex, calls
    // acq_->reset() -> set_active(), which takes d_setlock (channel_fsm.cc:81-93, 178-182): two locks taken in both orders.  The FSM's states keep the reference out of
    // that corner most of the time; here the corner does not exist.
    std::shared_ptr<ChannelFsm> fsm_to_tell;
    long event = 0;
    if (outcome == Hip_Pcps_Acquisition_Core::ACQ_ERROR) LOG(ERROR) << "pcps_acquisition_hip: channel " << d_channel << ": " << d_core.last_error() << " -- reported as a failed acquisition";
    {
        gr::thread::scoped_lock lock(d_setlock);
        if (outcome != Hip_Pcps_Acquisition_Core::ACQ_ERROR && d_gnss_synchro != nullptr) d_core.update_synchro(result, d_gnss_synchro);
        if (outcome != Hip_Pcps_Acquisition_Core::ACQ_ERROR && result.search_complete && d_dump && d_channel == d_dump_channel && d_gnss_synchro != nullptr)
            dump_results(result);  // acq.cc:719-723
        switch (outcome)
            {
            case Hip_Pcps_Acquisition_Core::ACQ_POSITIVE:
                d_state = 0;
                d_active = false;
                fsm_to_tell = d_channel_fsm.lock();
                if (!fsm_to_tell) event = 1;
                break;
            case Hip_Pcps_Acquisition_Core::ACQ_CONTINUE:
                d_buffer_count = 0U;
                // step one crossed its threshold and armed the fine-Doppler step: the reference restarts from state 0 (acq.cc:607, 617-624);
                // otherwise gather the next non-coherent dwell (acq.cc:694-698)
                d_state = (!was_step_two && d_core.step_two()) ? 0 : 1;
                break;
            case Hip_Pcps_Acquisition_Core::ACQ_NEGATIVE:
            case Hip_Pcps_Acquisition_Core::ACQ_ERROR:  // a GPU failure must look like "not found", never throw here
            default:
                d_state = 0;
                d_active = false;
                event = 2;
                break;
            }
    }
    if (fsm_to_tell)
        fsm_to_tell->Event_valid_acquisition();  // acq.cc:322-326: the channel FSM is told directly, to keep the acquisition-to-tracking delay short
    else if (event != 0)
        this->message_port_pub(pmt::mp("events"), pmt::from_long(event));  // 1 = ACQ_SUCCESS, 2 = ACQ_FAIL (acq.cc:331, 350)
}


int pcps_acquisition_hip::general_work(int /*noutput_items*/, gr_vector_int& ninput_items, gr_vector_const_void_star& input_items,
    gr_vector_void_star& /*output_items*/)
{
    gr::thread::scoped_lock lk(d_setlock);
    if (!d_active)
        {
            if (!d_blocking_on_standby)
                {
                    d_sample_count += static_cast<uint64_t>(ninput_items[0]);
                    consume_each(ninput_items[0]);
                }
            return 0;
        }
    if (d_state == 0)
        {
            if (d_gnss_synchro != nullptr)
                {
                    d_gnss_synchro->Acq_delay_samples = 0.0;
                    d_gnss_synchro->Acq_doppler_hz = 0.0;
                    d_gnss_synchro->Acq_samplestamp_samples = 0ULL;
                    d_gnss_synchro->Acq_doppler_step = 0U;
                }
            d_buffer_count = 0U;
            d_state = 1;
            // a dwell that may be shared starts on the runtime's grid: the next multiple of the dwell length in absolute sample index, less than one
            // dwell length ahead.  The block says so now, so that the batch of that window waits for it.
            if (d_runtime && d_core.next_dwell_is_shareable())
                {
                    // (one step from "searching" / a previous window to the new window: withdrawing first would leave the channel unannounced for a moment,
                    // and a batch evaluated in that moment closes without it)
                    d_window = d_runtime->next_window(d_sample_count);
                    d_skip = static_cast<uint32_t>(d_window - d_sample_count);
                    d_shared_dwell = true;
                    d_runtime->announce(d_slot, d_window);
                }
            else
                {
                    leave_shared_window();
                }
        }
    else if (d_state == 1 && d_skip > 0)
        {
            const uint32_t pass = std::min<uint32_t>(d_skip, static_cast<uint32_t>(ninput_items[0]));
            d_skip -= pass;
            d_sample_count += pass;
            consume_each(static_cast<int>(pass));
        }
    else if (d_state == 1)
        {
            const uint32_t want = d_core.consumed_samples();
            const uint32_t room = want - std::min(d_buffer_count, want);
            const uint32_t take = std::min<uint32_t>(room, static_cast<uint32_t>(ninput_items[0]));
            if (d_cshort)  // acq.cc:789-798
                {
                    const auto* in = reinterpret_cast<const std::complex<int16_t>*>(input_items[0]);
                    std::copy(in, in + take, d_data_buffer_sc.begin() + d_buffer_count);
                }
            else
                {
                    const auto* in = reinterpret_cast<const gr_complex*>(input_items[0]);
                    std::copy(in, in + take, d_data_buffer.begin() + d_buffer_count);
                }
            if (d_buffer_count >= want) d_state = 2;  // same one-call latency as acq.cc:807-811
            d_buffer_count += take;
            d_sample_count += take;
            consume_each(static_cast<int>(take));
        }
    else
        {
            const uint64_t stamp = d_sample_count;
            // (decided and recorded under the lock: is this buffered window still one of a shared batch?)
            const std::shared_ptr<Hip_Acquisition_Runtime> runtime = d_runtime;
            const int slot = d_slot;
            const uint64_t window = d_window;
            bool shared = d_shared_dwell && runtime;
            if (shared && !d_core.next_dwell_is_shareable())
                {
                    // something changed while the window was buffered (set_doppler_center ...): this dwell is the block's own after all
                    runtime->withdraw(slot);
                    shared = false;
                }
            d_shared_dwell = false;
            lk.unlock();
            run_dwell(stamp, runtime, slot, window, shared);
            consume_each(0);
        }
    return 0;
}
