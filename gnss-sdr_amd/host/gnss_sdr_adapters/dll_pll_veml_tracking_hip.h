/*!
 * \file dll_pll_veml_tracking_hip.h
 * \brief GNU Radio block with dll_pll_veml_tracking's contract over the MI355X device-closed loop (gsh_trk_*).
 *
 * Contract kept from the reference block (src/algorithms/tracking/gnuradio_blocks/dll_pll_veml_tracking.{h,cc}, "trk.cc"):
 *   ports      1 input of gr_complex, 1 output of Gnss_Synchro, message out-port "events" (3 = loss of lock, trk.cc:1218), message
 *              in-port "telemetry_to_trk" (int 1 = telemetry fault -> forced loss of lock, trk.cc:757-769);          trk.cc:143-166
 *   rates      set_relative_rate(1 / vector_length), forecast = 2 * vector_length per output item;                  trk.cc:143-148, 747-754
 *   lifecycle  set_channel, set_gnss_synchro, start_tracking (reads Acq_delay_samples / Acq_doppler_hz / Acq_samplestamp_samples),
 *              stop_tracking; state 0 consumes everything it is offered, state 1 aligns the stream with the local replica
 *              (consume_each(samples_offset)), states 2..4 consume one code period per call;                       trk.cc:793-1116, 1898-2330
 *   output     one Gnss_Synchro per telemetry symbol (state 4, d_current_data_symbol == 0) filled as trk.cc:2212-2236 and :2285-2294,
 *              and one with Flag_valid_symbol_output = false at loss of lock (:2009-2014, 2285-2294).
 *   telemetry  "telemetry_to_trk" also carries the TOW hand-back (std::shared_ptr<TOW_to_trk>, Dll_Pll_Conf::tow_to_trk): kept and turned into
 *              TOW_at_current_symbol_ms of every item exactly as trk.cc:771-779, 1921-1935, 2255 do; "timetag" / "sensor_data" stream tags are
 *              re-generated on the output as trk.cc:2256-2316 does;
 *   dump       Dll_Pll_Conf::dump / dump_filename: one log_data record per logged period (trk.cc:707-735, 1599-1702, 1851-1873) written through
 *              gsh_trk_write_dump into <dump_filename><channel>.dat, the file the reference block writes (dump_mat, the matio conversion of that
 *              file in the reference's destructor, is left to the reference's own save_matfile / utils readers).
 * What is different inside: correlation, lock detectors, discriminators, loop filters, NCO update and the symbol state machine run on the GPU,
 * and not one launch per block: every block of a stream holds a slot of ONE Hip_Tracking_Runtime, whose launches advance all channels that
 * have samples (hip_tracking_runtime.h).  The host block feeds samples (de-duplicated by absolute index) and turns records into items.
 * hip_periods_per_call > 1 lets one call take several periods when the scheduler offers enough samples; the default 1 is the reference's
 * set_max_noutput_items(1) cadence (the shared launch still serves all channels).
 */
#ifndef GNSS_SDR_DLL_PLL_VEML_TRACKING_HIP_H
#define GNSS_SDR_DLL_PLL_VEML_TRACKING_HIP_H

#include "dll_pll_conf.h"
#include "dll_pll_conf_hip.h"
#include "gnss_block_interface.h"
#include "gnss_sdr_hip.h"
#include "gnss_time.h"
#include "hip_tracking_runtime.h"
#include "tow_to_trk.h"
#include <gnuradio/block.h>
#include <gnuradio/gr_complex.h>
#include <gnuradio/types.h>
#include <pmt/pmt.h>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

class Gnss_Synchro;
class dll_pll_veml_tracking_hip;

using dll_pll_veml_tracking_hip_sptr = gnss_shared_ptr<dll_pll_veml_tracking_hip>;

//! runtime: the channel-batching runtime (and through it the device sample ring) this block shares with the other channels of its stream
dll_pll_veml_tracking_hip_sptr dll_pll_veml_make_tracking_hip(const Dll_Pll_Conf& conf_, int hip_periods_per_call,
    std::shared_ptr<Hip_Tracking_Runtime> runtime);

class dll_pll_veml_tracking_hip : public gr::block
{
public:
    ~dll_pll_veml_tracking_hip() override;

    void set_channel(uint32_t channel);
    void set_gnss_synchro(Gnss_Synchro* p_gnss_synchro);
    void start_tracking();
    void stop_tracking();

    int general_work(int noutput_items, gr_vector_int& ninput_items, gr_vector_const_void_star& input_items,
        gr_vector_void_star& output_items) override;
    void forecast(int noutput_items, gr_vector_int& ninput_items_required) override;

    //! false when the signal / configuration has no MI355X loop or no GPU is present (the adapter then reports item_size() == 0)
    bool usable() const { return d_usable; }
    const std::string& last_error() const { return d_error; }
    const gsh_trk_conf& trk_conf() const { return d_conf; }

    // test / monitoring view
    int32_t state() const { return d_state; }
    const gsh_trk_epoch& last_record() const { return d_last; }
    const std::shared_ptr<Hip_Tracking_Runtime>& runtime() const { return d_runtime; }

private:
    friend dll_pll_veml_tracking_hip_sptr dll_pll_veml_make_tracking_hip(const Dll_Pll_Conf& conf_, int hip_periods_per_call,
        std::shared_ptr<Hip_Tracking_Runtime> runtime);
    dll_pll_veml_tracking_hip(const Dll_Pll_Conf& conf_, int hip_periods_per_call, std::shared_ptr<Hip_Tracking_Runtime> runtime);
    void msg_handler_telemetry_to_trk(const pmt::pmt_t& msg);
    void fill_symbol(Gnss_Synchro* out, const gsh_trk_epoch& r, bool loss_of_lock, uint64_t tow_ms) const;
    void estimate_tow(uint64_t period_start, int32_t prn_length_before, uint64_t* tow_ms, uint32_t* wn) const;
    void collect_time_tags(uint64_t from, uint64_t to);
    void emit_tags(uint64_t out_item, uint64_t tracking_sample_counter, uint64_t period_start);
    void dump_record(const gsh_trk_epoch& r, uint64_t tow_ms, uint32_t wn);
    void flush_dump();
    void drop_channel(int ninput);
    void give_back(int n_items);

    Dll_Pll_Conf d_trk_parameters;
    gsh_trk_conf d_conf{};
    Hip_Trk_Signal d_signal;
    std::shared_ptr<Hip_Tracking_Runtime> d_runtime;
    int d_slot{-1};
    Gnss_Synchro* d_acquisition_gnss_synchro{nullptr};
    std::vector<gsh_trk_epoch> d_records;
    std::vector<float> d_code, d_data_code;
    gsh_trk_epoch d_last{};
    std::string d_error;
    int d_periods_per_call{1};
    int32_t d_current_prn_length_samples{0};  // what the last processed period consumed (trk.cc:1925 reads it at the top of the next call)
    // TOW hand-back and time tags (trk.cc:771-779, 1921-1935, 2256-2316)
    std::shared_ptr<TOW_to_trk> d_last_tow_received;
    std::string d_signal_type;
    GnssTime d_last_timetag{};
    uint64_t d_last_timetag_samplecounter{0};
    bool d_timetag_waiting{false};
    // dump (trk.cc:707-735, 1851-1873)
    bool d_dump{false};
    std::string d_dump_filename;       // path + base name, without channel number and extension
    std::string d_dump_path;           // the open file: <d_dump_filename><channel>.dat
    std::vector<gsh_trk_epoch> d_dump_records;
    std::vector<uint64_t> d_dump_tow;
    std::vector<uint32_t> d_dump_wn;
    gsh_trk_epoch d_loop_fields{};     // run_dll_pll's outputs of the last period that ran the loop: what log_data prints in state 3
    int32_t d_state{0};        // 0 standby, 1 pull-in, 2 tracking (the device keeps the reference's states 2 / 3 / 4)
    uint32_t d_channel{0};
    bool d_usable{false};
    bool d_force_loss_of_lock{false};  // telemetry fault received (trk.cc:763-768)
};

#endif  // GNSS_SDR_DLL_PLL_VEML_TRACKING_HIP_H
