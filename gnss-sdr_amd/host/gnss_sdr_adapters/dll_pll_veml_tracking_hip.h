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
 * What is different inside: correlation, lock detectors, discriminators, loop filters, NCO update and the symbol state machine of one
 * code period run in ONE kernel launch on the GPU (Hip_Tracking_Loop); the host block only feeds samples and turns records into items.
 * hip_periods_per_call > 1 lets one call run several periods when the scheduler offers enough samples (one launch instead of several);
 * the default 1 is the reference's set_max_noutput_items(1) behaviour.
 */
#ifndef GNSS_SDR_DLL_PLL_VEML_TRACKING_HIP_H
#define GNSS_SDR_DLL_PLL_VEML_TRACKING_HIP_H

#include "dll_pll_conf.h"
#include "dll_pll_conf_hip.h"
#include "gnss_block_interface.h"
#include "gnss_sdr_hip.h"
#include "hip_tracking_loop.h"
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

dll_pll_veml_tracking_hip_sptr dll_pll_veml_make_tracking_hip(const Dll_Pll_Conf& conf_, int hip_device, int hip_periods_per_call,
    std::shared_ptr<Hip_Sample_Ring> shared_ring);

class dll_pll_veml_tracking_hip : public gr::block
{
public:
    ~dll_pll_veml_tracking_hip() override = default;

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

private:
    friend dll_pll_veml_tracking_hip_sptr dll_pll_veml_make_tracking_hip(const Dll_Pll_Conf& conf_, int hip_device, int hip_periods_per_call,
        std::shared_ptr<Hip_Sample_Ring> shared_ring);
    dll_pll_veml_tracking_hip(const Dll_Pll_Conf& conf_, int hip_device, int hip_periods_per_call, std::shared_ptr<Hip_Sample_Ring> shared_ring);
    void msg_handler_telemetry_to_trk(const pmt::pmt_t& msg);
    void fill_symbol(Gnss_Synchro* out, const gsh_trk_epoch& r, bool loss_of_lock) const;

    Dll_Pll_Conf d_trk_parameters;
    gsh_trk_conf d_conf{};
    Hip_Trk_Signal d_signal;
    std::unique_ptr<Hip_Tracking_Loop> d_loop;
    std::shared_ptr<Hip_Sample_Ring> d_shared_ring;
    Gnss_Synchro* d_acquisition_gnss_synchro{nullptr};
    std::vector<gsh_trk_epoch> d_records;
    std::vector<float> d_code, d_data_code;
    gsh_trk_epoch d_last{};
    std::string d_error;
    int d_device{0};
    int d_periods_per_call{1};
    int32_t d_state{0};        // 0 standby, 1 pull-in, 2 tracking (the device keeps the reference's states 2 / 3 / 4)
    uint32_t d_channel{0};
    bool d_usable{false};
    bool d_force_loss_of_lock{false};  // telemetry fault received (trk.cc:763-768)
};

#endif  // GNSS_SDR_DLL_PLL_VEML_TRACKING_HIP_H
