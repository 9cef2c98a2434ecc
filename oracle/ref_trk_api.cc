/*
 * TEST INFRASTRUCTURE ONLY (oracle/_ref build) -- never linked into the product.
 *
 * C driver over the REFERENCE's own tracking chain, compiled from where it lies under /root/reference:
 *   adapters  src/algorithms/tracking/adapters/{base_dll_pll_tracking, gps_l1_ca_dll_pll_tracking, galileo_e1_dll_pll_veml_tracking,
 *             gps_l5_dll_pll_tracking}.cc            (TrackingInterface: configuration -> Dll_Pll_Conf -> block)
 *   block     src/algorithms/tracking/gnuradio_blocks/dll_pll_veml_tracking.cc    (the 2 300-line loop)
 *   libs      dll_pll_conf.cc, cpu_multicorrelator_real_codes.cc + the volk_gnsssdr `_generic` protokernels, tracking_discriminators.cc,
 *             tracking_loop_filter.cc, tracking_FLL_PLL_filter.cc, lock_detectors.cc, exponential_smoother.cc, bit_synchronizer.cc,
 *             the PRN generators, gnss_sdr_flags.cc (defaults of the command-line flags)
 * against tests/host/mock_gnuradio (gr::block without a scheduler: the harness below plays it) and oracle/shim_blocks (gflags macros,
 * boost::circular_buffer, matio).  Nothing of the loop is restated here: every number that comes out is the reference's.
 *
 * Driven the way Channel / the scheduler drive it: adapter(configuration, role, 1, 1) -> set_channel -> set_gnss_synchro ->
 * start_tracking -> general_work(noutput = 1, ninput_items, input_items, output_items) over a stream the caller slices, with
 * nitems_read advanced by what the block consumed.
 */
#include <algorithm>
#include <any>
#include <array>
#include <complex>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <functional>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include <gnuradio/block.h>
#include <volk_gnsssdr/volk_gnsssdr_alloc.h>
#include <boost/circular_buffer.hpp>

#include "gnss_synchro.h"
#include "gnss_time.h"
#include "in_memory_configuration.h"
#include "tow_to_trk.h"

#define private public
#define protected public
#include "dll_pll_veml_tracking.h"
#include "base_dll_pll_tracking.h"
#undef protected
#undef private
#include "beidou_b1i_dll_pll_tracking.h"
#include "beidou_b3i_dll_pll_tracking.h"
#include "galileo_e1_dll_pll_veml_tracking.h"
#include "galileo_e5a_dll_pll_tracking.h"
#include "galileo_e5b_dll_pll_tracking.h"
#include "galileo_e6_dll_pll_tracking.h"
#include "glonass_l1_ca_dll_pll_tracking.h"
#include "glonass_l2_ca_dll_pll_tracking.h"
#include "gps_l1_ca_dll_pll_tracking.h"
#include "gps_l2_m_dll_pll_tracking.h"
#include "gps_l5_dll_pll_tracking.h"
#include "qzss_l1_dll_pll_tracking.h"
#include "qzss_l5_dll_pll_tracking.h"

namespace
{
struct Handle
{
    InMemoryConfiguration cfg;
    std::shared_ptr<TrackingInterface> adapter;
    dll_pll_veml_tracking_sptr own_block;  // reftrk_create_block: the block without an adapter around it
    dll_pll_veml_tracking* block{nullptr};
    Gnss_Synchro synchro{};
    std::vector<Gnss_Synchro> out_items;
};
}  // namespace

extern "C" {

struct reftrk_output  /* one Gnss_Synchro as the tracking block fills it (trk.cc:2280-2300) + the loop state behind it */
{
    double fs, prompt_i, prompt_q, cn0_db_hz, carrier_doppler_hz, carrier_phase_rads, code_phase_samples;
    uint64_t tracking_sample_counter;
    int32_t flag_valid_symbol_output, correlation_length_ms, flag_pll_180_deg_phase_locked, prn;
    /* loop state after the call (private members of the block) */
    int32_t state, current_prn_length_samples, n_correlator_taps, cn0_estimation_counter, carrier_lock_fail_counter, code_lock_fail_counter;
    double code_freq_chips, rem_code_phase_samples, rem_code_phase_chips, acc_carrier_phase_rad, carrier_lock_test, carr_phase_error_hz,
        carr_freq_error_hz, carr_error_filt_hz, code_error_chips, code_error_filt_chips, carrier_phase_step_rad, code_phase_step_chips,
        carrier_phase_rate_step_rad, code_phase_rate_step_chips, current_correlation_time_s;
    float rem_carr_phase_rad;
    float corr[10];        /* d_correlator_outs: VE,E,P,L,VL or E,P,L, interleaved complex */
    float prompt_data[2];  /* d_Prompt_Data[0] */
    float accu[10];        /* d_VE_accu .. d_VL_accu */
    float p_data_accu[2];
    int32_t n_events;
    int32_t events[16];
    uint64_t tow_at_current_symbol_ms;  /* Gnss_Synchro::TOW_at_current_symbol_ms of the produced item (trk.cc:2255) */
};

struct reftrk_conf_out  /* Dll_Pll_Conf after the adapter has finished with it, and what the block's constructor derived */
{
    double fs_in, carrier_lock_th, signal_carrier_freq, code_period, code_chip_rate, bs_dominance_ratio;
    float pll_bw_hz, dll_bw_hz, fll_bw_hz, pll_bw_narrow_hz, dll_bw_narrow_hz, early_late_space_chips, very_early_late_space_chips,
        early_late_space_narrow_chips, very_early_late_space_narrow_chips, slope, spc, y_intercept, cn0_smoother_alpha,
        carrier_lock_test_smoother_alpha, bs_min_prompt_mag;
    uint32_t pull_in_time_s, bit_synchronization_time_limit_s, vector_length, smoother_length;
    int32_t pll_filter_order, dll_filter_order, fll_filter_order, extend_correlation_symbols, cn0_samples, cn0_smoother_samples,
        carrier_lock_test_smoother_samples, cn0_min, max_code_lock_fail, max_carrier_lock_fail, bs_stable_best_required, bs_min_events_for_lock;
    int32_t enable_fll_pull_in, enable_fll_steady_state, track_pilot, carrier_aiding, high_dyn, bs_use_phase_dot_detector;
    int32_t code_length_chips, code_samples_per_chip, symbols_per_bit, secondary, veml, cloop, use_histogram_bit_sync, interchange_iq,
        secondary_code_length, data_secondary_code_length, correlation_length_ms, n_correlator_taps, enable_doppler_correction;
    char secondary_code[256], data_secondary_code[256];
    char system, signal[3];
};

void* reftrk_create(const char* implementation, const char* role, const char* const* keys, const char* const* values, int n_props)
{
    try
        {
            auto h = std::make_unique<Handle>();
            for (int i = 0; i < n_props; i++) h->cfg.set_property(keys[i], values[i]);
            const std::string impl(implementation);
            if (impl == "GPS_L1_CA_DLL_PLL_Tracking")
                h->adapter = std::make_shared<GpsL1CaDllPllTracking>(&h->cfg, role, 1, 1);
            else if (impl == "Galileo_E1_DLL_PLL_VEML_Tracking")
                h->adapter = std::make_shared<GalileoE1DllPllVemlTracking>(&h->cfg, role, 1, 1);
            else if (impl == "GPS_L5_DLL_PLL_Tracking")
                h->adapter = std::make_shared<GpsL5DllPllTracking>(&h->cfg, role, 1, 1);
            else if (impl == "GPS_L2_M_DLL_PLL_Tracking")
                h->adapter = std::make_shared<GpsL2MDllPllTracking>(&h->cfg, role, 1, 1);
            else if (impl == "Galileo_E5a_DLL_PLL_Tracking")
                h->adapter = std::make_shared<GalileoE5aDllPllTracking>(&h->cfg, role, 1, 1);
            else if (impl == "Galileo_E5b_DLL_PLL_Tracking")
                h->adapter = std::make_shared<GalileoE5bDllPllTracking>(&h->cfg, role, 1, 1);
            else if (impl == "Galileo_E6_DLL_PLL_Tracking")
                h->adapter = std::make_shared<GalileoE6DllPllTracking>(&h->cfg, role, 1, 1);
            else if (impl == "BEIDOU_B1I_DLL_PLL_Tracking")
                h->adapter = std::make_shared<BeidouB1iDllPllTracking>(&h->cfg, role, 1, 1);
            else if (impl == "BEIDOU_B3I_DLL_PLL_Tracking")
                h->adapter = std::make_shared<BeidouB3iDllPllTracking>(&h->cfg, role, 1, 1);
            else if (impl == "GLONASS_L1_CA_DLL_PLL_Tracking")
                h->adapter = std::make_shared<GlonassL1CaDllPllTracking>(&h->cfg, role, 1, 1);
            else if (impl == "GLONASS_L2_CA_DLL_PLL_Tracking")
                h->adapter = std::make_shared<GlonassL2CaDllPllTracking>(&h->cfg, role, 1, 1);
            else if (impl == "QZSS_L1_CA_DLL_PLL_Tracking")
                h->adapter = std::make_shared<QzssL1DllPllTracking>(&h->cfg, role, 1, 1);
            else if (impl == "QZSS_L5_DLL_PLL_Tracking")
                h->adapter = std::make_shared<QzssL5DllPllTracking>(&h->cfg, role, 1, 1);
            else
                return nullptr;
            if (h->adapter->item_size() == 0) return nullptr;  // gnss_block_factory.cc:1048-1052
            h->block = dynamic_cast<dll_pll_veml_tracking*>(h->adapter->get_left_block().get());
            if (h->block == nullptr) return nullptr;
            h->adapter->set_channel(0);
            h->adapter->set_gnss_synchro(&h->synchro);
            return h.release();
        }
    catch (const std::exception& e)
        {
            std::cerr << "reftrk_create: " << e.what() << '\n';
            return nullptr;
        }
}

/* the block on its own, configured like an adapter would but with ANY (system, signal) tag -- reaches the constructor branches no adapter of this
 * reference version selects (Galileo "E6": its adapter writes the tag "5X", galileo_e6_dll_pll_tracking.cc:62-64) */
void* reftrk_create_block(char system, const char* signal, uint32_t vector_length, const char* role, const char* const* keys, const char* const* values, int n_props)
{
    try
        {
            auto h = std::make_unique<Handle>();
            for (int i = 0; i < n_props; i++) h->cfg.set_property(keys[i], values[i]);
            Dll_Pll_Conf p;
            p.SetFromConfiguration(&h->cfg, role);
            p.system = system;
            std::memset(p.signal, 0, sizeof(p.signal));
            std::strncpy(p.signal, signal, 2);
            p.vector_length = vector_length;
            if (p.extend_correlation_symbols < 1) p.extend_correlation_symbols = 1;
            h->own_block = dll_pll_veml_make_tracking(p);
            h->block = h->own_block.get();
            h->block->set_channel(0);
            h->block->set_gnss_synchro(&h->synchro);
            return h.release();
        }
    catch (const std::exception& e)
        {
            std::cerr << "reftrk_create_block: " << e.what() << '\n';
            return nullptr;
        }
}

void reftrk_destroy(void* hv) { delete static_cast<Handle*>(hv); }

/* what acquisition leaves in the shared Gnss_Synchro (acq.cc:580-602) */
void reftrk_set_acquisition(void* hv, char system, const char* signal, uint32_t prn, double acq_delay_samples, double acq_doppler_hz,
    uint64_t acq_samplestamp_samples)
{
    auto* h = static_cast<Handle*>(hv);
    h->synchro.System = system;
    std::memset(h->synchro.Signal, 0, sizeof(h->synchro.Signal));
    std::strncpy(h->synchro.Signal, signal, 2);
    h->synchro.PRN = prn;
    h->synchro.Acq_delay_samples = acq_delay_samples;
    h->synchro.Acq_doppler_hz = acq_doppler_hz;
    h->synchro.Acq_samplestamp_samples = acq_samplestamp_samples;
}

void reftrk_start_tracking(void* hv)
{
    auto* h = static_cast<Handle*>(hv);
    if (h->adapter)
        h->adapter->start_tracking();
    else
        h->block->start_tracking();
}
// Dll_Pll_Conf::enable_doppler_correction (dll_pll_conf.h:82) has no configuration key in the reference (dll_pll_conf.cc never reads it): the
// experimental branch of trk.cc:1326-1346 can only be reached by setting the block's member, which is what the pin of its restatement does
void reftrk_set_doppler_correction(void* hv, int on)
{
    auto* h = static_cast<Handle*>(hv);
    h->block->d_trk_parameters.enable_doppler_correction = on != 0;
}
void reftrk_stop_tracking(void* hv)
{
    auto* h = static_cast<Handle*>(hv);
    if (h->adapter)
        h->adapter->stop_tracking();
    else
        h->block->stop_tracking();
}

int reftrk_forecast(void* hv, int noutput)
{
    gr_vector_int req(1, 0);
    static_cast<Handle*>(hv)->block->forecast(noutput, req);
    return req[0];
}

/* skip `n` samples as the scheduler would while the block is in standby (consumes everything it is offered, trk.cc:1905-1910) */
uint64_t reftrk_nitems_read(void* hv) { return static_cast<Handle*>(hv)->block->nitems_read(0); }

static void fill_output(Handle* h, const Gnss_Synchro* g, reftrk_output* o)
{
    std::memset(o, 0, sizeof(*o));
    auto* b = h->block;
    if (g != nullptr)
        {
            o->fs = static_cast<double>(g->fs);
            o->prompt_i = g->Prompt_I;
            o->prompt_q = g->Prompt_Q;
            o->cn0_db_hz = g->CN0_dB_hz;
            o->carrier_doppler_hz = g->Carrier_Doppler_hz;
            o->carrier_phase_rads = g->Carrier_phase_rads;
            o->code_phase_samples = g->Code_phase_samples;
            o->tracking_sample_counter = g->Tracking_sample_counter;
            o->flag_valid_symbol_output = g->Flag_valid_symbol_output ? 1 : 0;
            o->correlation_length_ms = g->correlation_length_ms;
            o->flag_pll_180_deg_phase_locked = g->Flag_PLL_180_deg_phase_locked ? 1 : 0;
            o->prn = static_cast<int32_t>(g->PRN);
            o->tow_at_current_symbol_ms = g->TOW_at_current_symbol_ms;
        }
    o->state = b->d_state;
    o->current_prn_length_samples = b->d_current_prn_length_samples;
    o->n_correlator_taps = b->d_n_correlator_taps;
    o->cn0_estimation_counter = b->d_cn0_estimation_counter;
    o->carrier_lock_fail_counter = b->d_carrier_lock_fail_counter;
    o->code_lock_fail_counter = b->d_code_lock_fail_counter;
    o->code_freq_chips = b->d_code_freq_chips;
    o->rem_code_phase_samples = b->d_rem_code_phase_samples;
    o->rem_code_phase_chips = b->d_rem_code_phase_chips;
    o->acc_carrier_phase_rad = b->d_acc_carrier_phase_rad;
    o->carrier_lock_test = b->d_carrier_lock_test;
    o->carr_phase_error_hz = b->d_carr_phase_error_hz;
    o->carr_freq_error_hz = b->d_carr_freq_error_hz;
    o->carr_error_filt_hz = b->d_carr_error_filt_hz;
    o->code_error_chips = b->d_code_error_chips;
    o->code_error_filt_chips = b->d_code_error_filt_chips;
    o->carrier_phase_step_rad = b->d_carrier_phase_step_rad;
    o->code_phase_step_chips = b->d_code_phase_step_chips;
    o->carrier_phase_rate_step_rad = b->d_carrier_phase_rate_step_rad;
    o->code_phase_rate_step_chips = b->d_code_phase_rate_step_chips;
    o->current_correlation_time_s = b->d_current_correlation_time_s;
    o->rem_carr_phase_rad = b->d_rem_carr_phase_rad;
    for (int t = 0; t < b->d_n_correlator_taps && t < 5; t++)
        {
            o->corr[2 * t] = b->d_correlator_outs[t].real();
            o->corr[2 * t + 1] = b->d_correlator_outs[t].imag();
        }
    if (!b->d_Prompt_Data.empty())
        {
            o->prompt_data[0] = b->d_Prompt_Data[0].real();
            o->prompt_data[1] = b->d_Prompt_Data[0].imag();
        }
    const gr_complex acc[5] = {b->d_VE_accu, b->d_E_accu, b->d_P_accu, b->d_L_accu, b->d_VL_accu};
    for (int t = 0; t < 5; t++)
        {
            o->accu[2 * t] = acc[t].real();
            o->accu[2 * t + 1] = acc[t].imag();
        }
    o->p_data_accu[0] = b->d_P_data_accu.real();
    o->p_data_accu[1] = b->d_P_data_accu.imag();
    for (const auto& ev : b->published)
        if (ev.first == "events" && o->n_events < 16) o->events[o->n_events++] = static_cast<int32_t>(pmt::to_long(ev.second));
}

/* One scheduler call.  iq: n_items complex64 samples available from the block's read pointer.  Returns general_work's value (number of
 * Gnss_Synchro produced, 0 or 1); *consumed = what the block passed to consume_each; out (may be NULL) = the produced item + loop state. */
int reftrk_general_work(void* hv, const float* iq, int n_items, int* consumed, reftrk_output* out)
{
    auto* h = static_cast<Handle*>(hv);
    gr_vector_int ninput{n_items};
    gr_vector_const_void_star in{static_cast<const void*>(iq)};
    h->out_items.assign(2, Gnss_Synchro{});
    gr_vector_void_star outv{static_cast<void*>(h->out_items.data())};
    h->block->consumed_last = 0;
    const int r = h->block->general_work(1, ninput, in, outv);
    h->block->mock_advance(r);
    if (consumed != nullptr) *consumed = h->block->consumed_last;
    if (out != nullptr) fill_output(h, r > 0 ? &h->out_items[0] : nullptr, out);
    return r;
}

/* Channel::set_channel after construction (the dump file is named after it, trk.cc:1851-1873) */
void reftrk_set_channel(void* hv, uint32_t channel)
{
    auto* h = static_cast<Handle*>(hv);
    if (h->adapter)
        h->adapter->set_channel(channel);
    else
        h->block->set_channel(channel);
}

/* the telemetry decoder's TOW hand-back (e.g. gps_l1_ca_telemetry_decoder_gs.cc: "telemetry_to_trk" <- shared_ptr<TOW_to_trk>), trk.cc:771-779 */
void reftrk_deliver_tow(void* hv, const char* signal, int32_t channel, uint32_t tow, uint64_t sample_stamp, int32_t wn, uint32_t prn)
{
    const std::shared_ptr<TOW_to_trk> obj = std::make_shared<TOW_to_trk>(TOW_to_trk(signal, channel, tow, sample_stamp, wn, prn));
    static_cast<Handle*>(hv)->block->deliver("telemetry_to_trk", pmt::make_any(obj));
}

/* a "timetag" stream tag on the block's input at absolute sample `offset` (signal source -> tracking, trk.cc:2256-2283) */
void reftrk_add_input_timetag(void* hv, uint64_t offset, double rx_time, int week, int tow_ms, double tow_ms_fraction)
{
    const std::shared_ptr<GnssTime> t = std::make_shared<GnssTime>();
    t->rx_time = rx_time;
    t->week = week;
    t->tow_ms = tow_ms;
    t->tow_ms_fraction = tow_ms_fraction;
    gr::tag_t tag;
    tag.offset = offset;
    tag.key = pmt::mp("timetag");
    tag.value = pmt::make_any(t);
    static_cast<Handle*>(hv)->block->input_tags.push_back(tag);
}

/* the "timetag" tags the block has attached to its output so far (trk.cc:2295-2312): returns how many (at most capacity are copied) */
int reftrk_output_timetags(void* hv, uint64_t* offsets, int* week, int* tow_ms, double* tow_ms_fraction, double* rx_time, int capacity)
{
    int n = 0;
    for (const auto& t : static_cast<Handle*>(hv)->block->output_tags)
        {
            if (pmt::symbol_to_string(t.key) != "timetag") continue;
            if (n < capacity)
                {
                    const auto g = std::any_cast<const std::shared_ptr<GnssTime>>(pmt::any_ref(t.value));
                    offsets[n] = t.offset;
                    week[n] = g->week;
                    tow_ms[n] = g->tow_ms;
                    tow_ms_fraction[n] = g->tow_ms_fraction;
                    rx_time[n] = g->rx_time;
                }
            n++;
        }
    return n;
}

void reftrk_clear_events(void* hv) { static_cast<Handle*>(hv)->block->published.clear(); }

void reftrk_get_conf(void* hv, reftrk_conf_out* c)
{
    auto* h = static_cast<Handle*>(hv);
    std::memset(c, 0, sizeof(*c));
    const auto* b = h->block;
    const Dll_Pll_Conf& p = b->d_trk_parameters;
    c->fs_in = p.fs_in;
    c->carrier_lock_th = p.carrier_lock_th;
    c->bs_dominance_ratio = p.bs_dominance_ratio;
    c->pll_bw_hz = p.pll_bw_hz;
    c->dll_bw_hz = p.dll_bw_hz;
    c->fll_bw_hz = p.fll_bw_hz;
    c->pll_bw_narrow_hz = p.pll_bw_narrow_hz;
    c->dll_bw_narrow_hz = p.dll_bw_narrow_hz;
    c->early_late_space_chips = p.early_late_space_chips;
    c->very_early_late_space_chips = p.very_early_late_space_chips;
    c->early_late_space_narrow_chips = p.early_late_space_narrow_chips;
    c->very_early_late_space_narrow_chips = p.very_early_late_space_narrow_chips;
    c->slope = p.slope;
    c->spc = p.spc;
    c->y_intercept = p.y_intercept;
    c->cn0_smoother_alpha = p.cn0_smoother_alpha;
    c->carrier_lock_test_smoother_alpha = p.carrier_lock_test_smoother_alpha;
    c->bs_min_prompt_mag = p.bs_min_prompt_mag;
    c->pull_in_time_s = p.pull_in_time_s;
    c->bit_synchronization_time_limit_s = p.bit_synchronization_time_limit_s;
    c->vector_length = p.vector_length;
    c->smoother_length = p.smoother_length;
    c->pll_filter_order = p.pll_filter_order;
    c->dll_filter_order = p.dll_filter_order;
    c->fll_filter_order = p.fll_filter_order;
    c->extend_correlation_symbols = p.extend_correlation_symbols;
    c->cn0_samples = p.cn0_samples;
    c->cn0_smoother_samples = p.cn0_smoother_samples;
    c->carrier_lock_test_smoother_samples = p.carrier_lock_test_smoother_samples;
    c->cn0_min = p.cn0_min;
    c->max_code_lock_fail = p.max_code_lock_fail;
    c->max_carrier_lock_fail = p.max_carrier_lock_fail;
    c->bs_stable_best_required = p.bs_stable_best_required;
    c->bs_min_events_for_lock = p.bs_min_events_for_lock;
    c->enable_fll_pull_in = p.enable_fll_pull_in ? 1 : 0;
    c->enable_fll_steady_state = p.enable_fll_steady_state ? 1 : 0;
    c->track_pilot = p.track_pilot ? 1 : 0;
    c->carrier_aiding = p.carrier_aiding ? 1 : 0;
    c->high_dyn = p.high_dyn ? 1 : 0;
    c->bs_use_phase_dot_detector = p.bs_use_phase_dot_detector ? 1 : 0;
    c->system = p.system;
    std::memcpy(c->signal, p.signal, 3);
    c->signal_carrier_freq = b->d_signal_carrier_freq;
    c->code_period = b->d_code_period;
    c->code_chip_rate = b->d_code_chip_rate;
    c->code_length_chips = b->d_code_length_chips;
    c->code_samples_per_chip = static_cast<int32_t>(b->d_code_samples_per_chip);
    c->symbols_per_bit = b->d_symbols_per_bit;
    c->secondary = b->d_secondary ? 1 : 0;
    c->veml = b->d_veml ? 1 : 0;
    c->cloop = b->d_cloop ? 1 : 0;
    c->use_histogram_bit_sync = b->d_use_histogram_bit_sync ? 1 : 0;
    c->interchange_iq = b->d_interchange_iq ? 1 : 0;
    c->secondary_code_length = static_cast<int32_t>(b->d_secondary_code_length);
    c->data_secondary_code_length = static_cast<int32_t>(b->d_data_secondary_code_length);
    c->correlation_length_ms = b->d_correlation_length_ms;
    c->n_correlator_taps = b->d_n_correlator_taps;
    c->enable_doppler_correction = p.enable_doppler_correction ? 1 : 0;
    std::strncpy(c->secondary_code, b->d_secondary_code_string.c_str(), 255);
    std::strncpy(c->data_secondary_code, b->d_data_secondary_code_string.c_str(), 255);
}

/* what start_tracking changes per satellite besides the members reftrk_get_conf reads (trk.cc:930-1013) */
void reftrk_get_live(void* hv, int32_t* extend_correlation_symbols, double* cfo_frequency_hz)
{
    const auto* b = static_cast<Handle*>(hv)->block;
    *extend_correlation_symbols = b->d_extend_correlation_symbols;
    *cfo_frequency_hz = b->d_cfo_frequency_hz;
}

/* the local replica(s) the block generated in start_tracking (trk.cc:796-866): code_len floats each; data may be NULL */
int reftrk_get_codes(void* hv, float* tracking_code, float* data_code, int capacity)
{
    auto* b = static_cast<Handle*>(hv)->block;
    const int n = static_cast<int>(b->d_code_length_chips * b->d_code_samples_per_chip);
    if (n > capacity) return -n;
    std::memcpy(tracking_code, b->d_tracking_code.data(), sizeof(float) * n);
    if (data_code != nullptr && b->d_trk_parameters.track_pilot) std::memcpy(data_code, b->d_data_code.data(), sizeof(float) * n);
    return n;
}
}
