/*
 * TEST INFRASTRUCTURE ONLY (oracle/_ref build) -- never linked into the product.
 *
 * C driver over the REFERENCE's own acquisition blocks, compiled from where they lie under /root/reference
 * (src/algorithms/acquisition/gnuradio_blocks/{pcps_acquisition, pcps_tong_acquisition_cc, galileo_pcps_8ms_acquisition_cc,
 * pcps_cccwsr_acquisition_cc, pcps_quicksync_acquisition_cc, pcps_acquisition_fine_doppler_cc,
 * galileo_e5a_noncoherent_iq_acquisition_caf_cc}.cc + acquisition/libs/acq_conf.cc), against stand-ins for the libraries this
 * image lacks: tests/host/mock_gnuradio (gr::block without a scheduler), oracle/shim_blocks (gr::fft, VOLK element-wise loops,
 * Armadillo dump matrix, matio, Boost gamma_p_inv).  Every line of block logic -- buffer sizes, zero padding, wipe-off tables,
 * the Doppler loop, both statistics, thresholds, dwell counting, the two-step machine, Gnss_Synchro filling, message ports --
 * is the reference's translation unit; the only arithmetic that is not the reference's is the transform (see ref_fft.cc).
 *
 * The blocks are driven the way GNU Radio's scheduler drives them: general_work(noutput, ninput_items, input_items, output_items)
 * on a caller-supplied chunk; `consume_each` is recorded by the mock and returned.  Configuration goes through the reference's own
 * InMemoryConfiguration + Acq_Conf::SetFromConfiguration exactly as the adapters do it
 * (src/algorithms/acquisition/adapters/base_pcps_acquisition.cc:40-49).
 *
 * Private block state is read for comparison through `#define private public` around the block headers in THIS translation unit
 * only (the blocks themselves are compiled untouched).
 */
#include <algorithm>
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
#include <queue>
#include <span>
#include <sstream>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include <gnuradio/block.h>
#include <gnuradio/fft/fft.h>
#include <volk_gnsssdr/volk_gnsssdr_alloc.h>
#include <armadillo>

#include "acq_conf.h"
#include "channel_fsm.h"
#include "gnss_synchro.h"
#include "in_memory_configuration.h"

#define private public
#include "galileo_e5a_noncoherent_iq_acquisition_caf_cc.h"
#include "galileo_pcps_8ms_acquisition_cc.h"
#include "pcps_acquisition.h"
#include "pcps_acquisition_fine_doppler_cc.h"
#include "pcps_cccwsr_acquisition_cc.h"
#include "pcps_quicksync_acquisition_cc.h"
#include "pcps_tong_acquisition_cc.h"
#undef private

// pcps_acquisition.cc:322-326 notifies the channel FSM directly when one is set: the reference's own channel_fsm.cc / channel_event.cc are part of this library
// (oracle/Makefile, _ref/chan_channel_fsm.o) -- until round 4 a stub stood here.  The driver below never sets an FSM (the block then publishes on "events").

namespace
{
enum Kind
{
    K_PCPS = 0,
    K_TONG = 1,
    K_8MS = 2,
    K_CCCWSR = 3,
    K_QUICKSYNC = 4,
    K_FINE_DOPPLER = 5,
    K_E5A_CAF = 6
};

struct Handle
{
    int kind{0};
    Acq_Conf conf;
    Gnss_Synchro synchro{};
    acquisition_impl_interface_sptr block;
    std::vector<Gnss_Synchro> monitor_out;
};
}  // namespace

extern "C" {

/* what the generated volk_gnsssdr library would provide (volk_gnsssdr::vector's allocator calls these) */
size_t volk_gnsssdr_get_alignment(void) { return 32; }
void* volk_gnsssdr_malloc(size_t size, size_t alignment)
{
    void* p = nullptr;
    if (alignment < sizeof(void*)) alignment = sizeof(void*);
    if (posix_memalign(&p, alignment, size ? size : alignment) != 0) return nullptr;
    return p;
}
void volk_gnsssdr_free(void* p) { std::free(p); }

struct refacq_status
{
    int32_t state, active, step_two, positive_acq;
    uint32_t dwell_count, tong_count, num_doppler_bins, fft_size, effective_fft_size, consumed_samples, code_phase, doppler_bins_step2;
    uint64_t sample_counter;
    float mag, input_power, test_statistics, threshold, threshold_step_two, doppler_center_step_two;
    // Gnss_Synchro fields the acquisition writes (gnss_synchro.h:46-82)
    double acq_delay_samples, acq_doppler_hz;
    uint64_t acq_samplestamp_samples;
    uint32_t acq_doppler_step;
    int64_t fs;
    // Acq_Conf after SetFromConfiguration (acq_conf.cc:29-124)
    int64_t conf_fs_in, conf_resampled_fs;
    float conf_samples_per_ms, conf_samples_per_code, conf_resampler_ratio, conf_threshold, conf_pfa, conf_pfa2, conf_doppler_step2;
    uint32_t conf_samples_per_chip, conf_doppler_max, conf_doppler_step, conf_sampled_ms, conf_ms_per_code, conf_max_dwells, conf_num_doppler_bins_step2;
    int32_t conf_it_size, conf_use_cfar, conf_bit_transition_flag, conf_make_2_steps, conf_blocking, conf_use_automatic_resampler;
    // what the block did with the GNU Radio runtime in the last call
    int32_t consumed_last;
    int64_t consumed_total;
    int32_t n_events;
    int32_t events[32];
};

/* keys/values: n_props pairs of C strings, e.g. "GNSS-SDR.internal_fs_sps" -> "25000000", "Acquisition_1C.doppler_max" -> "5000".
 * extra[]: kind-specific constructor arguments (Tong: init, max, max_dwells; QuickSync: folding_factor, max_dwells;
 * E5a: both_signal_components, CAF_window_hz, Zero_padding).  Returns nullptr when the reference throws. */
void* refacq_create(int kind, const char* role, const char* const* keys, const char* const* values, int n_props, double chip_rate,
    double opt_freq, uint32_t ms_per_code, const int32_t* extra)
{
    try
        {
            auto h = std::make_unique<Handle>();
            h->kind = kind;
            InMemoryConfiguration cfg;
            for (int i = 0; i < n_props; i++) cfg.set_property(keys[i], values[i]);
            h->conf.ms_per_code = ms_per_code;  // base_pcps_acquisition.cc:43-45
            h->conf.sampled_ms = ms_per_code;
            h->conf.SetFromConfiguration(&cfg, role, chip_rate, opt_freq);
            switch (kind)
                {
                case K_PCPS:
                    h->block = pcps_make_acquisition(h->conf);
                    break;
                case K_TONG:
                    h->block = pcps_tong_make_acquisition_cc(h->conf, extra[0], extra[1], extra[2]);
                    break;
                case K_8MS:
                    h->block = galileo_pcps_8ms_make_acquisition_cc(h->conf);
                    break;
                case K_CCCWSR:
                    h->block = pcps_cccwsr_make_acquisition_cc(h->conf);
                    break;
                case K_QUICKSYNC:
                    h->block = pcps_quicksync_make_acquisition_cc(h->conf, extra[0], extra[1]);
                    break;
                case K_FINE_DOPPLER:
                    h->block = pcps_make_acquisition_fine_doppler_cc(h->conf);
                    break;
                case K_E5A_CAF:
                    h->block = galileo_e5a_noncoherentIQ_make_acquisition_caf_cc(h->conf, extra[0] != 0, extra[1], extra[2]);
                    break;
                default:
                    return nullptr;
                }
            h->block->set_gnss_synchro(&h->synchro);
            return h.release();
        }
    catch (const std::exception& e)
        {
            std::cerr << "refacq_create: " << e.what() << '\n';
            return nullptr;
        }
}

void refacq_destroy(void* hv) { delete static_cast<Handle*>(hv); }

/* Direct write of the (already derived) Acq_Conf members a detector adapter overrides after SetFromConfiguration
 * (e.g. pcps_tong/quicksync adapters set doppler_step, samples_per_ms ... by hand) -- must be called BEFORE create; so this is
 * offered as a second constructor path: same as refacq_create but with a POD override block applied to Acq_Conf first. */
struct refacq_override
{
    int32_t has_samples_per_ms, has_samples_per_code, has_samples_per_chip, has_sampled_ms, has_threshold, has_doppler_step, has_doppler_max, has_max_dwells,
        has_bit_transition_flag, has_dump, has_code_length, has_vector_length, has_num_codes;
    float samples_per_ms, samples_per_code, threshold;
    uint32_t samples_per_chip, sampled_ms, doppler_step, doppler_max, max_dwells;
    int32_t bit_transition_flag, dump;
    uint32_t code_length, vector_length, num_codes;  // base_pcps_acquisition_custom.cc:78-80 ("not part of the configuration interface")
};

void* refacq_create_with_override(int kind, const char* role, const char* const* keys, const char* const* values, int n_props, double chip_rate,
    double opt_freq, uint32_t ms_per_code, const int32_t* extra, const refacq_override* ov)
{
    try
        {
            auto h = std::make_unique<Handle>();
            h->kind = kind;
            InMemoryConfiguration cfg;
            for (int i = 0; i < n_props; i++) cfg.set_property(keys[i], values[i]);
            h->conf.ms_per_code = ms_per_code;
            h->conf.sampled_ms = ms_per_code;
            h->conf.SetFromConfiguration(&cfg, role, chip_rate, opt_freq);
            if (ov->has_samples_per_ms) h->conf.samples_per_ms = ov->samples_per_ms;
            if (ov->has_samples_per_code) h->conf.samples_per_code = ov->samples_per_code;
            if (ov->has_samples_per_chip) h->conf.samples_per_chip = ov->samples_per_chip;
            if (ov->has_sampled_ms) h->conf.sampled_ms = ov->sampled_ms;
            if (ov->has_threshold) h->conf.threshold = ov->threshold;
            if (ov->has_doppler_step) h->conf.doppler_step = ov->doppler_step;
            if (ov->has_doppler_max) h->conf.doppler_max = ov->doppler_max;
            if (ov->has_max_dwells) h->conf.max_dwells = ov->max_dwells;
            if (ov->has_bit_transition_flag) h->conf.bit_transition_flag = ov->bit_transition_flag != 0;
            if (ov->has_dump) h->conf.dump = ov->dump != 0;
            if (ov->has_code_length) h->conf.code_length = ov->code_length;
            if (ov->has_vector_length) h->conf.vector_length = ov->vector_length;
            if (ov->has_num_codes) h->conf.num_codes = ov->num_codes;
            switch (kind)
                {
                case K_PCPS:
                    h->block = pcps_make_acquisition(h->conf);
                    break;
                case K_TONG:
                    h->block = pcps_tong_make_acquisition_cc(h->conf, extra[0], extra[1], extra[2]);
                    break;
                case K_8MS:
                    h->block = galileo_pcps_8ms_make_acquisition_cc(h->conf);
                    break;
                case K_CCCWSR:
                    h->block = pcps_cccwsr_make_acquisition_cc(h->conf);
                    break;
                case K_QUICKSYNC:
                    h->block = pcps_quicksync_make_acquisition_cc(h->conf, extra[0], extra[1]);
                    break;
                case K_FINE_DOPPLER:
                    h->block = pcps_make_acquisition_fine_doppler_cc(h->conf);
                    break;
                case K_E5A_CAF:
                    h->block = galileo_e5a_noncoherentIQ_make_acquisition_caf_cc(h->conf, extra[0] != 0, extra[1], extra[2]);
                    break;
                default:
                    return nullptr;
                }
            h->block->set_gnss_synchro(&h->synchro);
            return h.release();
        }
    catch (const std::exception& e)
        {
            std::cerr << "refacq_create: " << e.what() << '\n';
            return nullptr;
        }
}

void refacq_set_satellite(void* hv, char system, const char* signal, uint32_t prn)
{
    auto* h = static_cast<Handle*>(hv);
    h->synchro.System = system;
    std::memset(h->synchro.Signal, 0, sizeof(h->synchro.Signal));
    std::strncpy(h->synchro.Signal, signal, 2);
    h->synchro.PRN = prn;
}

void refacq_set_channel(void* hv, uint32_t ch) { static_cast<Handle*>(hv)->block->set_channel(ch); }

/* code / code2: interleaved complex64; code2 only for the two-code blocks (CCCWSR data+pilot, E5a I+Q) */
void refacq_set_local_code(void* hv, const float* code, const float* code2)
{
    auto* h = static_cast<Handle*>(hv);
    auto* c1 = reinterpret_cast<std::complex<float>*>(const_cast<float*>(code));
    auto* c2 = reinterpret_cast<std::complex<float>*>(const_cast<float*>(code2));
    if (h->kind == K_CCCWSR || h->kind == K_E5A_CAF)
        h->block->set_local_code(c1, c2);
    else
        h->block->set_local_code(c1);
}

void refacq_set_active(void* hv, int active) { static_cast<Handle*>(hv)->block->set_active(active != 0); }

int refacq_set_doppler_center(void* hv, int32_t center)
{
    auto* h = static_cast<Handle*>(hv);
    if (h->kind != K_PCPS) return -1;
    dynamic_cast<pcps_acquisition*>(h->block.get())->set_doppler_center(center);
    return 0;
}

int refacq_set_resampler_latency(void* hv, uint32_t samples)
{
    auto* h = static_cast<Handle*>(hv);
    if (h->kind != K_PCPS) return -1;
    dynamic_cast<pcps_acquisition*>(h->block.get())->set_resampler_latency(samples);
    return 0;
}

/* One scheduler call: the block sees `n_items` available input items at `items`.  Returns general_work's return value;
 * *consumed = what it passed to consume_each during the call. */
int refacq_general_work(void* hv, const void* items, int n_items, int noutput_items, int* consumed)
{
    auto* h = static_cast<Handle*>(hv);
    gr_vector_int ninput{n_items};
    gr_vector_const_void_star in{items};
    h->monitor_out.assign(8, Gnss_Synchro{});
    std::vector<Gnss_Synchro*> outp;
    for (auto& g : h->monitor_out) outp.push_back(&g);
    gr_vector_void_star out{static_cast<void*>(outp.data())};
    // pcps_acquisition.cc:826 indexes output_items as an array of Gnss_Synchro*: hand it one
    gr_vector_void_star out_items;
    for (auto* p : outp) out_items.push_back(p);
    h->block->consumed_last = 0;
    const int r = h->block->general_work(noutput_items, ninput, in, out_items);
    if (consumed != nullptr) *consumed = h->block->consumed_last;
    return r;
}

int refacq_forecast(void* hv, int noutput)
{
    auto* h = static_cast<Handle*>(hv);
    gr_vector_int req(1, 0);
    h->block->forecast(noutput, req);
    return req[0];
}

#define COMMON_DETECTOR_FIELDS(B, DWELL)                      \
    st->state = B->d_state;                                   \
    st->active = B->d_active ? 1 : 0;                         \
    st->mag = B->d_mag;                                       \
    st->input_power = B->d_input_power;                       \
    st->test_statistics = B->d_test_statistics;               \
    st->sample_counter = B->d_sample_counter;                 \
    st->dwell_count = B->DWELL;                               \
    st->num_doppler_bins = B->d_num_doppler_bins;             \
    st->fft_size = B->d_fft_size;                             \
    st->code_phase = B->d_code_phase;

void refacq_get_status(void* hv, refacq_status* st)
{
    auto* h = static_cast<Handle*>(hv);
    std::memset(st, 0, sizeof(*st));
    switch (h->kind)
        {
        case K_PCPS:
            {
                auto* b = dynamic_cast<pcps_acquisition*>(h->block.get());
                b->wait_if_active();
                st->state = b->d_state;
                st->active = b->d_active ? 1 : 0;
                st->step_two = b->d_step_two ? 1 : 0;
                st->dwell_count = b->d_num_noncoherent_integrations_counter;
                st->num_doppler_bins = b->d_num_doppler_bins;
                st->doppler_bins_step2 = b->d_num_doppler_bins_step2;
                st->fft_size = b->d_fft_size;
                st->effective_fft_size = b->d_effective_fft_size;
                st->consumed_samples = b->d_consumed_samples;
                st->sample_counter = b->d_sample_count;
                st->input_power = b->d_input_power;
                st->threshold = b->d_threshold;
                st->threshold_step_two = b->d_threshold_step_two;
                st->doppler_center_step_two = b->d_doppler_center_step_two;
                break;
            }
        case K_TONG:
            {
                auto* b = dynamic_cast<pcps_tong_acquisition_cc*>(h->block.get());
                COMMON_DETECTOR_FIELDS(b, d_dwell_count)
                st->tong_count = b->d_tong_count;
                break;
            }
        case K_8MS:
            {
                auto* b = dynamic_cast<galileo_pcps_8ms_acquisition_cc*>(h->block.get());
                COMMON_DETECTOR_FIELDS(b, d_well_count)
                break;
            }
        case K_CCCWSR:
            {
                auto* b = dynamic_cast<pcps_cccwsr_acquisition_cc*>(h->block.get());
                COMMON_DETECTOR_FIELDS(b, d_well_count)
                break;
            }
        case K_QUICKSYNC:
            {
                auto* b = dynamic_cast<pcps_quicksync_acquisition_cc*>(h->block.get());
                COMMON_DETECTOR_FIELDS(b, d_well_count)
                break;
            }
        case K_FINE_DOPPLER:
            {
                auto* b = dynamic_cast<pcps_acquisition_fine_doppler_cc*>(h->block.get());
                st->state = b->d_state;
                st->active = b->d_active ? 1 : 0;
                st->positive_acq = b->d_positive_acq;
                st->test_statistics = b->d_test_statistics;
                st->sample_counter = b->d_sample_counter;
                st->dwell_count = b->d_well_count;
                st->num_doppler_bins = b->d_num_doppler_points;
                st->fft_size = b->d_fft_size;
                break;
            }
        case K_E5A_CAF:
            {
                auto* b = dynamic_cast<galileo_e5a_noncoherentIQ_acquisition_caf_cc*>(h->block.get());
                COMMON_DETECTOR_FIELDS(b, d_well_count)
                break;
            }
        }
    st->acq_delay_samples = h->synchro.Acq_delay_samples;
    st->acq_doppler_hz = h->synchro.Acq_doppler_hz;
    st->acq_samplestamp_samples = h->synchro.Acq_samplestamp_samples;
    st->acq_doppler_step = h->synchro.Acq_doppler_step;
    st->fs = h->synchro.fs;
    const Acq_Conf& c = h->conf;
    st->conf_fs_in = c.fs_in;
    st->conf_resampled_fs = c.resampled_fs;
    st->conf_samples_per_ms = c.samples_per_ms;
    st->conf_samples_per_code = c.samples_per_code;
    st->conf_resampler_ratio = c.resampler_ratio;
    st->conf_threshold = c.threshold;
    st->conf_pfa = c.pfa;
    st->conf_pfa2 = c.pfa2;
    st->conf_doppler_step2 = c.doppler_step2;
    st->conf_samples_per_chip = c.samples_per_chip;
    st->conf_doppler_max = c.doppler_max;
    st->conf_doppler_step = c.doppler_step;
    st->conf_sampled_ms = c.sampled_ms;
    st->conf_ms_per_code = c.ms_per_code;
    st->conf_max_dwells = c.max_dwells;
    st->conf_num_doppler_bins_step2 = c.num_doppler_bins_step2;
    st->conf_it_size = static_cast<int32_t>(c.it_size);
    st->conf_use_cfar = c.use_CFAR_algorithm_flag ? 1 : 0;
    st->conf_bit_transition_flag = c.bit_transition_flag ? 1 : 0;
    st->conf_make_2_steps = c.make_2_steps ? 1 : 0;
    st->conf_blocking = c.blocking ? 1 : 0;
    st->conf_use_automatic_resampler = c.use_automatic_resampler ? 1 : 0;
    st->consumed_last = h->block->consumed_last;
    st->consumed_total = h->block->consumed_total;
    st->n_events = 0;
    for (const auto& ev : h->block->published)
        {
            if (ev.first == "events" && st->n_events < 32) st->events[st->n_events++] = static_cast<int32_t>(pmt::to_long(ev.second));
        }
}

void refacq_clear_events(void* hv) { static_cast<Handle*>(hv)->block->published.clear(); }

/* pcps_acquisition: rows of d_magnitude_grid ([bin][0..effective)), the conjugated code spectrum, one wipe-off table.
 * Tong / fine Doppler: d_grid_data.  Returns the number of floats written (0 if the block keeps no such array). */
int64_t refacq_read_grid(void* hv, float* dst, int64_t capacity)
{
    auto* h = static_cast<Handle*>(hv);
    int64_t n = 0;
    auto put_rows = [&](const auto& rows, size_t bins, size_t len) {
        for (size_t d = 0; d < bins; d++)
            for (size_t i = 0; i < len && n < capacity; i++) dst[n++] = rows[d][i];
    };
    if (h->kind == K_PCPS)
        {
            auto* b = dynamic_cast<pcps_acquisition*>(h->block.get());
            b->wait_if_active();
            const size_t bins = b->d_step_two ? b->d_num_doppler_bins_step2 : b->d_num_doppler_bins;
            put_rows(b->d_magnitude_grid, std::min<size_t>(bins, b->d_magnitude_grid.size()), b->d_effective_fft_size);
        }
    else if (h->kind == K_TONG)
        {
            auto* b = dynamic_cast<pcps_tong_acquisition_cc*>(h->block.get());
            put_rows(b->d_grid_data, b->d_grid_data.size(), b->d_fft_size);
        }
    else if (h->kind == K_FINE_DOPPLER)
        {
            auto* b = dynamic_cast<pcps_acquisition_fine_doppler_cc*>(h->block.get());
            put_rows(b->d_grid_data, b->d_grid_data.size(), b->d_fft_size);
        }
    return n;
}

int64_t refacq_read_fft_codes(void* hv, float* dst_iq, int64_t capacity_complex)
{
    auto* h = static_cast<Handle*>(hv);
    if (h->kind != K_PCPS) return 0;
    auto* b = dynamic_cast<pcps_acquisition*>(h->block.get());
    const int64_t n = std::min<int64_t>(capacity_complex, b->d_fft_codes.size());
    std::memcpy(dst_iq, b->d_fft_codes.data(), n * sizeof(gr_complex));
    return n;
}

int64_t refacq_read_wipeoff(void* hv, uint32_t bin, int step_two, float* dst_iq, int64_t capacity_complex)
{
    auto* h = static_cast<Handle*>(hv);
    if (h->kind != K_PCPS) return 0;
    auto* b = dynamic_cast<pcps_acquisition*>(h->block.get());
    const auto& tab = step_two ? b->d_grid_doppler_wipeoffs_step_two : b->d_grid_doppler_wipeoffs;
    if (bin >= tab.size()) return 0;
    const int64_t n = std::min<int64_t>(capacity_complex, tab[bin].size());
    std::memcpy(dst_iq, tab[bin].data(), n * sizeof(gr_complex));
    return n;
}

/* the anonymous-namespace compute_threshold of pcps_acquisition.cc:52-56 is reachable through the constructor: a block with
 * pfa > 0 stores it in d_threshold (read it with refacq_get_status). */
}
