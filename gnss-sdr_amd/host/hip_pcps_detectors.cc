/*!
 * \file hip_pcps_detectors.cc
 * \brief See the header.
 */
#include "hip_pcps_detectors.h"
#include "gnss_sdr_hip.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace
{
uint32_t count_bins(int32_t doppler_max, int32_t doppler_step)
{
    uint32_t n = 0;  // tong.cc:97-100, 8ms.cc:65-68: inclusive of +doppler_max
    for (int32_t doppler = -doppler_max; doppler <= doppler_max; doppler += doppler_step) n++;
    return n;
}

gsh_acq* make_handle(const Hip_Acq_Conf& conf, uint32_t fft_size, uint32_t bins, uint32_t max_prn, int device, std::string* err, uint32_t fold = 0)
{
    gsh_acq_conf c{};
    c.fs_in = conf.fs_in;
    c.fft_size = fft_size;
    c.effective_fft_size = fft_size;
    c.consumed_samples = fold > 1 ? fold * fft_size : fft_size;
    c.fold = fold;
    c.num_doppler_bins = bins;
    c.doppler_max = conf.doppler_max;
    c.doppler_step = conf.doppler_step;
    c.samples_per_chip = conf.samples_per_chip;
    c.samples_per_code = conf.samples_per_code;
    c.use_cfar = 1;
    c.max_prn = max_prn;
    c.no_grid = 0;
    gsh_acq* h = nullptr;
    if (gsh_acq_create(device, &c, &h) != GSH_OK)
        {
            *err = gsh_last_error();
            return nullptr;
        }
    return h;
}
}  // namespace


float hip_threshold_compute_doppler(float pfa, uint32_t vector_length, int32_t doppler_max, int32_t doppler_step)
{
    const uint32_t frequency_bins = count_bins(doppler_max, doppler_step);
    const auto ncells = vector_length * frequency_bins;
    const auto exponent = 1 / static_cast<double>(ncells);
    const auto val = std::pow(1.0 - pfa, exponent);
    const auto lambda = static_cast<double>(vector_length);
    return static_cast<float>(-std::log1p(-val) / lambda);  // quantile of the exponential distribution
}


float hip_threshold_compute_quicksync(float pfa, uint32_t code_length, uint32_t folding_factor, int32_t doppler_max, int32_t doppler_step)
{
    const uint32_t frequency_bins = count_bins(doppler_max, doppler_step);
    const auto ncells = (code_length / folding_factor) * frequency_bins;
    const auto exponent = 1.0 / static_cast<double>(ncells);
    const auto val = std::pow(1.0 - pfa, exponent);
    const auto lambda = static_cast<double>(code_length) / static_cast<double>(folding_factor);
    return static_cast<float>(-std::log1p(-val) / lambda);
}


// ---------------------------------------------------------------------------------------------------------------- Tong
Hip_Pcps_Tong_Core::Hip_Pcps_Tong_Core(const Hip_Acq_Conf& conf, uint32_t tong_init_val, uint32_t tong_max_val, uint32_t tong_max_dwells, int device)
    : d_acq_params(conf), d_tong_init_val(tong_init_val), d_tong_max_val(tong_max_val), d_tong_max_dwells(tong_max_dwells), d_tong_count(tong_init_val)
{
    d_fft_size = static_cast<uint32_t>(conf.sampled_ms * conf.samples_per_ms);  // tong.cc:85
    d_num_doppler_bins = count_bins(conf.doppler_max, conf.doppler_step);
    d_handle = make_handle(conf, d_fft_size, d_num_doppler_bins, 1, device, &d_error);
}


Hip_Pcps_Tong_Core::~Hip_Pcps_Tong_Core()
{
    if (d_handle != nullptr) gsh_acq_destroy(d_handle);
}


void Hip_Pcps_Tong_Core::set_local_code(const std::complex<float>* code)
{
    if (d_handle == nullptr) return;
    if (gsh_acq_set_local_code(d_handle, 0, reinterpret_cast<const float*>(code)) != GSH_OK) d_error = gsh_last_error();
}


void Hip_Pcps_Tong_Core::init()
{
    d_result = Hip_Detector_Result();
    d_dwell_count = 0;
    d_tong_count = d_tong_init_val;
    d_mag = 0.0;
    d_input_power = 0.0;
    d_test_statistics = 0.0;
    // d_grid_data is cleared by the first dwell overwriting it (accumulate = 0)
    d_state = 1;
}


int Hip_Pcps_Tong_Core::work(uint64_t sample_counter, const std::complex<float>* in)
{
    if (d_handle == nullptr) return -1;
    const float fft_normalization_factor = static_cast<float>(d_fft_size) * static_cast<float>(d_fft_size);
    d_input_power = 0.0;
    d_mag = 0.0;
    d_dwell_count++;

    // 1- input signal power estimation of THIS block (tong.cc:208-210): it scales this block's magnitudes
    if (gsh_acq_stage_input(d_handle, reinterpret_cast<const float*>(in)) != GSH_OK || gsh_acq_input_power(d_handle, &d_input_power) != GSH_OK)
        {
            d_error = gsh_last_error();
            return -1;
        }
    // 2..4- Doppler loop: |y|^2 / (norm^2 P) added to d_grid_data, per-bin maxima, first strictly greater bin wins (tong.cc:213-264)
    const float weight = 1 / (fft_normalization_factor * fft_normalization_factor * d_input_power);
    gsh_acq_result r{};
    if (gsh_acq_set_grid_weight(d_handle, weight) != GSH_OK || gsh_acq_dwell_resident(d_handle, 1, d_dwell_count > 1 ? 1 : 0, d_dwell_count, &r) != GSH_OK)
        {
            d_error = gsh_last_error();
            return -1;
        }
    if (d_mag < r.peak)
        {
            d_mag = r.peak;
            d_result.index_time = r.index_time;
            d_result.index_doppler = r.index_doppler;
            d_result.Acq_delay_samples = static_cast<double>(r.index_time % static_cast<int32_t>(d_acq_params.samples_per_code));
            d_result.Acq_doppler_hz = static_cast<double>(-d_acq_params.doppler_max + d_acq_params.doppler_step * static_cast<int32_t>(r.index_doppler));
            d_result.Acq_samplestamp_samples = sample_counter;
            d_result.Acq_doppler_step = static_cast<uint32_t>(d_acq_params.doppler_step);
        }

    // 5- test statistics against the threshold (tong.cc:277-299)
    d_test_statistics = d_mag;
    if (d_test_statistics > d_acq_params.threshold * d_dwell_count)
        {
            d_tong_count++;
            if (d_tong_count == d_tong_max_val) d_state = 2;  // Positive acquisition
        }
    else
        {
            d_tong_count--;
            if (d_tong_count == 0) d_state = 3;  // Negative acquisition
        }
    if (d_dwell_count >= d_tong_max_dwells) d_state = 3;  // Negative acquisition
    return d_state;
}


// ---------------------------------------------------------------------------------------------------------------- 8 ms
Hip_Galileo_Pcps_8ms_Core::Hip_Galileo_Pcps_8ms_Core(const Hip_Acq_Conf& conf, int device) : d_acq_params(conf)
{
    d_fft_size = static_cast<uint32_t>(conf.sampled_ms * conf.samples_per_ms);  // 8ms.cc:53
    d_num_doppler_bins = count_bins(conf.doppler_max, conf.doppler_step);
    d_handle = make_handle(conf, d_fft_size, d_num_doppler_bins, 2, device, &d_error);  // slot 0: code A, slot 1: code B
    d_peak_a.resize(d_num_doppler_bins);
    d_peak_b.resize(d_num_doppler_bins);
    d_index_a.resize(d_num_doppler_bins);
    d_index_b.resize(d_num_doppler_bins);
    d_code_b.resize(d_fft_size);
}


Hip_Galileo_Pcps_8ms_Core::~Hip_Galileo_Pcps_8ms_Core()
{
    if (d_handle != nullptr) gsh_acq_destroy(d_handle);
}


void Hip_Galileo_Pcps_8ms_Core::set_local_code(const std::complex<float>* code)
{
    if (d_handle == nullptr) return;
    // code A: two replicas of a primary code
    if (gsh_acq_set_local_code(d_handle, 0, reinterpret_cast<const float*>(code)) != GSH_OK) d_error = gsh_last_error();
    // code B: two replicas of a primary code; the second replica is inverted
    const auto samples_per_code = static_cast<uint32_t>(d_acq_params.samples_per_code);
    for (uint32_t i = 0; i < d_fft_size; i++) d_code_b[i] = code[i];
    for (uint32_t i = samples_per_code; i < 2 * samples_per_code && i < d_fft_size; i++) d_code_b[i] = code[i] * std::complex<float>(-1, 0);
    if (gsh_acq_set_local_code(d_handle, 1, reinterpret_cast<const float*>(d_code_b.data())) != GSH_OK) d_error = gsh_last_error();
}


void Hip_Galileo_Pcps_8ms_Core::init()
{
    d_result = Hip_Detector_Result();
    d_well_count = 0;
    d_mag = 0.0;
    d_input_power = 0.0;
    d_test_statistics = 0.0;
    d_state = 1;
}


int Hip_Galileo_Pcps_8ms_Core::work(uint64_t sample_counter, const std::complex<float>* in)
{
    if (d_handle == nullptr) return -1;
    const float fft_normalization_factor = static_cast<float>(d_fft_size) * static_cast<float>(d_fft_size);
    d_input_power = 0.0;
    d_mag = 0.0;
    d_well_count++;

    // one dwell searches both local codes over shared forward transforms (8ms.cc:195-237 does two inverse FFTs per bin)
    gsh_acq_result r[2]{};
    if (gsh_acq_dwell(d_handle, reinterpret_cast<const float*>(in), 2, 0, 1, r) != GSH_OK || gsh_acq_input_power(d_handle, &d_input_power) != GSH_OK ||
        gsh_acq_read_row_peaks(d_handle, 0, d_peak_a.data(), d_index_a.data()) != GSH_OK ||
        gsh_acq_read_row_peaks(d_handle, 1, d_peak_b.data(), d_index_b.data()) != GSH_OK)
        {
            d_error = gsh_last_error();
            return -1;
        }
    for (uint32_t doppler_index = 0; doppler_index < d_num_doppler_bins; doppler_index++)
        {
            const int32_t doppler = -d_acq_params.doppler_max + d_acq_params.doppler_step * static_cast<int32_t>(doppler_index);
            // normalise the maxima (8ms.cc:222, :237), take the greater (:240-249), first strictly greater bin wins (:252)
            const float magt_A = d_peak_a[doppler_index] / (fft_normalization_factor * fft_normalization_factor);
            const float magt_B = d_peak_b[doppler_index] / (fft_normalization_factor * fft_normalization_factor);
            float magt;
            uint32_t indext;
            int which;
            if (magt_A >= magt_B)
                {
                    magt = magt_A;
                    indext = d_index_a[doppler_index];
                    which = 0;
                }
            else
                {
                    magt = magt_B;
                    indext = d_index_b[doppler_index];
                    which = 1;
                }
            if (d_mag < magt)
                {
                    d_mag = magt;
                    d_winning_code = which;
                    d_result.index_time = indext;
                    d_result.index_doppler = doppler_index;
                    d_result.Acq_delay_samples = static_cast<double>(indext % static_cast<int32_t>(d_acq_params.samples_per_code));
                    d_result.Acq_doppler_hz = static_cast<double>(doppler);
                    d_result.Acq_samplestamp_samples = sample_counter;
                    d_result.Acq_doppler_step = static_cast<uint32_t>(d_acq_params.doppler_step);
                }
        }

    // 5- test statistics against the threshold (8ms.cc:278-287)
    d_test_statistics = d_mag / d_input_power;
    if (d_test_statistics > d_acq_params.threshold)
        {
            d_state = 2;  // Positive acquisition
        }
    else if (d_well_count == d_acq_params.max_dwells)
        {
            d_state = 3;  // Negative acquisition
        }
    return d_state;
}


// ---------------------------------------------------------------------------------------------------------------- CCCWSR
Hip_Pcps_Cccwsr_Core::Hip_Pcps_Cccwsr_Core(const Hip_Acq_Conf& conf, int device) : d_acq_params(conf)
{
    d_fft_size = static_cast<uint32_t>(conf.sampled_ms * conf.samples_per_ms);  // cccwsr.cc:61
    d_num_doppler_bins = count_bins(conf.doppler_max, conf.doppler_step);       // cccwsr.cc:79-82
    d_handle = make_handle(conf, d_fft_size, d_num_doppler_bins, 2, device, &d_error);  // slot 0: data - j pilot, slot 1: data + j pilot
    d_peak_plus.resize(d_num_doppler_bins);
    d_peak_minus.resize(d_num_doppler_bins);
    d_index_plus.resize(d_num_doppler_bins);
    d_index_minus.resize(d_num_doppler_bins);
    d_code_combined.resize(d_fft_size);
}


Hip_Pcps_Cccwsr_Core::~Hip_Pcps_Cccwsr_Core()
{
    if (d_handle != nullptr) gsh_acq_destroy(d_handle);
}


void Hip_Pcps_Cccwsr_Core::set_local_code(const std::complex<float>* code_data, const std::complex<float>* code_pilot)
{
    if (d_handle == nullptr) return;
    // correlating with (data - j pilot) yields data_corr + j pilot_corr, the block's d_correlation_plus (cccwsr.cc:237-239)
    for (uint32_t i = 0; i < d_fft_size; i++)
        {
            d_code_combined[i] = std::complex<float>(code_data[i].real() + code_pilot[i].imag(), code_data[i].imag() - code_pilot[i].real());
        }
    if (gsh_acq_set_local_code(d_handle, 0, reinterpret_cast<const float*>(d_code_combined.data())) != GSH_OK) d_error = gsh_last_error();
    // correlating with (data + j pilot) yields data_corr - j pilot_corr, d_correlation_minus (cccwsr.cc:241-243)
    for (uint32_t i = 0; i < d_fft_size; i++)
        {
            d_code_combined[i] = std::complex<float>(code_data[i].real() - code_pilot[i].imag(), code_data[i].imag() + code_pilot[i].real());
        }
    if (gsh_acq_set_local_code(d_handle, 1, reinterpret_cast<const float*>(d_code_combined.data())) != GSH_OK) d_error = gsh_last_error();
}


void Hip_Pcps_Cccwsr_Core::init()
{
    d_result = Hip_Detector_Result();
    d_well_count = 0;
    d_mag = 0.0;  // cleared here only (cccwsr.cc:160): a running maximum over the dwells of one acquisition
    d_input_power = 0.0;
    d_test_statistics = 0.0;
    d_state = 1;
}


int Hip_Pcps_Cccwsr_Core::work(uint64_t sample_counter, const std::complex<float>* in)
{
    if (d_handle == nullptr) return -1;
    const float fft_normalization_factor = static_cast<float>(d_fft_size) * static_cast<float>(d_fft_size);  // cccwsr.cc:178
    d_well_count++;

    // one dwell, two code slots, shared forward transforms; the input power is overwritten by every dwell (cccwsr.cc:192-194)
    gsh_acq_result r[2]{};
    if (gsh_acq_dwell(d_handle, reinterpret_cast<const float*>(in), 2, 0, 1, r) != GSH_OK || gsh_acq_input_power(d_handle, &d_input_power) != GSH_OK ||
        gsh_acq_read_row_peaks(d_handle, 0, d_peak_plus.data(), d_index_plus.data()) != GSH_OK ||
        gsh_acq_read_row_peaks(d_handle, 1, d_peak_minus.data(), d_index_minus.data()) != GSH_OK)
        {
            d_error = gsh_last_error();
            return -1;
        }
    for (uint32_t doppler_index = 0; doppler_index < d_num_doppler_bins; doppler_index++)
        {
            const int32_t doppler = -d_acq_params.doppler_max + d_acq_params.doppler_step * static_cast<int32_t>(doppler_index);  // cccwsr.cc:200
            const float magt_plus = d_peak_plus[doppler_index] / (fft_normalization_factor * fft_normalization_factor);    // :248
            const float magt_minus = d_peak_minus[doppler_index] / (fft_normalization_factor * fft_normalization_factor);  // :252
            const bool plus = magt_plus >= magt_minus;                                                                      // :254
            const float magt = plus ? magt_plus : magt_minus;
            const uint32_t indext = plus ? d_index_plus[doppler_index] : d_index_minus[doppler_index];
            if (d_mag < magt)  // :266, strictly greater
                {
                    d_mag = magt;
                    d_winning_branch = plus ? 0 : 1;
                    d_result.index_time = indext;
                    d_result.index_doppler = doppler_index;
                    d_result.Acq_delay_samples = static_cast<double>(indext % static_cast<int32_t>(d_acq_params.samples_per_code));
                    d_result.Acq_doppler_hz = static_cast<double>(doppler);
                    d_result.Acq_samplestamp_samples = sample_counter;
                    d_result.Acq_doppler_step = static_cast<uint32_t>(d_acq_params.doppler_step);
                }
        }

    d_test_statistics = d_mag / d_input_power;  // cccwsr.cc:292
    if (d_test_statistics > d_acq_params.threshold)
        {
            d_state = 2;  // Positive acquisition
        }
    else if (d_well_count == d_acq_params.max_dwells)
        {
            d_state = 3;  // Negative acquisition
        }
    return d_state;
}


// ---------------------------------------------------------------------------------------------------------------- QuickSync
Hip_Pcps_Quicksync_Core::Hip_Pcps_Quicksync_Core(const Hip_Acq_Conf& conf, uint32_t code_length, uint32_t folding_factor, uint32_t max_dwells, int device)
    : d_acq_params(conf), d_samples_per_code(code_length), d_folding_factor(folding_factor), d_max_dwells(max_dwells)
{
    if (folding_factor < 1 || folding_factor > 100 || code_length < folding_factor)  // complex_acumulator is std::array<gr_complex, 100> (qs.cc:295)
        {
            d_error = "folding_factor outside 1..100";
            return;
        }
    d_fft_size = d_samples_per_code / d_folding_factor;  // qs.cc:58
    d_num_doppler_bins = count_bins(conf.doppler_max, conf.doppler_step);
    d_handle = make_handle(conf, d_fft_size, d_num_doppler_bins, 1, device, &d_error, d_folding_factor * d_folding_factor);
    d_code.assign(d_samples_per_code, std::complex<float>(0.0F, 0.0F));
    d_code_folded.assign(d_fft_size, std::complex<float>(0.0F, 0.0F));
    d_accumulator.resize(d_folding_factor);
    d_corr_output_f.resize(d_folding_factor);
    d_possible_delay.resize(d_folding_factor);
    d_peak.resize(d_num_doppler_bins);
    d_index.resize(d_num_doppler_bins);
}


Hip_Pcps_Quicksync_Core::~Hip_Pcps_Quicksync_Core()
{
    if (d_handle != nullptr) gsh_acq_destroy(d_handle);
}


void Hip_Pcps_Quicksync_Core::set_local_code(const std::complex<float>* code)
{
    if (d_handle == nullptr) return;
    // a local copy of the code without the folding, for the correlation in time of the final step
    for (uint32_t i = 0; i < d_samples_per_code; i++) d_code[i] = code[i];
    // folding of the code by the folding factor (qs.cc:142-150)
    for (uint32_t k = 0; k < d_fft_size; k++) d_code_folded[k] = std::complex<float>(0.0F, 0.0F);
    for (uint32_t i = 0; i < d_folding_factor; i++)
        for (uint32_t k = 0; k < d_fft_size; k++) d_code_folded[k] += code[i * d_fft_size + k];
    if (gsh_acq_set_local_code(d_handle, 0, reinterpret_cast<const float*>(d_code_folded.data())) != GSH_OK) d_error = gsh_last_error();
}


void Hip_Pcps_Quicksync_Core::init()
{
    d_result = Hip_Detector_Result();
    d_well_count = 0;
    d_mag = 0.0;
    d_input_power = 0.0;
    d_test_statistics = 0.0;
    d_state = 1;
}


int Hip_Pcps_Quicksync_Core::work(uint64_t sample_counter, const std::complex<float>* in)
{
    if (d_handle == nullptr) return -1;
    const float fft_normalization_factor = static_cast<float>(d_fft_size) * static_cast<float>(d_fft_size);
    d_input_power = 0.0;
    d_mag = 0.0;
    d_test_statistics = 0.0;
    d_well_count++;

    // wipe-off, folding, transforms, |.|^2 and per-bin maxima for every Doppler bin (qs.cc:232-289): one dwell
    gsh_acq_result r{};
    if (gsh_acq_dwell(d_handle, reinterpret_cast<const float*>(in), 1, 0, 1, &r) != GSH_OK || gsh_acq_input_power(d_handle, &d_input_power) != GSH_OK ||
        gsh_acq_read_row_peaks(d_handle, 0, d_peak.data(), d_index.data()) != GSH_OK)
        {
            d_error = gsh_last_error();
            return -1;
        }
    bool found = false;
    uint32_t best_bin = 0, best_index = 0;
    for (uint32_t doppler_index = 0; doppler_index < d_num_doppler_bins; doppler_index++)
        {
            const float magt = d_peak[doppler_index] / (fft_normalization_factor * fft_normalization_factor);  // qs.cc:289
            if (d_mag < magt)  // qs.cc:292
                {
                    d_mag = magt;
                    best_bin = doppler_index;
                    best_index = d_index[doppler_index];
                    found = true;
                }
        }
    if (found)
        {
            // qs.cc:306-343 runs at every update of d_mag and each run overwrites the previous one: only the winning bin's survives
            const uint32_t detected_delay_samples_folded = best_index % d_samples_per_code;
            for (uint32_t i = 0; i < d_folding_factor; i++) d_possible_delay[i] = detected_delay_samples_folded + i * d_fft_size;
            if (gsh_acq_time_correlate(d_handle, reinterpret_cast<const float*>(d_code.data()), d_samples_per_code, best_bin, d_possible_delay.data(), d_folding_factor,
                    reinterpret_cast<float*>(d_accumulator.data())) != GSH_OK)
                {
                    d_error = gsh_last_error();
                    return -1;
                }
            uint32_t indext = 0;
            for (uint32_t i = 0; i < d_folding_factor; i++)
                {
                    d_corr_output_f[i] = d_accumulator[i].real() * d_accumulator[i].real() + d_accumulator[i].imag() * d_accumulator[i].imag();
                    if (d_corr_output_f[i] > d_corr_output_f[indext]) indext = i;  // volk_gnsssdr_32f_index_max_32u: first maximum
                }
            d_result.index_time = best_index;
            d_result.index_doppler = best_bin;
            d_result.Acq_delay_samples = static_cast<double>(d_possible_delay[indext]);
            d_result.Acq_doppler_hz = static_cast<double>(-d_acq_params.doppler_max + d_acq_params.doppler_step * static_cast<int32_t>(best_bin));
            d_result.Acq_samplestamp_samples = sample_counter;
            d_result.Acq_doppler_step = static_cast<uint32_t>(d_acq_params.doppler_step);
            d_test_statistics = d_mag / d_input_power;  // qs.cc:342
        }

    if (!d_acq_params.bit_transition_flag)  // qs.cc:366-390
        {
            if (d_test_statistics > d_acq_params.threshold)
                {
                    d_state = 2;  // Positive acquisition
                }
            else if (d_well_count == d_max_dwells)
                {
                    d_state = 3;  // Negative acquisition
                }
        }
    else
        {
            if (d_well_count == d_max_dwells)  // d_max_dwells = 2
                {
                    d_state = (d_test_statistics > d_acq_params.threshold) ? 2 : 3;
                }
        }
    return d_state;
}


// ---------------------------------------------------------------------------------------------------------------- fine Doppler
Hip_Pcps_Fine_Doppler_Core::Hip_Pcps_Fine_Doppler_Core(const Hip_Acq_Conf& conf, bool consistent_grid, int device) : d_acq_params(conf), d_device(device)
{
    d_fft_size = static_cast<uint32_t>(conf.samples_per_ms);  // fd.cc:61
    d_num_doppler_points = static_cast<int>(std::floor(std::abs(2 * conf.doppler_max) / conf.doppler_step));  // :58 (integer division, as there)
    if (d_num_doppler_points < 1) return;
    gsh_acq_conf c{};
    c.fs_in = conf.fs_in;
    c.fft_size = d_fft_size;
    c.effective_fft_size = d_fft_size;
    c.consumed_samples = d_fft_size;
    c.num_doppler_bins = static_cast<uint32_t>(d_num_doppler_points);
    // bin i is wiped off at -doppler_max' + doppler_step * i: doppler_max' = doppler_step gives fd.cc:170 to the letter
    c.doppler_max = consistent_grid ? conf.doppler_max : conf.doppler_step;
    c.doppler_step = conf.doppler_step;
    c.samples_per_chip = static_cast<uint32_t>(std::ceil((1.0 / 1.023e6) * static_cast<float>(conf.fs_in)));  // :214
    c.samples_per_code = static_cast<float>(d_fft_size);
    c.use_cfar = 0;  // first to second peak, compute_CAF :182-236
    c.max_prn = 1;
    c.no_grid = 0;   // the grid accumulates over the dwells (:296)
    if (gsh_acq_create(device, &c, &d_handle) != GSH_OK)
        {
            d_error = gsh_last_error();
            d_handle = nullptr;
        }
    d_code.assign(d_fft_size, std::complex<float>(0.0F, 0.0F));
    d_10_ms_buffer.reserve(10U * d_fft_size);
}


Hip_Pcps_Fine_Doppler_Core::~Hip_Pcps_Fine_Doppler_Core()
{
    if (d_handle != nullptr) gsh_acq_destroy(d_handle);
}


void Hip_Pcps_Fine_Doppler_Core::set_local_code(const std::complex<float>* code)
{
    if (d_handle == nullptr) return;
    for (uint32_t i = 0; i < d_fft_size; i++) d_code[i] = code[i];
    if (gsh_acq_set_local_code(d_handle, 0, reinterpret_cast<const float*>(code)) != GSH_OK) d_error = gsh_last_error();
}


void Hip_Pcps_Fine_Doppler_Core::reset_grid()
{
    d_result = Hip_Detector_Result();
    d_well_count = 0;  // the first dwell overwrites the grid (accumulate = 0)
    d_test_statistics = 0.0;
    d_10_ms_buffer.clear();
}


int Hip_Pcps_Fine_Doppler_Core::compute_and_accumulate_grid(const std::complex<float>* in)
{
    if (d_handle == nullptr) return -1;
    gsh_acq_result r{};
    d_well_count++;
    if (gsh_acq_dwell(d_handle, reinterpret_cast<const float*>(in), 1, d_well_count > 1 ? 1 : 0, static_cast<uint32_t>(d_well_count), &r) != GSH_OK)
        {
            d_error = gsh_last_error();
            return -1;
        }
    d_last_statistic = r.test_statistics;
    d_last_index_time = r.index_time;
    d_last_index_doppler = r.index_doppler;
    d_10_ms_buffer.insert(d_10_ms_buffer.end(), in, in + d_fft_size);  // fd.cc:451-452
    return (d_well_count >= static_cast<int>(d_acq_params.max_dwells)) ? 2 : 1;
}


int Hip_Pcps_Fine_Doppler_Core::compute_CAF(uint64_t sample_counter)
{
    // the statistic of the accumulated grid was formed on the device by the last dwell (first peak, +-1 chip blanked, second peak)
    d_test_statistics = d_last_statistic;
    d_result.index_time = d_last_index_time;
    d_result.index_doppler = d_last_index_doppler;
    d_result.Acq_delay_samples = static_cast<double>(d_last_index_time);
    d_result.Acq_doppler_hz = static_cast<double>(static_cast<int32_t>(d_last_index_doppler) * d_acq_params.doppler_step - d_acq_params.doppler_max);  // :243
    d_result.Acq_samplestamp_samples = sample_counter;
    d_result.Acq_doppler_step = static_cast<uint32_t>(d_acq_params.doppler_step);
    return (d_test_statistics > d_acq_params.threshold) ? 3 : 5;
}


uint32_t Hip_Pcps_Fine_Doppler_Core::buffer_more(const std::complex<float>* in, uint32_t n_items)
{
    const size_t want = 10U * static_cast<size_t>(d_fft_size);
    const size_t room = d_10_ms_buffer.size() < want ? want - d_10_ms_buffer.size() : 0;
    const uint32_t take = static_cast<uint32_t>(std::min<size_t>(room, n_items));
    d_10_ms_buffer.insert(d_10_ms_buffer.end(), in, in + take);
    return take;
}


bool Hip_Pcps_Fine_Doppler_Core::estimate_Doppler()
{
    // Direct FFT
    const int zero_padding_factor = 8;
    const int prn_replicas = 10;
    const int signal_samples = prn_replicas * static_cast<int>(d_fft_size);
    const int fft_size_extended = signal_samples * zero_padding_factor;
    if (d_10_ms_buffer.size() < static_cast<size_t>(signal_samples)) return false;

    // 1. local code aligned with the acquisition code phase estimation: the replica handed to set_local_code is the one :333 generates
    std::vector<std::complex<float>> code_replica(static_cast<size_t>(signal_samples));
    for (uint32_t i = 0; i < d_fft_size; i++) code_replica[i] = d_code[i];
    const int shift_index = static_cast<int>(d_result.Acq_delay_samples);
    if (shift_index != 0)
        {
            // std::rotate(first, first + (d_fft_size - shift_index), first + d_fft_size - 1), :339 -- the last sample stays where it is
            const int len = static_cast<int>(d_fft_size) - 1;
            const int mid = static_cast<int>(d_fft_size) - shift_index;
            std::vector<std::complex<float>> head(code_replica.begin(), code_replica.begin() + len);
            for (int i = 0; i < len; i++) code_replica[i] = head[(i + mid) % len];
        }
    for (int n = 0; n < prn_replicas - 1; n++)
        for (uint32_t i = 0; i < d_fft_size; i++) code_replica[(n + 1) * d_fft_size + i] = code_replica[i];

    // 2.-4. code wipe-off, zero-padded FFT, |.|^2, arg-max: on the GPU
    uint32_t tmp_index_freq = 0;
    if (gsh_spectrum_peak(d_device, reinterpret_cast<const float*>(d_10_ms_buffer.data()), reinterpret_cast<const float*>(code_replica.data()),
            static_cast<uint32_t>(signal_samples), static_cast<uint32_t>(fft_size_extended), &tmp_index_freq, nullptr) != GSH_OK)
        {
            d_error = gsh_last_error();
            return false;
        }
    // fftFreqBins[tmp_index_freq], :361-373
    float f;
    const int half = fft_size_extended / 2;
    if (static_cast<int>(tmp_index_freq) < half)
        f = ((static_cast<float>(d_acq_params.fs_in) / 2.0F) * static_cast<float>(tmp_index_freq)) / (static_cast<float>(fft_size_extended) / 2.0F);
    else
        f = ((-static_cast<float>(d_acq_params.fs_in) / 2.0F) * static_cast<float>(half - (static_cast<int>(tmp_index_freq) - half))) / (static_cast<float>(fft_size_extended) / 2.0F);
    d_fine_doppler_hz = f;
    // 5. update the Doppler estimation in Hz
    if (std::abs(f - d_result.Acq_doppler_hz) < 1000) d_result.Acq_doppler_hz = static_cast<double>(f);
    return true;
}

// ------------------------------------------------------------------------------------------------ Galileo E5a non-coherent I + Q
Hip_Galileo_E5a_Noncoherent_Iq_Core::Hip_Galileo_E5a_Noncoherent_Iq_Core(const Hip_Acq_Conf& conf, bool both_signal_components, int CAF_window_hz,
    int Zero_padding, int device)
    : d_acq_params(conf), d_CAF_window_hz(CAF_window_hz), d_both_signal_components(both_signal_components)
{
    d_fft_size = static_cast<uint32_t>(static_cast<int>(conf.sampled_ms) * static_cast<int>(conf.samples_per_ms));  // e5a.cc:70
    d_sampled_ms = Zero_padding > 0 ? 1U : conf.sampled_ms;                                                        // e5a.cc:85-92
    d_num_doppler_bins = count_bins(conf.doppler_max, conf.doppler_step);                                           // e5a.cc:113-116
    int32_t n_slots = 1;
    if (d_both_signal_components) d_slot_QA = n_slots++;
    if (d_sampled_ms > 1)
        {
            d_slot_IB = n_slots++;
            if (d_both_signal_components) d_slot_QB = n_slots++;
        }
    d_handle = make_handle(conf, d_fft_size, d_num_doppler_bins, static_cast<uint32_t>(n_slots), device, &d_error);
    d_pair.resize(d_num_doppler_bins);
    d_inbuf.assign(d_fft_size, std::complex<float>(0.0F, 0.0F));
    if (d_CAF_window_hz > 0)
        {
            d_CAF_vector.assign(d_num_doppler_bins, 0.0F);
            d_CAF_vector_I.assign(d_num_doppler_bins, 0.0F);
            if (d_both_signal_components) d_CAF_vector_Q.assign(d_num_doppler_bins, 0.0F);
        }
}


Hip_Galileo_E5a_Noncoherent_Iq_Core::~Hip_Galileo_E5a_Noncoherent_Iq_Core()
{
    if (d_handle != nullptr) gsh_acq_destroy(d_handle);
}


void Hip_Galileo_E5a_Noncoherent_Iq_Core::set_local_code(const std::complex<float>* codeI, const std::complex<float>* codeQ)
{
    if (d_handle == nullptr) return;
    auto load = [&](int32_t slot) {
        if (gsh_acq_set_local_code(d_handle, static_cast<uint32_t>(slot), reinterpret_cast<const float*>(d_inbuf.data())) != GSH_OK) d_error = gsh_last_error();
    };
    const auto samples_per_code = std::min(static_cast<uint32_t>(d_acq_params.samples_per_code), d_fft_size);
    // DATA SIGNAL, CODE A: (1,1,1)
    std::copy(codeI, codeI + d_fft_size, d_inbuf.begin());
    load(d_slot_IA);
    // SAME FOR PILOT SIGNAL
    if (d_both_signal_components)
        {
            std::copy(codeQ, codeQ + d_fft_size, d_inbuf.begin());
            load(d_slot_QA);
        }
    // integration time above one code: the other possible combination.  Only the first replica is rewritten (e5a.cc:191-199); the rest of the
    // buffer keeps what the previous transform's input was
    if (d_sampled_ms > 1)
        {
            for (uint32_t i = 0; i < samples_per_code; i++) d_inbuf[i] = codeI[i] * std::complex<float>(-1, 0);
            load(d_slot_IB);
            if (d_both_signal_components)
                {
                    for (uint32_t i = 0; i < samples_per_code; i++) d_inbuf[i] = codeQ[i] * std::complex<float>(-1, 0);
                    load(d_slot_QB);
                }
        }
}


void Hip_Galileo_E5a_Noncoherent_Iq_Core::init()
{
    d_result = Hip_Detector_Result();
    d_well_count = 0;
    d_mag = 0.0;
    d_input_power = 0.0;
    d_test_statistics = 0.0;
    d_state = 1;
}


// The CAF stage of the E5a detector (galileo_e5a_noncoherent_iq_acquisition_caf_cc.cc:546-631, "e5a.cc"): the per-bin maxima of the I (and Q) searches are
// smoothed along the Doppler axis before the peak is taken -- every bin becomes a triangle-weighted mean of the bins within +-H of it, H = CAF_window_hz /
// (2 doppler_step), weight 1 - |distance| / (2 H), the window cut off at the ends of the grid.  Restated here as ONE windowed mean evaluated over three stretches of
// the grid (leading edge, interior, trailing edge); what differs between the stretches -- and between the I and Q profiles -- is kept exactly, because the float
// results are pinned to the reference block's (tests/detector_cases.py):
//   * distance with or without its sign: the reference drops the sign only in some of its loops; where it keeps it, the bins above the centre weigh MORE than 1;
//   * the normaliser (the sum of the weights the window would have with unsigned distances), each with its own association of the float operations, and the
//     trailing edge's Q normaliser partly in double.
void Hip_Galileo_E5a_Noncoherent_Iq_Core::caf_filter()
{
    const int bins = static_cast<int>(d_num_doppler_bins);
    const int H = d_CAF_window_hz / (2 * d_acq_params.doppler_step);
    const float Hf = static_cast<float>(H);
    const float w = 0.5F / Hf;

    // sum over bins [lo, hi) of profile[i] * weight(centre, i), accumulated in float in index order
    auto windowed_sum = [w](const std::vector<float>& profile, int centre, int lo, int hi, bool unsigned_distance) {
        float sum = 0.0F;
        for (int i = lo; i < hi; i++)
            {
                const int distance = unsigned_distance ? std::abs(centre - i) : (centre - i);
                sum += profile[static_cast<size_t>(i)] * (1.0F - w * static_cast<float>(distance));
            }
        return sum;
    };
    // one output bin: the I mean, plus the Q mean when both components are searched
    auto smooth = [&](int centre, int lo, int hi, bool unsigned_I, bool unsigned_Q, float norm_I, float norm_Q) {
        float value = windowed_sum(d_CAF_vector_I, centre, lo, hi, unsigned_I);
        value /= norm_I;
        if (d_both_signal_components)
            {
                float q = windowed_sum(d_CAF_vector_Q, centre, lo, hi, unsigned_Q);
                q /= norm_Q;
                value += q;
            }
        d_CAF_vector[static_cast<size_t>(centre)] = value;
    };

    // leading edge: the window starts at bin 0 (e5a.cc:551-571).  I keeps the sign of the distance, Q does not.
    for (int c = 0; c < H; c++)
        {
            const float cf = static_cast<float>(c);
            const float norm_I = 1.0F + static_cast<float>(H + c) - w * Hf * ((Hf + 1.0F) / 2.0F) - w * cf * (cf + 1.0F) / 2.0F;
            const float norm_Q = 1.0F + static_cast<float>(H + c) - w * Hf * static_cast<float>(H + 1) / 2.0F - w * cf * static_cast<float>(c + 1) / 2.0F;
            smooth(c, 0, H + c + 1, false, true, norm_I, norm_Q);
        }
    // interior: the whole window fits (e5a.cc:573-591).  Both profiles keep the sign; one normaliser for every bin.
    {
        const float norm = 1.0F + 2.0F * Hf - 2.0F * w * Hf * static_cast<float>(H + 1) / 2.0F;
        for (int c = H; c < bins - H; c++) smooth(c, c - H, c + H + 1, false, false, norm, norm);
    }
    // trailing edge: the window ends with the grid (e5a.cc:593-613).  Unsigned distances; the Q normaliser's two halved terms are halved in double.
    for (int c = bins - H; c < bins; c++)
        {
            if (c < 0) continue;  // (a window wider than the grid: the reference indexes out of bounds here)
            const int beyond = bins - c - 1;  // bins of the window above the centre
            const float norm_I = 1.0F + Hf + static_cast<float>(beyond) - w * Hf * (Hf + 1.0F) / 2.0F - w * static_cast<float>(beyond) * static_cast<float>(bins - c) / 2.0F;
            const float whole = 1.0F + Hf + static_cast<float>(beyond);
            const float norm_Q = static_cast<float>(static_cast<double>(whole) - static_cast<double>(w * Hf * static_cast<float>(H + 1.0)) / 2.0 -
                                                    static_cast<double>(w * static_cast<float>(beyond) * static_cast<float>(bins - c)) / 2.0);
            smooth(c, c - H, bins, true, true, norm_I, norm_Q);
        }
}


int Hip_Galileo_E5a_Noncoherent_Iq_Core::work(uint64_t sample_counter, const std::complex<float>* in)
{
    if (d_handle == nullptr) return -1;
    const float fft_normalization_factor = static_cast<float>(d_fft_size) * static_cast<float>(d_fft_size);
    d_input_power = 0.0;
    d_mag = 0.0;
    d_well_count++;

    // one dwell over the block's two to four local codes (the block: one forward and up to four inverse transforms per bin, e5a.cc:340-395), then
    // the per-bin choice, the I + Q addition and the arg-max of the sum on the device (:399-492)
    std::vector<gsh_acq_result> r(4);
    const uint32_t n_slots = 1U + (d_slot_QA >= 0 ? 1U : 0U) + (d_slot_IB >= 0 ? 1U : 0U) + (d_slot_QB >= 0 ? 1U : 0U);
    if (gsh_acq_dwell(d_handle, reinterpret_cast<const float*>(in), n_slots, 0, 1, r.data()) != GSH_OK || gsh_acq_input_power(d_handle, &d_input_power) != GSH_OK ||
        gsh_acq_noncoherent_pair_peaks(d_handle, d_slot_IA, d_slot_QA, d_slot_IB, d_slot_QB, d_pair.data()) != GSH_OK)
        {
            d_error = gsh_last_error();
            return -1;
        }
    for (uint32_t doppler_index = 0; doppler_index < d_num_doppler_bins; doppler_index++)
        {
            const int32_t doppler = -d_acq_params.doppler_max + d_acq_params.doppler_step * static_cast<int32_t>(doppler_index);
            const gsh_acq_pair_peak& p = d_pair[doppler_index];
            if (d_CAF_window_hz > 0)
                {
                    d_CAF_vector_I[doppler_index] = p.caf_i;
                    if (d_both_signal_components) d_CAF_vector_Q[doppler_index] = p.caf_q;
                }
            const uint32_t indext = p.index_time;
            const float magt = p.peak / (fft_normalization_factor * fft_normalization_factor);
            // 4- record the maximum peak and the associated synchronization parameters (e5a.cc:496-515)
            if (d_mag < magt)
                {
                    d_mag = magt;
                    if (d_test_statistics < (d_mag / d_input_power) || !d_acq_params.bit_transition_flag)
                        {
                            d_result.index_time = indext;
                            d_result.index_doppler = doppler_index;
                            d_result.Acq_delay_samples = static_cast<double>(indext % static_cast<int32_t>(d_acq_params.samples_per_code));
                            d_result.Acq_doppler_hz = static_cast<double>(doppler);
                            d_result.Acq_samplestamp_samples = sample_counter;
                            d_result.Acq_doppler_step = static_cast<uint32_t>(d_acq_params.doppler_step);
                            d_test_statistics = d_mag / d_input_power;
                        }
                }
        }
    // 6 OPTIONAL: CAF filter to avoid Doppler ambiguity in bit transition (e5a.cc:546-650)
    if (d_CAF_window_hz > 0)
        {
            caf_filter();
            uint32_t indext = 0;
            for (uint32_t i = 1; i < d_num_doppler_bins; i++)
                if (d_CAF_vector[i] > d_CAF_vector[indext]) indext = i;  // volk_gnsssdr_32f_index_max_32u: the lowest index among equals
            d_result.Acq_doppler_hz = static_cast<double>(-d_acq_params.doppler_max + d_acq_params.doppler_step * static_cast<int32_t>(indext));
        }
    if (d_well_count == d_acq_params.max_dwells)
        {
            d_state = d_test_statistics > d_acq_params.threshold ? 3 : 4;  // positive : negative acquisition (e5a.cc:651-661)
        }
    else
        {
            d_state = 1;
        }
    return d_state;
}
