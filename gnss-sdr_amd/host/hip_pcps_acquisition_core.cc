/*!
 * \file hip_pcps_acquisition_core.cc
 * \brief See the header.  Sizes follow acq.cc:110-117, the dwell / threshold logic acq.cc:686-727.
 */
#include "hip_pcps_acquisition_core.h"
#include <algorithm>
#include "gnss_sdr_hip.h"

Hip_Pcps_Acquisition_Core::Hip_Pcps_Acquisition_Core(const Hip_Acq_Conf& conf, int device, uint32_t num_doppler_bins_override)
    : d_acq_parameters(conf)
{
    // acq.cc:110-113
    d_consumed_samples = static_cast<uint32_t>(conf.sampled_ms * conf.samples_per_ms * (conf.bit_transition_flag ? 2.0 : 1.0));
    d_fft_size = (conf.sampled_ms == conf.ms_per_code) ? d_consumed_samples : d_consumed_samples * 2;
    d_effective_fft_size = conf.bit_transition_flag ? (d_fft_size / 2) : d_fft_size;
    d_num_doppler_bins = num_doppler_bins_override ? num_doppler_bins_override
                                                   : static_cast<uint32_t>(std::ceil(static_cast<double>(2 * conf.doppler_max) / static_cast<double>(conf.doppler_step)));
    // acq.cc:116-117
    d_threshold = conf.pfa > 0.0F ? compute_threshold(conf.pfa, d_effective_fft_size, d_num_doppler_bins, conf.bit_transition_flag ? 1 : conf.max_dwells) : conf.threshold;
    d_threshold_step_two = conf.pfa2 > 0.0F ? compute_threshold(conf.pfa2, d_effective_fft_size, conf.num_doppler_bins_step2, conf.bit_transition_flag ? 1 : conf.max_dwells) : conf.threshold;

    gsh_acq_conf c{};
    c.fs_in = conf.use_automatic_resampler ? conf.resampled_fs : conf.fs_in;  // acq.cc:277
    c.fft_size = d_fft_size;
    c.effective_fft_size = d_effective_fft_size;
    c.consumed_samples = d_consumed_samples;
    c.num_doppler_bins = d_num_doppler_bins;
    c.doppler_max = conf.doppler_max;
    c.doppler_step = conf.doppler_step;
    c.doppler_center = 0;
    c.doppler_bias = 0;
    c.samples_per_chip = conf.samples_per_chip;
    c.samples_per_code = conf.samples_per_code;
    c.bit_transition_flag = conf.bit_transition_flag ? 1 : 0;
    c.use_cfar = conf.use_CFAR_algorithm_flag ? 1 : 0;
    c.max_prn = 1;
    // the |.|^2 grid is only consumed by later non-coherent dwells (acq.cc:549-553) and by dump (acq.cc:555-558)
    c.no_grid = (conf.max_dwells <= 1U && !conf.dump) ? 1 : 0;
    c.transform_path = 0;
    c.num_doppler_bins_step2 = conf.make_2_steps ? conf.num_doppler_bins_step2 : 0U;  // acq.cc:171-175
    c.doppler_step2 = conf.doppler_step2;
    d_engine_conf = c;
    if (gsh_acq_create(device, &c, &d_handle) != GSH_OK)
        {
            d_error = gsh_last_error();
            d_handle = nullptr;
        }
}


Hip_Pcps_Acquisition_Core::~Hip_Pcps_Acquisition_Core()
{
    if (d_handle != nullptr) gsh_acq_destroy(d_handle);
}


float Hip_Pcps_Acquisition_Core::compute_threshold(float pfa, uint32_t effective_fft_size, uint32_t num_doppler_bins, uint32_t max_dwells)
{
    return gsh_acq_compute_threshold(pfa, effective_fft_size, num_doppler_bins, max_dwells);
}


void Hip_Pcps_Acquisition_Core::set_local_code(const std::complex<float>* code)
{
    if (d_handle == nullptr) return;
    if (gsh_acq_set_local_code(d_handle, 0, reinterpret_cast<const float*>(code)) != GSH_OK) d_error = gsh_last_error();
}


void Hip_Pcps_Acquisition_Core::set_doppler_center(int32_t doppler_center)
{
    if (d_handle == nullptr) return;
    if (doppler_center != d_doppler_center)  // acq.cc:741
        {
            d_doppler_center = doppler_center;
            if (gsh_acq_set_doppler_center(d_handle, doppler_center) != GSH_OK) d_error = gsh_last_error();
        }
}


void Hip_Pcps_Acquisition_Core::set_doppler_bias(int32_t doppler_bias)
{
    if (d_handle == nullptr) return;
    d_doppler_bias = doppler_bias;
    if (gsh_acq_set_doppler_bias(d_handle, doppler_bias) != GSH_OK) d_error = gsh_last_error();
}


// one gsh_acq_* dwell in the current step; `in` is complex<float> or complex<int16_t> data
template <typename In>
static int run_dwell(gsh_acq* h, const In* data, bool cshort, bool step_two, float center_step_two, float input_power, int accumulate, uint32_t count,
    gsh_acq_result* r)
{
    if (step_two)
        {
            const uint32_t slot = 0;
            if (cshort) return GSH_ERR_UNSUPPORTED;  // handled by the caller (converted on the host side of the ABI is not offered)
            return gsh_acq_dwell_step2(h, reinterpret_cast<const float*>(data), 1, &slot, &center_step_two, &input_power, accumulate, count, r);
        }
    if (cshort) return gsh_acq_dwell_cshort(h, reinterpret_cast<const int16_t*>(data), 1, accumulate, count, r);
    return gsh_acq_dwell(h, reinterpret_cast<const float*>(data), 1, accumulate, count, r);
}


Hip_Pcps_Acquisition_Core::Outcome Hip_Pcps_Acquisition_Core::core_after_dwell(uint64_t sample_count, bool dwell_ok, const void* gsh_result, AcquisitionResult* result)
{
    if (!dwell_ok)
        {
            d_error = gsh_last_error();
            d_num_noncoherent_integrations_counter = 0;
            d_step_two = false;
            return ACQ_ERROR;  // surfaces as "no detection", never as an exception across the GNU Radio thread
        }
    const gsh_acq_result& r = *static_cast<const gsh_acq_result*>(gsh_result);
    result->sample_count = sample_count;
    result->index_time = r.index_time;
    result->doppler = r.doppler_hz;
    result->test_statistics = r.test_statistics;
    result->positive_acq = false;
    result->step_two = d_step_two;
    if (!d_step_two) d_input_power = r.input_power;  // acq.cc:428-431: only step one refreshes d_input_power

    Outcome out = ACQ_CONTINUE;
    bool integration_done = false;
    // acq.cc:605-632 handle_threshold_reached
    auto threshold_reached = [&]() {
        if (d_acq_parameters.make_2_steps)
            {
                if (d_step_two)
                    {
                        result->positive_acq = true;
                        out = ACQ_POSITIVE;
                    }
                else
                    {
                        d_doppler_center_step_two = static_cast<float>(result->doppler);  // acq.cc:619
                        d_num_noncoherent_integrations_counter = 0;                       // acq.cc:621
                    }
                d_step_two = !d_step_two;  // acq.cc:624
            }
        else
            {
                result->positive_acq = true;
                out = ACQ_POSITIVE;
            }
    };
    if (!d_acq_parameters.bit_transition_flag)  // acq.cc:686-704
        {
            if (result->test_statistics > get_threshold())
                {
                    threshold_reached();
                }
            if (d_num_noncoherent_integrations_counter == d_acq_parameters.max_dwells) integration_done = true;
        }
    else  // acq.cc:705-715
        {
            if (result->test_statistics > d_threshold)
                {
                    threshold_reached();
                }
            else
                {
                    integration_done = true;
                }
        }
    if (integration_done)  // acq.cc:635-645 handle_integration_done: negative unless this very dwell was the positive one
        {
            if (out != ACQ_POSITIVE) out = ACQ_NEGATIVE;
            d_step_two = false;
        }
    // acq.cc:717-725
    if (d_num_noncoherent_integrations_counter == d_acq_parameters.max_dwells || result->positive_acq || d_acq_parameters.bit_transition_flag)
        {
            result->search_complete = true;  // the block dumps here when asked to (acq.cc:719-723)
            result->num_dwells = d_num_noncoherent_integrations_counter;
            d_num_noncoherent_integrations_counter = 0U;
        }
    return out;
}


Hip_Pcps_Acquisition_Core::Outcome Hip_Pcps_Acquisition_Core::acquisition_core(uint64_t sample_count, const std::complex<float>* data, AcquisitionResult* result)
{
    if (d_handle == nullptr || data == nullptr || result == nullptr) return ACQ_ERROR;
    d_num_noncoherent_integrations_counter++;  // acq.cc:668
    gsh_acq_result r{};
    const int accumulate = d_num_noncoherent_integrations_counter > 1 ? 1 : 0;  // acq.cc:545-553
    const bool was_step_two = d_step_two;
    const int rc = run_dwell(d_handle, data, false, d_step_two, d_doppler_center_step_two, d_input_power, accumulate, d_num_noncoherent_integrations_counter, &r);
    if (rc == GSH_OK) keep_dump_grids(was_step_two);
    return core_after_dwell(sample_count, rc == GSH_OK, &r, result);
}


Hip_Pcps_Acquisition_Core::Outcome Hip_Pcps_Acquisition_Core::acquisition_core_shared(uint64_t sample_count, bool dwell_ok, const gsh_acq_result& r, AcquisitionResult* result)
{
    if (d_handle == nullptr || result == nullptr) return ACQ_ERROR;
    d_num_noncoherent_integrations_counter++;  // acq.cc:668
    if (dwell_ok) keep_dump_grids(d_step_two);
    return core_after_dwell(sample_count, dwell_ok, &r, result);
}


Hip_Pcps_Acquisition_Core::Outcome Hip_Pcps_Acquisition_Core::acquisition_core(uint64_t sample_count, const std::complex<int16_t>* data, AcquisitionResult* result)
{
    if (d_handle == nullptr || data == nullptr || result == nullptr) return ACQ_ERROR;
    d_num_noncoherent_integrations_counter++;  // acq.cc:668
    gsh_acq_result r{};
    const int accumulate = d_num_noncoherent_integrations_counter > 1 ? 1 : 0;
    int rc;
    if (d_step_two)
        {
            // acq.cc:653-656 converts before either step; the narrow-grid entry point takes complex64
            d_cshort_scratch.resize(d_consumed_samples);
            for (uint32_t i = 0; i < d_consumed_samples; i++)
                d_cshort_scratch[i] = std::complex<float>(static_cast<float>(data[i].real()), static_cast<float>(data[i].imag()));
            rc = run_dwell(d_handle, d_cshort_scratch.data(), false, true, d_doppler_center_step_two, d_input_power, accumulate,
                d_num_noncoherent_integrations_counter, &r);
        }
    else
        {
            rc = run_dwell(d_handle, data, true, false, 0.0F, 0.0F, accumulate, d_num_noncoherent_integrations_counter, &r);
        }
    if (rc == GSH_OK) keep_dump_grids(d_step_two);
    return core_after_dwell(sample_count, rc == GSH_OK, &r, result);
}


void Hip_Pcps_Acquisition_Core::keep_dump_grids(bool dwell_was_step_two)
{
    if (!(d_acq_parameters.dump && d_acq_parameters.make_2_steps) || d_handle == nullptr) return;
    const size_t eff = d_effective_fft_size, wide = static_cast<size_t>(d_num_doppler_bins) * eff, narrow = static_cast<size_t>(d_acq_parameters.num_doppler_bins_step2) * eff;
    if (d_dump_narrow_grid.size() != narrow) d_dump_narrow_grid.assign(narrow, 0.0F);
    std::vector<float> g(wide);
    if (gsh_acq_read_grid(d_handle, 0, g.data()) != GSH_OK) return;  // (the dump then reads the device itself, as for a one-step search)
    if (dwell_was_step_two)
        std::copy(g.begin(), g.begin() + static_cast<std::ptrdiff_t>(narrow), d_dump_narrow_grid.begin());  // rows 0 .. nbins2 - 1 of the PRN's grid
    else
        {
            d_dump_grid.swap(g);
            d_have_dump_grid = true;
        }
}


bool Hip_Pcps_Acquisition_Core::read_narrow_grid(float* grid) const
{
    const size_t narrow = static_cast<size_t>(d_acq_parameters.num_doppler_bins_step2) * d_effective_fft_size;
    if (grid == nullptr || narrow == 0) return false;
    if (d_dump_narrow_grid.size() == narrow)
        std::copy(d_dump_narrow_grid.begin(), d_dump_narrow_grid.end(), grid);
    else
        std::fill(grid, grid + narrow, 0.0F);
    return true;
}


bool Hip_Pcps_Acquisition_Core::read_grid(float* grid)
{
    if (d_handle == nullptr) return false;
    if (d_have_dump_grid && d_dump_grid.size() == static_cast<size_t>(d_num_doppler_bins) * d_effective_fft_size)
        {
            std::copy(d_dump_grid.begin(), d_dump_grid.end(), grid);
            return true;
        }
    if (gsh_acq_read_grid(d_handle, 0, grid) != GSH_OK)
        {
            d_error = gsh_last_error();
            return false;
        }
    return true;
}
