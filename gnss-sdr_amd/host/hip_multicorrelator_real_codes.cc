/*!
 * \file hip_multicorrelator_real_codes.cc
 * \brief Thin C++ shell over the gsh_mcorr_* C ABI; see the header.
 */
#include "hip_multicorrelator_real_codes.h"
#include <string>
#include "gnss_sdr_hip.h"
#include <cstdlib>

Hip_Multicorrelator_Real_Codes::~Hip_Multicorrelator_Real_Codes()
{
    if (d_handle != nullptr)
        {
            gsh_mcorr_destroy(d_handle);
            d_handle = nullptr;
        }
}


bool Hip_Multicorrelator_Real_Codes::check(int rc)
{
    if (rc == GSH_OK) return true;
    d_error = gsh_last_error();
    return false;
}


bool Hip_Multicorrelator_Real_Codes::ensure_handle()
{
    if (d_handle != nullptr) return true;
    int device = d_device;
    if (device < 0)
        {
            const char* env = std::getenv("GNSS_SDR_HIP_DEVICE");
            device = env ? std::atoi(env) : 0;
        }
    if (!check(gsh_mcorr_create(device, &d_handle))) return false;
    return check(gsh_mcorr_set_high_dynamics_resampler(d_handle, d_use_high_dynamics_resampler ? 1 : 0));
}


void Hip_Multicorrelator_Real_Codes::set_high_dynamics_resampler(bool use_high_dynamics_resampler)
{
    d_use_high_dynamics_resampler = use_high_dynamics_resampler;
    if (d_handle != nullptr) check(gsh_mcorr_set_high_dynamics_resampler(d_handle, use_high_dynamics_resampler ? 1 : 0));
}


bool Hip_Multicorrelator_Real_Codes::init(int max_signal_length_samples, int n_correlators)
{
    d_n_correlators = n_correlators;
    d_max_signal_length_samples = max_signal_length_samples;
    return ensure_handle() && check(gsh_mcorr_init(d_handle, max_signal_length_samples, n_correlators));
}


bool Hip_Multicorrelator_Real_Codes::set_local_code_and_taps(int code_length_chips, const float* local_code_in, float* shifts_chips)
{
    return ensure_handle() && check(gsh_mcorr_set_local_code_and_taps(d_handle, code_length_chips, local_code_in, shifts_chips));
}


bool Hip_Multicorrelator_Real_Codes::set_input_output_vectors(std::complex<float>* corr_out, const std::complex<float>* sig_in)
{
    d_corr_out = corr_out;
    return ensure_handle() && check(gsh_mcorr_set_input_output_vectors(d_handle, reinterpret_cast<float*>(corr_out), reinterpret_cast<const float*>(sig_in)));
}


void Hip_Multicorrelator_Real_Codes::update_local_code(int correlator_length_samples, float rem_code_phase_chips, float code_phase_step_chips,
    float code_phase_rate_step_chips)
{
    // nothing to resample ahead of time (see the header); a call that the reference would turn into an out-of-bounds write is reported instead
    (void)rem_code_phase_chips;
    (void)code_phase_step_chips;
    (void)code_phase_rate_step_chips;
    if (correlator_length_samples < 0 || correlator_length_samples > d_max_signal_length_samples)
        d_error = "update_local_code: correlator_length_samples " + std::to_string(correlator_length_samples) + " outside what init() sized (" +
                  std::to_string(d_max_signal_length_samples) + ")";
}


bool Hip_Multicorrelator_Real_Codes::Carrier_wipeoff_multicorrelator_resampler(float rem_carrier_phase_in_rad, float phase_step_rad,
    float phase_rate_step_rad, float rem_code_phase_chips, float code_phase_step_chips, float code_phase_rate_step_chips,
    int signal_length_samples)
{
    const bool ok = ensure_handle() && check(gsh_mcorr_carrier_wipeoff_multicorrelator_resampler(d_handle, rem_carrier_phase_in_rad, phase_step_rad,
                                           phase_rate_step_rad, rem_code_phase_chips, code_phase_step_chips, code_phase_rate_step_chips, signal_length_samples));
    if (!ok) invalidate_outputs();
    return ok;
}


bool Hip_Multicorrelator_Real_Codes::Carrier_wipeoff_multicorrelator_resampler(float rem_carrier_phase_in_rad, float phase_step_rad,
    float rem_code_phase_chips, float code_phase_step_chips, float code_phase_rate_step_chips, int signal_length_samples)
{
    const bool ok = ensure_handle() && check(gsh_mcorr_carrier_wipeoff_multicorrelator_resampler6(d_handle, rem_carrier_phase_in_rad, phase_step_rad,
                                           rem_code_phase_chips, code_phase_step_chips, code_phase_rate_step_chips, signal_length_samples));
    if (!ok) invalidate_outputs();
    return ok;
}


void Hip_Multicorrelator_Real_Codes::invalidate_outputs()
{
    // The tracking block never looks at the return value (trk.cc:1236-1256), so a failed call must not leave the previous period's
    // correlator outputs in place: zeros make the C/N0 estimate collapse and the lock detectors drop the channel within their fail counts,
    // instead of the loop "tracking" stale data for ever.  The error text stays in last_error().
    if (d_corr_out != nullptr)
        for (int i = 0; i < d_n_correlators; i++) d_corr_out[i] = std::complex<float>(0.0F, 0.0F);
}


bool Hip_Multicorrelator_Real_Codes::free()
{
    if (d_handle == nullptr) return true;
    return check(gsh_mcorr_free(d_handle));
}
