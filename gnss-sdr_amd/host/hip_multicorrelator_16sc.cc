/*!
 * \file hip_multicorrelator_16sc.cc
 * \brief Thin C++ shell over the gsh_mcorr16_* C ABI; see the header.
 */
#include "hip_multicorrelator_16sc.h"
#include "gnss_sdr_hip.h"
#include <cstdlib>

static_assert(sizeof(Hip_Multicorrelator_16sc::lv_16sc) == 2 * sizeof(int16_t), "complex int16 is two int16 side by side");


Hip_Multicorrelator_16sc::~Hip_Multicorrelator_16sc()
{
    if (d_handle != nullptr)
        {
            gsh_mcorr16_destroy(d_handle);
            d_handle = nullptr;
        }
}


bool Hip_Multicorrelator_16sc::check(int rc)
{
    if (rc == GSH_OK) return true;
    d_error = gsh_last_error();
    return false;
}


bool Hip_Multicorrelator_16sc::ensure_handle()
{
    if (d_handle != nullptr) return true;
    int device = d_device;
    if (device < 0)
        {
            const char* env = std::getenv("GNSS_SDR_HIP_DEVICE");
            device = env ? std::atoi(env) : 0;
        }
    return check(gsh_mcorr16_create(device, &d_handle));
}


bool Hip_Multicorrelator_16sc::init(int max_signal_length_samples, int n_correlators)
{
    d_n_correlators = n_correlators;
    d_max_signal_length_samples = max_signal_length_samples;
    return ensure_handle() && check(gsh_mcorr16_init(d_handle, max_signal_length_samples, n_correlators));
}


bool Hip_Multicorrelator_16sc::set_local_code_and_taps(int code_length_chips, const lv_16sc* local_code_in, float* shifts_chips)
{
    return ensure_handle() && check(gsh_mcorr16_set_local_code_and_taps(d_handle, code_length_chips, reinterpret_cast<const int16_t*>(local_code_in), shifts_chips));
}


bool Hip_Multicorrelator_16sc::set_input_output_vectors(lv_16sc* corr_out, const lv_16sc* sig_in)
{
    d_corr_out = corr_out;
    return ensure_handle() && check(gsh_mcorr16_set_input_output_vectors(d_handle, reinterpret_cast<int16_t*>(corr_out), reinterpret_cast<const int16_t*>(sig_in)));
}


void Hip_Multicorrelator_16sc::update_local_code(int correlator_length_samples, float rem_code_phase_chips, float code_phase_step_chips)
{
    (void)rem_code_phase_chips;
    (void)code_phase_step_chips;
    if (correlator_length_samples < 0 || correlator_length_samples > d_max_signal_length_samples)
        d_error = "update_local_code: correlator_length_samples " + std::to_string(correlator_length_samples) + " outside what init() sized (" +
                  std::to_string(d_max_signal_length_samples) + ")";
}


bool Hip_Multicorrelator_16sc::Carrier_wipeoff_multicorrelator_resampler(float rem_carrier_phase_in_rad, float phase_step_rad, float rem_code_phase_chips,
    float code_phase_step_chips, int signal_length_samples)
{
    const bool ok = ensure_handle() && check(gsh_mcorr16_carrier_wipeoff_multicorrelator_resampler(d_handle, rem_carrier_phase_in_rad, phase_step_rad, rem_code_phase_chips,
                                           code_phase_step_chips, signal_length_samples));
    if (!ok && d_corr_out != nullptr)
        for (int i = 0; i < d_n_correlators; i++) d_corr_out[i] = lv_16sc(0, 0);  // never leave the previous call's sums behind a failed one
    return ok;
}


bool Hip_Multicorrelator_16sc::free()
{
    if (d_handle == nullptr) return true;
    return check(gsh_mcorr16_free(d_handle));
}
