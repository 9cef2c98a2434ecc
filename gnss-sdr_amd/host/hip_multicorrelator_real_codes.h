/*!
 * \file hip_multicorrelator_real_codes.h
 * \brief MI355X drop-in for gnss-sdr's Cpu_Multicorrelator_Real_Codes
 *        (src/algorithms/tracking/libs/cpu_multicorrelator_real_codes.h:37-61).
 *
 * Same public methods, same argument order and meaning, same borrowed-pointer rules, so the tracking blocks that
 * own a correlator member (dll_pll_veml_tracking.h:94-95, kf_tracking.h:97, gps_l1_ca_gaussian_tracking_cc.h:175)
 * can switch by changing the member's type -- see INTEGRATION.md.  All arithmetic runs on the GPU through the C ABI
 * (include/gnss_sdr_hip.h, gsh_mcorr_*); there is no CPU fallback: if the device or the library is unavailable the
 * methods return false and last_error() says why (the reference's bools are always true and never checked,
 * mcorr.cc:49,62,71,125 -- callers that ignore them keep working, callers that check them see the failure).
 */
#ifndef GNSS_SDR_HIP_MULTICORRELATOR_REAL_CODES_H
#define GNSS_SDR_HIP_MULTICORRELATOR_REAL_CODES_H

#include <complex>
#include <string>

struct gsh_mcorr;

class Hip_Multicorrelator_Real_Codes
{
public:
    Hip_Multicorrelator_Real_Codes() = default;
    explicit Hip_Multicorrelator_Real_Codes(int device) : d_device(device) {}
    ~Hip_Multicorrelator_Real_Codes();
    Hip_Multicorrelator_Real_Codes(const Hip_Multicorrelator_Real_Codes&) = delete;
    Hip_Multicorrelator_Real_Codes& operator=(const Hip_Multicorrelator_Real_Codes&) = delete;

    void set_high_dynamics_resampler(bool use_high_dynamics_resampler);
    bool init(int max_signal_length_samples, int n_correlators);
    bool set_local_code_and_taps(int code_length_chips, const float* local_code_in, float* shifts_chips);
    bool set_input_output_vectors(std::complex<float>* corr_out, const std::complex<float>* sig_in);
    /*! mcorr.h:46 / mcorr.cc:75-101.  In the reference this writes the n_correlators resampled replicas that the following dot products read; both
     * Carrier_wipeoff overloads call it first (mcorr.cc:112, :137) and nothing else in the reference does.  On this engine the replicas never exist:
     * the chip of every (tap, sample) is selected inside the correlation kernel by the same float32 expression.  The method is kept so that code
     * written against the reference class compiles; it checks its arguments against what init() sized and otherwise does nothing. */
    void update_local_code(int correlator_length_samples, float rem_code_phase_chips, float code_phase_step_chips, float code_phase_rate_step_chips = 0.0);
    bool Carrier_wipeoff_multicorrelator_resampler(float rem_carrier_phase_in_rad, float phase_step_rad, float phase_rate_step_rad,
        float rem_code_phase_chips, float code_phase_step_chips, float code_phase_rate_step_chips, int signal_length_samples);
    bool Carrier_wipeoff_multicorrelator_resampler(float rem_carrier_phase_in_rad, float phase_step_rad, float rem_code_phase_chips,
        float code_phase_step_chips, float code_phase_rate_step_chips, int signal_length_samples);
    bool free();

    // not in the reference class: which GPU to use (default 0, or the GNSS_SDR_HIP_DEVICE environment variable)
    void set_device(int device) { d_device = device; }
    const std::string& last_error() const { return d_error; }

private:
    bool ensure_handle();
    bool check(int rc);
    void invalidate_outputs();  //!< a failed correlation zeroes the borrowed output vector (the caller ignores the return value, trk.cc:1236-1256)
    gsh_mcorr* d_handle{nullptr};
    std::complex<float>* d_corr_out{nullptr};  // borrowed (mcorr.cc:70)
    int d_n_correlators{0};
    int d_max_signal_length_samples{0};
    int d_device{-1};
    bool d_use_high_dynamics_resampler{true};  // same default as the reference (mcorr.h:60)
    std::string d_error;
};

#endif  // GNSS_SDR_HIP_MULTICORRELATOR_REAL_CODES_H
