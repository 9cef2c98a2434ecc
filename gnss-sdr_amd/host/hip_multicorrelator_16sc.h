/*!
 * \file hip_multicorrelator_16sc.h
 * \brief MI355X drop-in for gnss-sdr's Cpu_Multicorrelator_16sc
 *        (src/algorithms/tracking/libs/cpu_multicorrelator_16sc.h:38-57).
 *
 * Same public methods, argument order and borrowed-pointer rules as the reference class (the 16-bit member of the correlator family: complex int16
 * samples, local code and outputs).  Results are those of the reference's generic protokernels bit for bit (include/gnss_sdr_hip.h, gsh_mcorr16_*).
 * All arithmetic runs on the GPU through the C ABI; there is no CPU fallback: on failure the methods return false, last_error() says why and the
 * borrowed output vector is zeroed.
 */
#ifndef GNSS_SDR_HIP_MULTICORRELATOR_16SC_H
#define GNSS_SDR_HIP_MULTICORRELATOR_16SC_H

#include <complex>
#include <cstdint>
#include <string>

struct gsh_mcorr16;

class Hip_Multicorrelator_16sc
{
public:
    using lv_16sc = std::complex<int16_t>;  //!< the reference's lv_16sc_t in C++ (volk_gnsssdr_complex.h)

    Hip_Multicorrelator_16sc() = default;
    explicit Hip_Multicorrelator_16sc(int device) : d_device(device) {}
    ~Hip_Multicorrelator_16sc();
    Hip_Multicorrelator_16sc(const Hip_Multicorrelator_16sc&) = delete;
    Hip_Multicorrelator_16sc& operator=(const Hip_Multicorrelator_16sc&) = delete;

    bool init(int max_signal_length_samples, int n_correlators);
    bool set_local_code_and_taps(int code_length_chips, const lv_16sc* local_code_in, float* shifts_chips);
    bool set_input_output_vectors(lv_16sc* corr_out, const lv_16sc* sig_in);
    /*! cpu_multicorrelator_16sc.cc:65-77: in the reference this writes the resampled replicas the dot products then read; here the chip of every (tap, sample) is
     * selected inside the correlation kernel by the same float32 expression and no replica exists.  Kept so that code written against the reference class compiles. */
    void update_local_code(int correlator_length_samples, float rem_code_phase_chips, float code_phase_step_chips);
    bool Carrier_wipeoff_multicorrelator_resampler(float rem_carrier_phase_in_rad, float phase_step_rad, float rem_code_phase_chips, float code_phase_step_chips,
        int signal_length_samples);
    bool free();

    // not in the reference class
    void set_device(int device) { d_device = device; }
    const std::string& last_error() const { return d_error; }

private:
    bool ensure_handle();
    bool check(int rc);
    gsh_mcorr16* d_handle{nullptr};
    lv_16sc* d_corr_out{nullptr};  // borrowed
    int d_n_correlators{0};
    int d_max_signal_length_samples{0};
    int d_device{-1};
    std::string d_error;
};

#endif  // GNSS_SDR_HIP_MULTICORRELATOR_16SC_H
