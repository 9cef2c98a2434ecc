/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the gnss-sdr
 * tracking-correlator hot path.  Never linked into / loaded by the product
 * (gnss-sdr_amd/); only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use it.
 *
 * Parity status: PINNED.  Every function below is checked bit-for-bit
 * (integer/code outputs) or value-for-value (float32 outputs, exact equality)
 * against the reference's own sources compiled into oracle/_ref/ by
 * tests/test_oracle_vs_ref.py, and against the .npz files in tests/golden/ minted from
 * oracle/_ref (tests/golden/make_golden.py).
 *
 * All file:line citations are relative to /root/reference/.
 *   K/  = src/algorithms/libs/volk_gnsssdr_module/volk_gnsssdr/kernels/volk_gnsssdr/
 */
#ifndef GNSS_ORACLE_H
#define GNSS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C"
{
#endif

    /* GPS L1 C/A PRN generator: src/algorithms/libs/gps_sdr_signal_replica.cc:24-114 */
    int oracle_gps_l1_ca_code_gen_float(float* dest1023, int prn, unsigned int chip_shift);
    /* sampled complex replica (code in the IMAGINARY part): gps_sdr_signal_replica.cc:117-173 */
    int oracle_gps_l1_ca_code_gen_complex_sampled(float* dest_iq, unsigned int prn, int fs, unsigned int chip_shift);

    /* K/volk_gnsssdr_32f_xn_resampler_32f_xn.h:63-80 (generic) */
    void oracle_resampler(float** result, const float* code, float rem, float step, const float* shifts,
        unsigned int code_len, int n_taps, unsigned int n);
    /* K/volk_gnsssdr_32f_xn_high_dynamics_resampler_32f_xn.h:67-91 (generic) */
    void oracle_hd_resampler(float** result, const float* code, float rem, float step, float rate,
        const float* shifts, unsigned int code_len, int n_taps, unsigned int n);

    /* chip indices only (int32), same arithmetic as the two resamplers above */
    void oracle_code_indices(int32_t* idx /* [n_taps][n] */, float rem, float step, float rate, const float* shifts,
        unsigned int code_len, int n_taps, unsigned int n, int high_dyn);

    /*
     * One Cpu_Multicorrelator_Real_Codes::Carrier_wipeoff_multicorrelator_resampler call,
     * src/algorithms/tracking/libs/cpu_multicorrelator_real_codes.cc:103-126, over the
     * _generic protokernels (K/..rotator_dot_prod_32fc_xn.h:66-98, K/..high_dynamic_rotator..:68-109).
     * float32 arithmetic in the reference's order.  in/out interleaved complex64.
     */
    int oracle_mcorr(const float* code, int code_len, const float* shifts, int n_taps, const float* in_iq, int n,
        float rem_carr, float phase_step, float phase_rate_step, float rem_code, float code_step,
        float code_rate_step, int high_dyn, float* out_iq);

    /*
     * float64 "truth" of the same call: identical chip selection (the float32 index
     * expression DEFINES which chip a sample sees), exact carrier phase
     * exp(-j(rem + n*step [+ rate*e(n)])) and double accumulation.  out_iq: 2*n_taps doubles.
     * Also returns sum_n |in[n]| in *sum_abs (accumulation scale for the parity norm).
     */
    int oracle_mcorr_f64(const float* code, int code_len, const float* shifts, int n_taps, const float* in_iq, int n,
        float rem_carr, float phase_step, float phase_rate_step, float rem_code, float code_step,
        float code_rate_step, int high_dyn, double* out_iq, double* sum_abs);

    /* K/volk_gnsssdr_s32f_sincos_32fc.h:390-400 (generic) */
    void oracle_sincos(float* out_iq, float phase_inc, float* phase, unsigned int n);
    /* K/volk_gnsssdr_32f_index_max_32u.h:446-465 (generic; first index wins ties) */
    void oracle_index_max(uint32_t* target, const float* src, uint32_t n);

    /*
     * CPU timing leg ("port" baseline): n_threads workers pull channels from a queue,
     * each channel runs `epochs` oracle_mcorr calls over consecutive windows of a shared
     * stream (same harness shape as cpu_multicorrelator_real_codes_test.cc:137-158).
     * params: 6 floats per channel {rem_carr, phase_step, rem_code, code_step, start_offset, 0}.
     * Returns elapsed seconds.
     */
    double oracle_mcorr_time(const float* codes, int code_len, const float* shifts, int n_taps,
        const float* stream_iq, long stream_len, int n, int n_channels, int epochs, int n_threads,
        const float* params, float* out_iq);

#ifdef __cplusplus
}
#endif
#endif
