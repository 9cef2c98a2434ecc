/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the gnss-sdr
 * tracking-correlator hot path.  Never linked into / loaded by the product
 * (gnss-sdr_amd/); only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use it.
 *
 * Parity status: PINNED.  Every function below is checked bit-for-bit
 * (integer/code outputs) or value-for-value (float32 outputs, exact equality)
 * against the reference's own sources compiled into oracle/_ref/ by
 * tests/test_oracle_golden.py, and against the .npz files in tests/golden/ minted from
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

    /*
     * The 16-bit family: one Cpu_Multicorrelator_16sc::Carrier_wipeoff_multicorrelator_resampler call,
     * T/cpu_multicorrelator_16sc.cc:80-96 over K/volk_gnsssdr_16ic_xn_resampler_16ic_xn.h:60-78 and
     * K/volk_gnsssdr_16ic_x2_rotator_dot_prod_16ic_xn.h:66-102 (generic).  code / in / out: interleaved int16 I, Q.
     * _phasors: the two complex floats the class forms with libm before it calls the kernel (phase0 re, im, increment re, im);
     * _phasor: the kernels alone with those handed in.
     */
    void oracle_mcorr16_phasors(float rem_carr, float phase_step, float* out4);
    int oracle_mcorr16_phasor(const int16_t* code_iq, int code_len, const float* shifts, int n_taps, const int16_t* in_iq, int n,
        float ph_re, float ph_im, float inc_re, float inc_im, float rem_code, float code_step, int16_t* out_iq);
    int oracle_mcorr16(const int16_t* code_iq, int code_len, const float* shifts, int n_taps, const int16_t* in_iq, int n,
        float rem_carr, float phase_step, float rem_code, float code_step, int16_t* out_iq);

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

    /* ================= loop closure (gnss_oracle_loop.c) ================= */
    /* T/ = src/algorithms/tracking/libs/ ; trk.cc = src/algorithms/tracking/gnuradio_blocks/dll_pll_veml_tracking.cc */
    double oracle_fll_diff_atan(float p1re, float p1im, float p2re, float p2im, double t1, double t2);        /* T/tracking_discriminators.cc:68-76 */
    double oracle_pll_four_quadrant_atan(float re, float im);                                                 /* :86-89 (atan2f for gr::fast_atan2f) */
    double oracle_pll_cloop_two_quadrant_atan(float re, float im);                                            /* :99-106 */
    double oracle_dll_nc_e_minus_l_normalized(float ere, float eim, float lre, float lim, float spc, float slope, float y_intercept); /* :117-127 */
    double oracle_dll_nc_vemlp_normalized(float vere, float veim, float ere, float eim, float lre, float lim, float vlre, float vlim); /* :139-149 */

    typedef struct oracle_loop_filter  /* Tracking_loop_filter, T/tracking_loop_filter.h */
    {
        float in_c[4], out_c[4];
        float in_h[4], out_h[4];
        int n_in, n_out, idx, order, last_integrator;
    } oracle_loop_filter;
    void oracle_loop_filter_design(oracle_loop_filter* f, float update_interval, float noise_bandwidth, int order, int include_last_integrator);
    void oracle_loop_filter_initialize(oracle_loop_filter* f, float initial_output);
    float oracle_loop_filter_apply(oracle_loop_filter* f, float x);

    typedef struct oracle_fll_pll_filter  /* Tracking_FLL_PLL_filter, T/tracking_FLL_PLL_filter.h */
    {
        float w, x, w0p, w0p2, w0p3, w0f, w0f2, a2, a3, b3;
        int order;
    } oracle_fll_pll_filter;
    void oracle_fll_pll_design(oracle_fll_pll_filter* f, float fll_bw_hz, float pll_bw_hz, int order);
    void oracle_fll_pll_initialize(oracle_fll_pll_filter* f, float acq_carrier_doppler_hz);
    float oracle_fll_pll_carrier_error(oracle_fll_pll_filter* f, float fll_disc, float pll_disc, float correlation_time_s);

    /* ---- direct resampler: direct_resampler_conditioner_cc.cc:39-129, the block's own stateful loop ---- */
    typedef struct oracle_direct_resampler_s
    {
        double fs_in, fs_out;
        uint32_t phase, lphase, phase_step;
    } oracle_direct_resampler_t;
    void oracle_direct_resampler_init(oracle_direct_resampler_t* r, double fs_in, double fs_out);
    /* one general_work call: produce up to noutput_items from `in` (n_in complex samples available); returns the number produced, *consumed
     * = min(count, n_in) as the block's consume_each (:127).  Like the block, it reads in[] as far as it needs to: the caller sizes n_in by
     * forecast (:64-69). */
    int oracle_direct_resampler_work(oracle_direct_resampler_t* r, const float* in_iq, int n_in, float* out_iq, int noutput_items, int* consumed);

    /* ---- lock detectors and C/N0: T/lock_detectors.cc, T/exponential_smoother.cc, trk.cc:1167-1224 ---- */
    float oracle_cn0_m2m4_estimator(const float* prompt_iq, int length, float coh_integration_time_s); /* T/lock_detectors.cc:61-110 */
    float oracle_carrier_lock_detector(const float* prompt_iq, int length);                             /* T/lock_detectors.cc:113-133 */
    typedef struct oracle_smoother  /* Exponential_Smoother, T/exponential_smoother.h:40-69 */
    {
        float alpha, one_minus_alpha, old_value, min_value, offset, init_sum;
        int samples_for_initialization, init_counter, initializing;
    } oracle_smoother;
    void oracle_smoother_init(oracle_smoother* s, float alpha, int samples_for_initialization, float min_value, float offset);
    void oracle_smoother_reset(oracle_smoother* s);
    float oracle_smoother_smooth(oracle_smoother* s, float raw);
#define ORACLE_MAX_CN0_SAMPLES 64
#define ORACLE_MAX_SECONDARY 320
    typedef struct oracle_lock_state  /* the members cn0_and_tracking_lock_status touches (trk.cc:1167-1224) */
    {
        float prompt_buffer[2 * ORACLE_MAX_CN0_SAMPLES];
        oracle_smoother cn0_smoother, carrier_lock_test_smoother;
        int cn0_estimation_counter, carrier_lock_fail_counter, code_lock_fail_counter;
        float cn0_db_hz;
        double carrier_lock_test;
    } oracle_lock_state;

    /* ---- histogram bit synchroniser (T/bit_synchronizer.{h,cc}) ---------------------------------------------------------- */
#define ORACLE_MAX_BITSYNC_BINS 64
    typedef struct oracle_bit_sync
    {
        /* Config (T/bit_synchronizer.h:104-137) */
        int32_t bins, min_events_for_lock, stable_best_required, use_phase_dot_detector;
        double dominance_ratio;
        float min_prompt_mag;
        /* state */
        int32_t hist[ORACLE_MAX_BITSYNC_BINS];
        int32_t total_events, locked, edge_phase, has_last_prompt, has_last_sign, last_sign, has_last_best_bin, last_best_bin, stable_best_count;
        int64_t epoch_count;
        float last_prompt[2];
    } oracle_bit_sync;
    void oracle_bit_sync_init(oracle_bit_sync* b, int bins, int min_events_for_lock, int stable_best_required, double dominance_ratio,
        float min_prompt_mag, int use_phase_dot_detector);
    int oracle_bit_sync_update(oracle_bit_sync* b, float p_re, float p_im, int tracking_quality_ok); /* 1 on the lock event */
    int oracle_bit_sync_epochs_until_next_edge(const oracle_bit_sync* b);

    /* same field order as gsh_trk_conf / gsh_trk_epoch (include/gnss_sdr_hip.h) so that one ctypes layout serves both */
    typedef struct oracle_trk_conf
    {
        double fs_in, code_chip_rate, signal_carrier_freq, cfo_frequency_hz;
        uint32_t code_length_chips, code_samples_per_chip, vector_length;
        int32_t veml, track_pilot;
        float early_late_space_chips, very_early_late_space_chips;
        float pll_bw_hz, dll_bw_hz, fll_bw_hz;
        int32_t pll_filter_order, dll_filter_order;
        int32_t enable_fll_pull_in, enable_fll_steady_state, carrier_aiding, cloop;
        uint32_t pull_in_time_s;
        float spc, slope, y_intercept;
        /* lock detectors and C/N0 (0 = off: the loop never declares loss of lock) */
        int32_t enable_lock_detectors;
        int32_t cn0_samples, cn0_min, max_code_lock_fail, max_carrier_lock_fail;
        int32_t cn0_smoother_samples, carrier_lock_test_smoother_samples;
        float cn0_smoother_alpha, carrier_lock_test_smoother_alpha;
        double carrier_lock_th;
        /* symbol synchronisation and the narrow-tracking state (0 = off: the loop stays in state 2) */
        int32_t enable_symbol_sync;
        int32_t symbols_per_bit;             /* d_symbols_per_bit */
        int32_t has_secondary;               /* d_secondary */
        int32_t secondary_code_length;       /* d_secondary_code_length (the telemetry preamble for signals without a secondary code) */
        int32_t data_secondary_code_length;  /* d_data_secondary_code_length */
        int32_t extend_correlation_symbols;  /* Dll_Pll_Conf::extend_correlation_symbols; > 1 enables extended integration (states 3/4) */
        uint8_t secondary_code[ORACLE_MAX_SECONDARY];       /* characters '0' / '1' */
        uint8_t data_secondary_code[ORACLE_MAX_SECONDARY];
        /* narrow-tracking parameters applied when extended integration starts (trk.cc:2126-2149; dll_pll_conf.h:49-54) */
        float pll_bw_narrow_hz, dll_bw_narrow_hz, early_late_space_narrow_chips, very_early_late_space_narrow_chips;
        /* histogram bit synchroniser (configure_bit_synchronizer, trk.cc:1387-1406; defaults dll_pll_conf.h:43,60,75-76,88) */
        int32_t use_histogram_bit_sync, bs_min_events_for_lock, bs_stable_best_required, bs_use_phase_dot_detector;
        float bs_min_prompt_mag;
        int32_t enable_bit_sync_time_limit; /* the fail-safe of trk.cc:2000-2007 (the reference applies it always; the engine's structure has a switch) */
        double bs_dominance_ratio;
        /* high dynamics (Dll_Pll_Conf::high_dyn, smoother_length; trk.cc:669-675, 1425-1443, 1458-1480) */
        int32_t high_dyn;
        uint32_t smoother_length;
        uint32_t bit_synchronization_time_limit_s; /* Dll_Pll_Conf, trk.cc:2002 */
        int32_t enable_doppler_correction;         /* Dll_Pll_Conf, trk.cc:1326-1346 */
    } oracle_trk_conf;
    void oracle_lock_init(oracle_lock_state* st, const oracle_trk_conf* c);
    /* cn0_and_tracking_lock_status, trk.cc:1167-1224: returns 1 while locked, 0 when loss of lock is declared */
    int oracle_lock_status(oracle_lock_state* st, const oracle_trk_conf* c, float p_re, float p_im, double coh_integration_time_s, int pull_in_transitory);

    typedef struct oracle_trk_epoch
    {
        uint64_t sample_counter;
        int32_t prn_length_samples;
        int32_t flags;
        float corr[10];
        float prompt_data[2];
        float rem_carr_phase_rad;
        float cn0_db_hz;
        double carrier_doppler_hz, code_freq_chips, carr_phase_error_hz, carr_freq_error_hz, carr_error_filt_hz;
        double code_error_chips, code_error_filt_chips, rem_code_phase_samples, acc_carrier_phase_rad;
        double carrier_lock_test;
        int32_t state;          /* d_state in which the period ran: 2 (wide tracking, symbol search) or 4 (narrow tracking) */
        int32_t symbol_flags;   /* bit 0: Flag_valid_symbol_output; bit 1: Flag_PLL_180_deg_phase_locked */
        float p_data_accu[2];   /* d_P_data_accu when a symbol is output (Prompt_I / Prompt_Q of the Gnss_Synchro), else the running sum */
        double carrier_phase_rate_step_rad, code_phase_rate_step_chips;  /* after update_tracking_vars (0 outside high_dyn) */
        float accu[10];         /* d_VE_accu .. d_VL_accu the loop worked on: the outputs in state 2, the running sums in states 3 / 4 */
    } oracle_trk_epoch;
#define ORACLE_MAX_SMOOTHER 32

    /* closed loop of ONE channel over a resident stream; returns the number of epochs completed */
    int oracle_trk_run(const oracle_trk_conf* c, const float* code, const float* data_code, int code_len, const float* stream_iq,
        uint64_t n_stream, uint64_t start_sample, uint64_t acq_sample_stamp, double acq_carrier_doppler_hz, int n_epochs,
        oracle_trk_epoch* rec);
    /* ... with flags; bit 0 = ORACLE_TRK_PULL_IN_OVER: the pull-in transitory was over at the pull-in call already (trk.cc:1910-1917 evaluated with the read pointer
     * of THAT call: oracle_pull_in_over) */
#define ORACLE_TRK_PULL_IN_OVER 1u
    int oracle_pull_in_over(const oracle_trk_conf* c, uint64_t nitems_read, uint64_t acq_sample_stamp);
    int oracle_trk_run_flags(const oracle_trk_conf* c, const float* code, const float* data_code, int code_len, const float* stream_iq,
        uint64_t n_stream, uint64_t start_sample, uint64_t acq_sample_stamp, double acq_carrier_doppler_hz, int n_epochs,
        oracle_trk_epoch* rec, unsigned flags);

#ifdef __cplusplus
}
#endif
#endif
