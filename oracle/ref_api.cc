/*
 * TEST INFRASTRUCTURE ONLY (oracle/_ref build) -- never linked into the product.
 *
 * C entry points over the *reference's own* objects, compiled from
 * /root/reference by oracle/Makefile:
 *   - Cpu_Multicorrelator_Real_Codes  (src/algorithms/tracking/libs/cpu_multicorrelator_real_codes.cc)
 *   - Cpu_Multicorrelator_16sc        (src/algorithms/tracking/libs/cpu_multicorrelator_16sc.cc)
 *   - gps_l1_ca_code_gen_*            (src/algorithms/libs/gps_sdr_signal_replica.cc)
 *   - galileo_e1_code_gen_*           (src/algorithms/libs/galileo_e1_signal_replica.cc)
 *   - gps_l5{i,q}_code_gen_*          (src/algorithms/libs/gps_l5_signal_replica.cc)
 *   - volk_gnsssdr protokernels       (through oracle/ref_kernels.c)
 *
 * Used to (1) pin oracle/gnss_oracle.c against the reference, (2) mint the
 * golden fixtures under tests/golden/, (3) time the reference CPU path
 * (bench.py cpu_baseline, kind "reference").
 */
#include "cpu_multicorrelator_real_codes.h"
#include "cpu_multicorrelator_16sc.h"
#include "tracking_FLL_PLL_filter.h"
#include "tracking_discriminators.h"
#include "tracking_loop_filter.h"
#include "bit_synchronizer.h"
#include "exponential_smoother.h"
#include "lock_detectors.h"
#include "galileo_e1_signal_replica.h"
#include "gps_l5_signal_replica.h"
#include "gps_sdr_signal_replica.h"
#include <volk_gnsssdr/volk_gnsssdr.h>
#include <array>
#include <atomic>
#include <chrono>
#include <complex>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

extern "C"
{
    /* protokernels with external linkage, oracle/ref_kernels.c */
    void ref_generic_resampler(float**, const float*, float, float, float*, unsigned int, int, unsigned int);
    void ref_generic_hd_resampler(float**, const float*, float, float, float, float*, unsigned int, int, unsigned int);
    void ref_generic_rotator_dot_prod(lv_32fc_t*, const lv_32fc_t*, const lv_32fc_t, lv_32fc_t*, const float**, int, unsigned int);
    void ref_generic_hd_rotator_dot_prod(lv_32fc_t*, const lv_32fc_t*, const lv_32fc_t, const lv_32fc_t, lv_32fc_t*, const float**, int, unsigned int);
    void ref_simd_resampler(float**, const float*, float, float, float*, unsigned int, int, unsigned int);
    void ref_simd_hd_resampler(float**, const float*, float, float, float, float*, unsigned int, int, unsigned int);
    void ref_simd_rotator_dot_prod(lv_32fc_t*, const lv_32fc_t*, const lv_32fc_t, lv_32fc_t*, const float**, int, unsigned int);
    void ref_simd_hd_rotator_dot_prod(lv_32fc_t*, const lv_32fc_t*, const lv_32fc_t, const lv_32fc_t, lv_32fc_t*, const float**, int, unsigned int);
    void ref_generic_sincos(lv_32fc_t*, float, float*, unsigned int);
    void ref_generic_resampler_16ic(lv_16sc_t**, const lv_16sc_t*, float, float, float*, unsigned int, int, unsigned int);
    void ref_simd_resampler_16ic(lv_16sc_t**, const lv_16sc_t*, float, float, float*, unsigned int, int, unsigned int);
    void ref_generic_rotator_dot_prod_16ic(lv_16sc_t*, const lv_16sc_t*, const lv_32fc_t, lv_32fc_t*, const lv_16sc_t**, int, unsigned int);
    void ref_simd_rotator_dot_prod_16ic(lv_16sc_t*, const lv_16sc_t*, const lv_32fc_t, lv_32fc_t*, const lv_16sc_t**, int, unsigned int);
    void ref_generic_index_max(uint32_t*, const float*, uint32_t);
}

namespace
{
/* 0 = _generic protokernels (parity oracle); 1 = x86 SIMD protokernels (timing baseline). */
std::atomic<int> g_flavour{0};
}  // namespace

extern "C"
{
    /* ---- what the generated volk_gnsssdr library would provide ---------------- */
    size_t volk_gnsssdr_get_alignment(void) { return 32; }

    void* volk_gnsssdr_malloc(size_t size, size_t alignment)
    {
        void* p = nullptr;
        if (alignment < sizeof(void*)) alignment = sizeof(void*);
        if (posix_memalign(&p, alignment, size ? size : alignment) != 0) return nullptr;
        return p;
    }

    void volk_gnsssdr_free(void* p) { std::free(p); }

    void volk_gnsssdr_32f_xn_resampler_32f_xn(float** r, const float* c, float rem, float step, float* sh, unsigned int L, int nv, unsigned int n)
    {
        (g_flavour.load() ? ref_simd_resampler : ref_generic_resampler)(r, c, rem, step, sh, L, nv, n);
    }

    void volk_gnsssdr_32f_xn_high_dynamics_resampler_32f_xn(float** r, const float* c, float rem, float step, float rate, float* sh, unsigned int L, int nv, unsigned int n)
    {
        (g_flavour.load() ? ref_simd_hd_resampler : ref_generic_hd_resampler)(r, c, rem, step, rate, sh, L, nv, n);
    }

    void volk_gnsssdr_32fc_32f_rotator_dot_prod_32fc_xn(lv_32fc_t* res, const lv_32fc_t* in, const lv_32fc_t inc, lv_32fc_t* ph, const float** a, int nv, unsigned int n)
    {
        (g_flavour.load() ? ref_simd_rotator_dot_prod : ref_generic_rotator_dot_prod)(res, in, inc, ph, a, nv, n);
    }

    void volk_gnsssdr_32fc_32f_high_dynamic_rotator_dot_prod_32fc_xn(lv_32fc_t* res, const lv_32fc_t* in, const lv_32fc_t inc, const lv_32fc_t inc_rate, lv_32fc_t* ph, const float** a, int nv, unsigned int n)
    {
        (g_flavour.load() ? ref_simd_hd_rotator_dot_prod : ref_generic_hd_rotator_dot_prod)(res, in, inc, inc_rate, ph, a, nv, n);
    }

    void volk_gnsssdr_16ic_xn_resampler_16ic_xn(lv_16sc_t** r, const lv_16sc_t* c, float rem, float step, float* sh, unsigned int L, int nv, unsigned int n)
    {
        (g_flavour.load() ? ref_simd_resampler_16ic : ref_generic_resampler_16ic)(r, c, rem, step, sh, L, nv, n);
    }

    void volk_gnsssdr_16ic_x2_rotator_dot_prod_16ic_xn(lv_16sc_t* res, const lv_16sc_t* in, const lv_32fc_t inc, lv_32fc_t* ph, const lv_16sc_t** a, int nv, unsigned int n)
    {
        (g_flavour.load() ? ref_simd_rotator_dot_prod_16ic : ref_generic_rotator_dot_prod_16ic)(res, in, inc, ph, a, nv, n);
    }

    /* ---- test-facing API --------------------------------------------------------- */

    /* returns 1 if the SIMD flavour can run on this CPU */
    int ref_simd_supported(void)
    {
        __builtin_cpu_init();
        return (__builtin_cpu_supports("avx") && __builtin_cpu_supports("sse4.1")) ? 1 : 0;
    }

    void ref_set_flavour(int simd) { g_flavour.store((simd && ref_simd_supported()) ? 1 : 0); }

    /*
     * One call of the reference correlator object, exactly as
     * dll_pll_veml_tracking::do_correlation_step drives it (trk.cc:1232-1243):
     * init(2*n) -> set_high_dynamics_resampler -> set_local_code_and_taps ->
     * set_input_output_vectors -> Carrier_wipeoff_multicorrelator_resampler.
     * in/out are interleaved complex64.
     */
    int ref_mcorr_run(const float* code, int code_len, const float* shifts, int n_taps,
        const float* in_iq, int n, float rem_carr, float phase_step, float phase_rate_step,
        float rem_code, float code_step, float code_rate_step, int high_dyn, float* out_iq)
    {
        Cpu_Multicorrelator_Real_Codes mc;
        std::vector<float> taps(shifts, shifts + n_taps);
        std::vector<std::complex<float>> out(n_taps);
        mc.init(2 * n, n_taps);
        mc.set_high_dynamics_resampler(high_dyn != 0);
        mc.set_local_code_and_taps(code_len, code, taps.data());
        mc.set_input_output_vectors(out.data(), reinterpret_cast<const std::complex<float>*>(in_iq));
        if (high_dyn == 2)
            {
                /* the 6-argument overload (cpu_multicorrelator_real_codes.cc:129-144) */
                mc.Carrier_wipeoff_multicorrelator_resampler(rem_carr, phase_step, rem_code, code_step, code_rate_step, n);
            }
        else
            {
                mc.Carrier_wipeoff_multicorrelator_resampler(rem_carr, phase_step, phase_rate_step, rem_code, code_step, code_rate_step, n);
            }
        std::memcpy(out_iq, out.data(), sizeof(float) * 2 * n_taps);
        mc.free();
        return 0;
    }

    /*
     * One call of the reference's 16-bit correlator object (cpu_multicorrelator_16sc.cc): init(2*n) -> set_local_code_and_taps ->
     * set_input_output_vectors -> Carrier_wipeoff_multicorrelator_resampler.  code / in / out: interleaved int16 I, Q.
     */
    int ref_mcorr16_run(const int16_t* code_iq, int code_len, const float* shifts, int n_taps, const int16_t* in_iq, int n,
        float rem_carr, float phase_step, float rem_code, float code_step, int16_t* out_iq)
    {
        Cpu_Multicorrelator_16sc mc;
        std::vector<float> taps(shifts, shifts + n_taps);
        std::vector<lv_16sc_t> out(n_taps);
        mc.init(2 * n, n_taps);
        mc.set_local_code_and_taps(code_len, reinterpret_cast<const lv_16sc_t*>(code_iq), taps.data());
        mc.set_input_output_vectors(out.data(), reinterpret_cast<const lv_16sc_t*>(in_iq));
        mc.Carrier_wipeoff_multicorrelator_resampler(rem_carr, phase_step, rem_code, code_step, n);
        std::memcpy(out_iq, out.data(), sizeof(int16_t) * 2 * n_taps);
        mc.free();
        return 0;
    }

    /* the two phasors the class hands to the kernel, formed by the same expressions (cpu_multicorrelator_16sc.cc:89-93) in this library's C++ */
    void ref_mcorr16_phasors(float rem_carr, float phase_step, float* out4)
    {
        const lv_32fc_t p0 = lv_cmake(std::cos(rem_carr), -std::sin(rem_carr));
        const lv_32fc_t inc = std::exp(lv_32fc_t(0, -phase_step));
        out4[0] = p0.real();
        out4[1] = p0.imag();
        out4[2] = inc.real();
        out4[3] = inc.imag();
    }

    /* epochs back-to-back calls of one 16-bit object over consecutive windows of a stream: seconds (the timing leg of bench.py's 16-bit entry) */
    double ref_mcorr16_time(const int16_t* code_iq, int code_len, const float* shifts, int n_taps, const int16_t* stream_iq, long stream_len, int n,
        int epochs, float rem_carr, float phase_step, float rem_code, float code_step, int16_t* out_iq)
    {
        Cpu_Multicorrelator_16sc mc;
        std::vector<float> taps(shifts, shifts + n_taps);
        std::vector<lv_16sc_t> out(n_taps);
        mc.init(2 * n, n_taps);
        mc.set_local_code_and_taps(code_len, reinterpret_cast<const lv_16sc_t*>(code_iq), taps.data());
        const auto t0 = std::chrono::steady_clock::now();
        for (int e = 0; e < epochs; e++)
            {
                const long off = (static_cast<long>(e) * n) % (stream_len - n + 1);
                mc.set_input_output_vectors(out.data(), reinterpret_cast<const lv_16sc_t*>(stream_iq) + off);
                mc.Carrier_wipeoff_multicorrelator_resampler(rem_carr, phase_step, rem_code, code_step, n);
            }
        const auto t1 = std::chrono::steady_clock::now();
        std::memcpy(out_iq, out.data(), sizeof(int16_t) * 2 * n_taps);
        mc.free();
        return std::chrono::duration<double>(t1 - t0).count();
    }

    /*
     * Timing harness in the style of cpu_multicorrelator_real_codes_test.cc:137-158 (one
     * Cpu_Multicorrelator_Real_Codes object per worker thread, all reading one shared stream), extended so
     * that every host core has work: the (channel, block of 16 epochs) items are pulled from a queue by
     * n_threads std::threads.  Returns elapsed seconds.
     * params: per channel 6 floats {rem_carr, phase_step, rem_code, code_step, start_offset, unused}.
     */
    double ref_mcorr_time(const float* codes, int code_len, const float* shifts, int n_taps,
        const float* stream_iq, long stream_len, int n, int n_channels, int epochs, int n_threads,
        const float* params, float* out_iq)
    {
        constexpr int kBlock = 16;
        const int blocks = (epochs + kBlock - 1) / kBlock;
        const int n_items = n_channels * blocks;
        std::vector<std::thread> pool;
        const auto t0 = std::chrono::steady_clock::now();
        std::atomic<int> next{0};
        auto worker = [&]() {
            Cpu_Multicorrelator_Real_Codes mc;
            std::vector<float> taps(shifts, shifts + n_taps);
            std::vector<std::complex<float>> out(n_taps);
            mc.init(2 * n, n_taps);
            mc.set_high_dynamics_resampler(false);
            for (;;)
                {
                    const int item = next.fetch_add(1);
                    if (item >= n_items) break;
                    const int ch = item / blocks;
                    const int e0 = (item % blocks) * kBlock;
                    const int e1 = e0 + kBlock < epochs ? e0 + kBlock : epochs;
                    mc.set_local_code_and_taps(code_len, codes + static_cast<size_t>(ch) * code_len, taps.data());
                    const float* p = params + 6 * ch;
                    const long span = stream_len - n;
                    for (int e = e0; e < e1; e++)
                        {
                            const long pos = (static_cast<long>(p[4]) + static_cast<long>(e) * n) % (span > 0 ? span : 1);
                            mc.set_input_output_vectors(out.data(), reinterpret_cast<const std::complex<float>*>(stream_iq) + pos);
                            mc.Carrier_wipeoff_multicorrelator_resampler(p[0], p[1], 0.0F, p[2], p[3], 0.0F, n);
                        }
                    if (e1 == epochs) std::memcpy(out_iq + static_cast<size_t>(ch) * 2 * n_taps, out.data(), sizeof(float) * 2 * n_taps);
                }
            mc.free();
        };
        for (int t = 0; t < n_threads; t++) pool.emplace_back(worker);
        for (auto& th : pool) th.join();
        return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }

    /* single protokernels, generic flavour */
    void ref_resampler_generic(float** r, const float* c, float rem, float step, float* sh, unsigned int L, int nv, unsigned int n)
    {
        ref_generic_resampler(r, c, rem, step, sh, L, nv, n);
    }

    void ref_hd_resampler_generic(float** r, const float* c, float rem, float step, float rate, float* sh, unsigned int L, int nv, unsigned int n)
    {
        ref_generic_hd_resampler(r, c, rem, step, rate, sh, L, nv, n);
    }

    void ref_sincos_generic(float* out_iq, float phase_inc, float* phase, unsigned int n)
    {
        ref_generic_sincos(reinterpret_cast<lv_32fc_t*>(out_iq), phase_inc, phase, n);
    }

    void ref_index_max_generic(uint32_t* target, const float* src, uint32_t n)
    {
        ref_generic_index_max(target, src, n);
    }

    /* PRN generators */
    void ref_gps_l1_ca_code_gen_float(float* dest, int prn, unsigned int chip_shift)
    {
        gps_l1_ca_code_gen_float(own::span<float>(dest, 1023), prn, chip_shift);
    }

    void ref_gps_l1_ca_code_gen_complex_sampled(float* dest_iq, int n, unsigned int prn, int fs, unsigned int chip_shift)
    {
        gps_l1_ca_code_gen_complex_sampled(own::span<std::complex<float>>(reinterpret_cast<std::complex<float>*>(dest_iq), n), prn, fs, chip_shift);
    }

    /* signal: "1B" or "1C"; dest holds 2*4092 floats */
    void ref_galileo_e1_code_gen_sinboc11_float(float* dest, const char* signal, unsigned int prn)
    {
        std::array<char, 3> sig{{signal[0], signal[1], '\0'}};
        galileo_e1_code_gen_sinboc11_float(own::span<float>(dest, 2 * 4092), sig, prn);
    }

    void ref_galileo_e1_code_gen_complex_sampled(float* dest_iq, int n, const char* signal, int cboc, unsigned int prn, int fs, unsigned int chip_shift)
    {
        std::array<char, 3> sig{{signal[0], signal[1], '\0'}};
        galileo_e1_code_gen_complex_sampled(own::span<std::complex<float>>(reinterpret_cast<std::complex<float>*>(dest_iq), n), sig, cboc != 0, prn, fs, chip_shift);
    }

    void ref_gps_l5i_code_gen_float(float* dest, unsigned int prn)
    {
        gps_l5i_code_gen_float(own::span<float>(dest, 10230), prn);
    }

    void ref_gps_l5q_code_gen_float(float* dest, unsigned int prn)
    {
        gps_l5q_code_gen_float(own::span<float>(dest, 10230), prn);
    }

    /* ---- loop closure: discriminators (tracking_discriminators.cc) and loop filters, the reference's own objects ---- */
    double ref_fll_diff_atan(float p1re, float p1im, float p2re, float p2im, double t1, double t2)
    {
        return fll_diff_atan(gr_complex(p1re, p1im), gr_complex(p2re, p2im), t1, t2);
    }
    double ref_pll_four_quadrant_atan(float re, float im) { return pll_four_quadrant_atan(gr_complex(re, im)); }
    double ref_pll_cloop_two_quadrant_atan(float re, float im) { return pll_cloop_two_quadrant_atan(gr_complex(re, im)); }
    double ref_dll_nc_e_minus_l_normalized(float ere, float eim, float lre, float lim, float spc, float slope, float y_intercept)
    {
        return dll_nc_e_minus_l_normalized(gr_complex(ere, eim), gr_complex(lre, lim), spc, slope, y_intercept);
    }
    double ref_dll_nc_vemlp_normalized(float vere, float veim, float ere, float eim, float lre, float lim, float vlre, float vlim)
    {
        return dll_nc_vemlp_normalized(gr_complex(vere, veim), gr_complex(ere, eim), gr_complex(lre, lim), gr_complex(vlre, vlim));
    }
    /* Tracking_loop_filter: construct, initialize(initial_output), apply n inputs */
    void ref_loop_filter_run(float update_interval, float noise_bandwidth, int order, int include_last_integrator, float initial_output,
        const float* in, float* out, int n)
    {
        Tracking_loop_filter f(update_interval, noise_bandwidth, order, include_last_integrator != 0);
        f.initialize(initial_output);
        for (int i = 0; i < n; i++) out[i] = f.apply(in[i]);
    }
    /* Tracking_FLL_PLL_filter: set_params, initialize(doppler), n get_carrier_error calls */
    void ref_fll_pll_filter_run(float fll_bw_hz, float pll_bw_hz, int order, float acq_doppler_hz, const float* fll_disc,
        const float* pll_disc, float correlation_time_s, float* out, int n)
    {
        Tracking_FLL_PLL_filter f;
        f.set_params(fll_bw_hz, pll_bw_hz, order);
        f.initialize(acq_doppler_hz);
        for (int i = 0; i < n; i++) out[i] = f.get_carrier_error(fll_disc[i], pll_disc[i], correlation_time_s);
    }

    /* ---- lock detectors and smoother: the reference's own objects (T/lock_detectors.cc, T/exponential_smoother.cc) ---- */
    float ref_cn0_m2m4_estimator(const float* prompt_iq, int length, float coh_integration_time_s)
    {
        return cn0_m2m4_estimator(reinterpret_cast<const gr_complex*>(prompt_iq), length, coh_integration_time_s);
    }
    float ref_carrier_lock_detector(const float* prompt_iq, int length)
    {
        return carrier_lock_detector(reinterpret_cast<const gr_complex*>(prompt_iq), length);
    }
    /* configure like trk.cc:680-692 (min_value / offset < -1e30 keep the class defaults), then smooth n values */
    void ref_smoother_run(float alpha, int samples_for_initialization, float min_value, float offset, const float* raw, int n, float* out)
    {
        Exponential_Smoother s;
        s.set_alpha(alpha);
        if (min_value > -1e30F) s.set_min_value(min_value);
        if (offset > -1e30F) s.set_offset(offset);
        s.set_samples_for_initialization(samples_for_initialization);
        for (int i = 0; i < n; i++) out[i] = s.smooth(raw[i]);
    }

    /* HistogramBitSynchronizer (T/bit_synchronizer.cc): feed n prompts, record the lock event, locked flag, edge phase and
     * epochs_until_next_edge after every update */
    void ref_bit_sync_run(int bit_period_ms, int epoch_ms, int min_events_for_lock, int stable_best_required, double dominance_ratio,
        float min_prompt_mag, int use_phase_dot_detector, const float* prompts_iq, const int* quality_ok, int n, int* lock_event, int* edge_phase,
        int* until_next_edge)
    {
        HistogramBitSynchronizer::Config cfg;
        cfg.bit_period_ms = bit_period_ms;
        cfg.epoch_ms = epoch_ms;
        cfg.min_events_for_lock = min_events_for_lock;
        cfg.stable_best_required = stable_best_required;
        cfg.dominance_ratio = dominance_ratio;
        cfg.min_prompt_mag = min_prompt_mag;
        cfg.use_phase_dot_detector = use_phase_dot_detector != 0;
        HistogramBitSynchronizer b(cfg);
        b.reset();
        for (int i = 0; i < n; i++)
            {
                lock_event[i] = b.update(std::complex<float>(prompts_iq[2 * i], prompts_iq[2 * i + 1]), quality_ok[i] != 0) ? 1 : 0;
                edge_phase[i] = b.edge_phase();
                until_next_edge[i] = b.epochs_until_next_edge();
            }
    }
}
