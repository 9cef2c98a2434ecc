// Batched Early/Prompt/Late (VE/E/P/L/VL) multicorrelator for MI355X (gfx950, wave64).
//
// One launch evaluates many "jobs"; a job is one call of the reference's
//   Cpu_Multicorrelator_Real_Codes::Carrier_wipeoff_multicorrelator_resampler
//   (src/algorithms/tracking/libs/cpu_multicorrelator_real_codes.cc:103-126)
// i.e. code resampling
//   (K/volk_gnsssdr_32f_xn_resampler_32f_xn.h:63-80, K/..high_dynamics_resampler..:67-91)
// fused with carrier wipe-off and multiply-accumulate
//   (K/volk_gnsssdr_32fc_32f_rotator_dot_prod_32fc_xn.h:66-98, K/..high_dynamic_rotator..:68-109),
// K/ = src/algorithms/libs/volk_gnsssdr_module/volk_gnsssdr/kernels/volk_gnsssdr/ in the gnss-sdr tree.
//
// Design (not a translation of the reference's CPU kernels or of its CUDA block):
//   * no resampled-code buffers and no wiped-off signal buffer ever exist: the chip of
//     sample n is looked up on the fly from an LDS-resident copy of the +-1 code;
//   * the chip index uses the reference's float32 expression operation for operation
//     ((step*(float)n + shift) - rem, one IEEE rounding each; this file is compiled with
//     -ffp-contract=off), so chip selection is BIT-EXACT with the _generic protokernel;
//   * the carrier NCO is evaluated, not recurred from n=0: every thread seeds
//     exp(-j(rem + n*step)) exactly (double-precision phase, reduced mod 2pi) and steps it
//     by exp(-j*512*step) for at most MC_RESEED (32) strides before re-seeding, which keeps the
//     rotator within ~1e-6 of the exact value instead of the reference's O(1e-5) drift;
//   * each work-group (256 threads = 4 waves) streams its window with 16-byte loads
//     (two complex64 per lane, 1 KiB per wave-instruction), accumulates T complex sums per
//     lane in registers, reduces them with wave64 shuffles and one 4-way LDS step, and stores
//     T complex results: algorithmic traffic 8*N + 8*T bytes per job;
//   * blockIdx is remapped so that consecutive jobs (host order: epoch-major, channel-minor,
//     i.e. jobs that read the same samples) run on the same XCD and share its L2.
// Two work-group sizes of the same source (round 6): this translation unit builds the 256-thread kernels (four waves per job: every flavour, and the launcher
// every caller sees); multicorrelator_t128.hip includes it again with GSH_MC_THREADS = 128 and builds the E/P/L whole-code flavours with TWO waves per job.
// What a job costs besides its trips -- set-up, edge trips, fold, wave sums: ~330 vector instructions per WAVE (profiles/ab/r06/session5.txt) -- is then paid
// twice, not four times: 5 928 instead of 6 689 vector instructions per 25 000-sample job, 173 us instead of 184 per launch of 12 800 jobs.  With few jobs in
// flight (closed-loop batches, split windows) four waves per job finish a job sooner and keep the chip fuller: mcorr_launch picks per launch.
#include "multicorrelator.h"
#ifdef GSH_MC_VARIANT_128
#define mcorr_kernel mcorr_kernel_t128
#define mcorr_reduce_partials mcorr_reduce_partials_t128
#endif
#include "mcorr_device.h"
#include <cmath>
#include <cstdlib>

namespace gsh
{
namespace
{
using namespace mcdev;
#ifndef GSH_MC_MIN_WAVES
#define GSH_MC_MIN_WAVES 5  // E/P/L: the packed body holds four accumulator sets; <= 96 VGPRs (5 waves per SIMD), the kernel is VALU-issue bound
#endif

#ifndef GSH_MC_RUN_LEN
#define GSH_MC_RUN_LEN 8
#endif
constexpr int MC_RUN_LEN = GSH_MC_RUN_LEN;  // samples per lane run of the run-based path (mcorr_device.h)

// WIN: the launch stages per-segment windows of the codes (a.window_floats > 0); otherwise every work-group stages its whole code at the start of the
// LDS and the look-ups use a constant offset
// PAIR: every job of the launch is pair_eligible (the host checked, a.pair): the early tap is read next to the late one
template <int NT, int MODE, bool AUX, bool RUNS = false, bool WIN = false, bool PAIR = false>
__global__ __launch_bounds__(MC_THREADS, ((NT <= 3 && !AUX) ? GSH_MC_MIN_WAVES : 1)) void mcorr_kernel(McorrArgs a)
{
    static_assert(!PAIR || (NT == 3 && MODE == 0 && !AUX && !RUNS), "paired taps: the plain E/P/L launch");
    static_assert(!RUNS || (MODE == 0 && !AUX), "the run-based path exists for the standard mode without a fused tap");
    extern __shared__ __align__(16) float lds[];
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int slot = static_cast<int>(lb) / a.splits;
    const int split = static_cast<int>(lb) - slot * a.splits;
    const int job = a.job_list ? a.job_list[slot] : slot;
    const gsh_corr_job& J = a.jobs[job];
    const int tid = threadIdx.x;
    const int aux_job = AUX ? a.aux[job] : -1;
    if (AUX && aux_job == -2) return;  // computed (and its output row written) by the job in front of it

    JobCtx c;
    c.n_total = J.n_samples;
    c.code_len = a.code_lens[J.code_slot];
    c.rem_carr = J.rem_carr_phase_rad;
    c.phase_step = J.phase_step_rad;
    c.phase_rate = J.phase_rate_step_rad;
    c.rem_code = J.rem_code_phase_chips;
    c.code_step = J.code_phase_step_chips;
    c.code_rate = J.code_phase_rate_step_chips;
    c.packed = a.packed != 0;
    c.runs = RUNS && a.packed == 2;

    // ---- this work-group's slice of the window
    int seg = (c.n_total + a.splits - 1) / a.splits;
    seg = (seg + 1) & ~1;
    c.n_begin = min(split * seg, c.n_total);
    c.n_end = min(c.n_total, c.n_begin + seg);
    unsigned long long win0 = J.sample_offset + a.sample_base;
    if (a.ring_capacity) win0 %= a.ring_capacity;
    const unsigned long long abs0 = win0 + static_cast<unsigned long long>(c.n_begin);
    const int odd = static_cast<int>(abs0 & 1ULL);
    c.n_first = c.n_begin - odd;
    const float2* __restrict__ base = a.stream + (abs0 - static_cast<unsigned long long>(odd));  // 16-byte aligned

    float sh[NT];
    int rot[NT];
#pragma unroll
    // unused slots (jobs with fewer taps than the kernel flavour) repeat the last real shift: the index range [lo, hi] below then equals the
    // one the host sized the code window for (bank_window_floats spans the real taps only); their sums are never stored
    for (int t = 0; t < NT; t++) sh[t] = J.shifts_chips[min(t, max(J.n_taps, 1) - 1)];
    rot[0] = 0;
    if (mode_hd_code(MODE))
        {
            // K/..high_dynamics_resampler..:82-85: shift_samples += (int)round((shift[t]-shift[t-1])/step)
            unsigned accum = 0;
#pragma unroll
            for (int t = 1; t < NT; t++)
                {
                    if (t < J.n_taps) accum += static_cast<unsigned>(static_cast<int>(roundf(__fdiv_rn(__fsub_rn(sh[t], sh[t - 1]), c.code_step))));
                    rot[t] = static_cast<int>(accum);
                }
        }
    else
        {
#pragma unroll
            for (int t = 1; t < NT; t++) rot[t] = 0;
        }

    // the raw chip index is monotone in n and in the shift (every rounding is monotone) when step >= 0: its range over this
    // segment is [lo, hi], evaluated with the very expressions the samples use
    int lo = 0, hi = -1;
    if (!mode_hd_code(MODE) && c.n_end > c.n_begin)
        {
            float smin = sh[0], smax = sh[0];
#pragma unroll
            for (int t = 1; t < NT; t++)
                {
                    smin = fminf(smin, sh[t]);
                    smax = fmaxf(smax, sh[t]);
                }
            lo = raw_chip_std(__fmul_rn(c.code_step, static_cast<float>(c.n_begin)), smin, c.rem_code);
            hi = raw_chip_std(__fmul_rn(c.code_step, static_cast<float>(c.n_end - 1)), smax, c.rem_code);
        }

    // ---- stage the local code in LDS: the whole code (+ guard bands holding the wrapped neighbours), or -- when the host found that
    // every segment of this launch touches only a short run of it (long codes, split windows) -- just the samples lo..hi, so that a
    // 10 230-chip code does not cost 41 KB of LDS per work-group and with it most of the compute unit's occupancy
    float* tab = lds;
    const float* __restrict__ gcode = a.codes + static_cast<size_t>(J.code_slot) * a.code_stride;
    bool windowed = false, misfit = false, aux_fast = true;
    int lds_floats;
    typedef const __attribute__((address_space(3))) float* lds_float_ptr;
    const bool pair_ok = !PAIR || gsh::mcorr_pair_eligible(J.n_taps, J.shifts_chips, J.code_phase_step_chips, J.high_dyn);
    if (WIN != (a.window_floats > 0) || !pair_ok || (!WIN && reinterpret_cast<size_t>((lds_float_ptr)lds) != 0))  // (this kernel has no static LDS: the dynamic array starts at 0)
        {
            misfit = (c.n_end > c.n_begin);  // cannot happen: the launcher picks the flavour from the same field; reported as NaN if it does
            lds_floats = a.window_floats > 0 ? a.window_floats : c.code_len + 2 * MC_MARGIN;
        }
    else if (a.window_floats > 0)
        {
            const long long span = static_cast<long long>(hi) - lo + 1;
            if (!mode_hd_code(MODE) && c.code_step >= 0.0f && span >= 1 && span <= a.window_floats)
                {
                    windowed = true;
                    for (int i = tid; i < static_cast<int>(span); i += MC_THREADS) tab[i] = gcode[wrap_chip(lo + i, c.code_len)];
                    c.k_lo = lo;
                    c.k_hi = hi;
                    c.k_off = -lo;
                }
            else
                misfit = (c.n_end > c.n_begin);  // cannot happen for a batch the host admitted (bank_window_floats); reported as NaN if it does
            lds_floats = a.window_floats;
        }
    else
        {
            // the code itself: straight copies (no wrap arithmetic); then the two guard bands, one element per thread of the first wave
            // (the single loop with a wrap per element cost ~150 VALU instructions per wave -- 7 % of everything a wave executes for a 25 000-sample window)
            const int tab_len = c.code_len + 2 * MC_MARGIN;
            for (int j = tid; j < c.code_len; j += MC_THREADS) tab[MC_MARGIN + j] = gcode[j];
            if (tid < 2 * MC_MARGIN)
                {
                    const int k = tid < MC_MARGIN ? tid - MC_MARGIN : c.code_len + (tid - MC_MARGIN);  // -MARGIN .. -1, len .. len + MARGIN - 1
                    tab[MC_MARGIN + k] = gcode[wrap_margin(k, c.code_len)];
                }
            lds_floats = AUX ? a.code_stride + 2 * MC_MARGIN : tab_len;
        }
    // ---- the fused correlator's code (AUX): a second table behind the first
    if (AUX && aux_job >= 0 && !misfit)
        {
            const gsh_corr_job& J2 = a.jobs[aux_job];
            const float* __restrict__ gcode2 = a.codes + static_cast<size_t>(J2.code_slot) * a.code_stride;
            c.aux_on = true;
            c.aux_shift = J2.shifts_chips[0];
            c.aux_zero = (c.aux_shift == 0.0f);
            c.aux_code_len = a.code_lens[J2.code_slot];
            const int aux_base = (lds_floats + 3) & ~3;
            int lo2 = 0, hi2 = -1;
            if (c.n_end > c.n_begin)
                {
                    lo2 = raw_chip_std(__fmul_rn(c.code_step, static_cast<float>(c.n_begin)), c.aux_shift, c.rem_code);
                    hi2 = raw_chip_std(__fmul_rn(c.code_step, static_cast<float>(c.n_end - 1)), c.aux_shift, c.rem_code);
                }
            if (a.window_floats > 0)
                {
                    const long long span2 = static_cast<long long>(hi2) - lo2 + 1;
                    if (windowed && span2 >= 1 && span2 <= a.window_floats)
                        {
                            for (int i = tid; i < static_cast<int>(span2); i += MC_THREADS) tab[aux_base + i] = gcode2[wrap_chip(lo2 + i, c.aux_code_len)];
                            c.aux_k_lo = lo2;
                            c.aux_k_hi = hi2;
                            c.aux_k_off = aux_base - lo2;
                        }
                    else
                        misfit = (c.n_end > c.n_begin);
                }
            else
                {
                    const int tab_len2 = c.aux_code_len + 2 * MC_MARGIN;
                    for (int i = tid; i < tab_len2; i += MC_THREADS) tab[aux_base + i] = gcode2[wrap_margin(i - MC_MARGIN, c.aux_code_len)];
                    c.aux_k_off = aux_base + MC_MARGIN;
                    aux_fast = (lo2 >= -MC_MARGIN) && (hi2 < c.aux_code_len + MC_MARGIN) && (c.aux_code_len >= MC_MARGIN);
                }
            lds_floats = aux_base + (a.window_floats > 0 ? a.window_floats : a.code_stride + 2 * MC_MARGIN);
        }
    else if (AUX)
        lds_floats = ((lds_floats + 3) & ~3) + (a.window_floats > 0 ? a.window_floats : a.code_stride + 2 * MC_MARGIN);  // same `red` place for every work-group
    float2* red = reinterpret_cast<float2*>(lds + ((lds_floats + 3) & ~3));
    // ---- the factors of every lane's carrier seeds for this job (mcorr_device.h fac_table_fill): one evaluation per lane of the first wave, beside the staging
    float2* const fac = red + MC_WAVES * GSH_MAX_TAPS;
    if constexpr (MODE == 0 && MC_THREADS <= 256)
        {
            if (a.packed != 0 && a.fac != 0)
                {
                    if ((tid >> 6) == 0) fac_table_fill<2>(fac, c.phase_step, c.rem_carr, c.n_first, tid);
                    c.fac_tab = fac;
                }
        }

    float2 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = make_float2(0.0f, 0.0f);
    float2 acc_aux = make_float2(0.0f, 0.0f);

    __syncthreads();  // code table(s) visible

    if (misfit)
        {
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = make_float2(__builtin_nanf(""), __builtin_nanf(""));
            acc_aux = make_float2(__builtin_nanf(""), __builtin_nanf(""));
        }
    else if (c.n_end > c.n_begin)
        {
            // skip the per-sample wrap when the indices stay inside what is staged
            const bool fast = windowed || (!mode_hd_code(MODE) && (c.code_step >= 0.0f) && (lo >= -MC_MARGIN) && (hi < c.code_len + MC_MARGIN) && (c.code_len >= MC_MARGIN) && aux_fast);
            // centre tap at exactly 0 (E/P/L, VE/E/P/L/VL): its (a + 0.0f) is skipped; needs sample indices exact in float
            const bool zp = (NT & 1) && (NT == J.n_taps) && (sh[NT / 2] == 0.0f) && !mode_hd_code(MODE) && (c.n_total < (1 << 24));
            if (RUNS && fast && c.runs && c.code_step > 1.0e-6f && c.n_total < (1 << 24))
                {
                    if constexpr (RUNS)
                        {
                            float* const runs_lds = reinterpret_cast<float*>(fac + FAC_ENTRIES) + (tid >> 6) * RunsLayout<MC_RUN_LEN>::FLOATS;
                            run_segment_runs<NT, MC_RUN_LEN>(c, base, tab, sh, acc, runs_lds);
                        }
                }
            else if (fast && zp)
                run_segment<NT, MODE, false, true, AUX, !WIN, PAIR>(c, base, tab, sh, rot, acc, &acc_aux);
            else if (fast)
                run_segment<NT, MODE, false, false, AUX, !WIN>(c, base, tab, sh, rot, acc, &acc_aux);
            else
                run_segment<NT, MODE, true, false, AUX>(c, base, tab, sh, rot, acc, &acc_aux);
        }

    // ---- integrate-and-dump: wave64 prefix sum in DPP steps (the last lane holds the wave's sum; six v_add_f32_dpp per value instead of the
    // ds_bpermute + address + add of a shuffle tree), then one LDS step over the 4 waves
    {
        float sums[2 * NT + (AUX ? 2 : 0)];
#pragma unroll
        for (int t = 0; t < NT; t++)
            {
                sums[2 * t] = acc[t].x;
                sums[2 * t + 1] = acc[t].y;
            }
        if (AUX)
            {
                sums[2 * NT] = acc_aux.x;
                sums[2 * NT + 1] = acc_aux.y;
            }
        wave_scan_incl_n(sums);
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t] = make_float2(sums[2 * t], sums[2 * t + 1]);
        if (AUX) acc_aux = make_float2(sums[2 * NT], sums[2 * NT + 1]);
    }
    const int wave = tid >> 6;
    if ((tid & 63) == 63)
        {
#pragma unroll
            for (int t = 0; t < NT; t++) red[wave * GSH_MAX_TAPS + t] = acc[t];
            if (AUX) red[wave * GSH_MAX_TAPS + NT] = acc_aux;  // NT < GSH_MAX_TAPS in the AUX kernels
        }
    __syncthreads();
    if (AUX && aux_job >= 0 && tid >= GSH_MAX_TAPS && tid < 2 * GSH_MAX_TAPS)
        {
            // the fused job's whole output row: its one tap, zeros behind it
            const int tap = tid - GSH_MAX_TAPS;
            float2 s = make_float2(0.0f, 0.0f);
            if (tap == 0)
                {
#pragma unroll
                    for (int w = 0; w < MC_WAVES; w++)
                        {
                            s.x += red[w * GSH_MAX_TAPS + NT].x;
                            s.y += red[w * GSH_MAX_TAPS + NT].y;
                        }
                }
            if (a.splits == 1)
                a.out[static_cast<size_t>(aux_job) * GSH_MAX_TAPS + tap] = s;
            else
                a.partials[(static_cast<size_t>(aux_job) * a.splits + split) * GSH_MAX_TAPS + tap] = s;
        }
    if (tid < GSH_MAX_TAPS)
        {
            float2 s = make_float2(0.0f, 0.0f);
            if (tid < NT && tid < J.n_taps)
                {
#pragma unroll
                    for (int w = 0; w < MC_WAVES; w++)
                        {
                            s.x += red[w * GSH_MAX_TAPS + tid].x;
                            s.y += red[w * GSH_MAX_TAPS + tid].y;
                        }
                }
            if (a.splits == 1)
                a.out[static_cast<size_t>(job) * GSH_MAX_TAPS + tid] = s;
            else
                a.partials[(static_cast<size_t>(job) * a.splits + split) * GSH_MAX_TAPS + tid] = s;
        }
}

// sums the per-split partials of each job in split order (deterministic)
__global__ __launch_bounds__(256) void mcorr_reduce_partials(const float2* __restrict__ partials, float2* __restrict__ out, int n_jobs, int splits)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;  // job*GSH_MAX_TAPS + tap
    if (i >= n_jobs * GSH_MAX_TAPS) return;
    const int job = i / GSH_MAX_TAPS, tap = i - job * GSH_MAX_TAPS;
    float2 s = make_float2(0.0f, 0.0f);
    for (int k = 0; k < splits; k++)
        {
            const float2 p = partials[(static_cast<size_t>(job) * splits + k) * GSH_MAX_TAPS + tap];
            s.x += p.x;
            s.y += p.y;
        }
    out[i] = s;
}

template <int NT>
int launch_nt(const McorrArgs& a, int mode, size_t lds, hipStream_t stream)
{
    const dim3 grid(static_cast<unsigned>(a.n_launch) * static_cast<unsigned>(a.splits));
    const dim3 block(MC_THREADS);
    if (a.aux != nullptr)
        {
            if constexpr (NT == 3 || NT == 5)
                {
                    GSH_REQUIRE(mode == 0, "fused jobs need the standard mode");
                    if (a.window_floats > 0)
                        hipLaunchKernelGGL((mcorr_kernel<NT, 0, true, false, true>), grid, block, lds, stream, a);
                    else
                        hipLaunchKernelGGL((mcorr_kernel<NT, 0, true>), grid, block, lds, stream, a);
                    GSH_HIP(hipGetLastError());
                    return GSH_OK;
                }
            else
                return set_error(GSH_ERR_INVALID, "fused jobs exist only for the 3- and 5-tap kernels");
        }
    switch (mode)
        {
        case 0:
#ifdef GSH_MC_RUNS_EXPERIMENT  // the run-based body (round 2, slower, 16 - 32 B of scratch per thread): only in a library built with this macro (profiles/ab/build_variant.py)
            if constexpr (NT <= 5)
                {
                    if (a.packed == 2 && a.window_floats == 0)
                        {
                            const size_t lds_runs = lds + static_cast<size_t>(MC_WAVES) * RunsLayout<MC_RUN_LEN>::FLOATS * sizeof(float);
                            if (lds_runs <= 64 * 1024)  // beyond that the scratch costs more occupancy than the path gains
                                {
                                    hipLaunchKernelGGL((mcorr_kernel<NT, 0, false, true>), grid, block, lds_runs, stream, a);
                                    break;
                                }
                        }
                }
#endif
            if constexpr (NT == 3)
                {
                    if (a.pair && a.packed == 1)
                        {
                            if (a.window_floats > 0)
                                hipLaunchKernelGGL((mcorr_kernel<NT, 0, false, false, true, true>), grid, block, lds, stream, a);
                            else
                                hipLaunchKernelGGL((mcorr_kernel<NT, 0, false, false, false, true>), grid, block, lds, stream, a);
                            break;
                        }
                }
            if (a.window_floats > 0)
                hipLaunchKernelGGL((mcorr_kernel<NT, 0, false, false, true>), grid, block, lds, stream, a);
            else
                hipLaunchKernelGGL((mcorr_kernel<NT, 0, false>), grid, block, lds, stream, a);
            break;
        case 1:
            hipLaunchKernelGGL((mcorr_kernel<NT, 1, false>), grid, block, lds, stream, a);
            break;
        case 2:
            hipLaunchKernelGGL((mcorr_kernel<NT, 2, false>), grid, block, lds, stream, a);
            break;
        default:
            return set_error(GSH_ERR_INVALID, "unknown correlator mode %d", mode);
        }
    GSH_HIP(hipGetLastError());
    return GSH_OK;
}
}  // namespace

#ifndef GSH_MC_VARIANT_128
int mcorr_packed_default()
{
    // GSH_MC_PACKED_BODY: 0 the round-1 body, 1 the packed trips, 2 the run-based path where a job qualifies, 3 the packed trips with the derived
    // early / late taps switched off (A/B switch, read once)
    static const int v = [] {
        const char* e = std::getenv("GSH_MC_PACKED_BODY");
        if (e != nullptr && e[0] == '0') return 0;
#ifdef GSH_MC_RUNS_EXPERIMENT
        if (e != nullptr && e[0] == '2') return 2;
#endif
        if (e != nullptr && e[0] == '3') return 3;  // packed trips without the derived early / late taps
        return 1;
    }();
    return v;
}

int mcorr_fac_default()
{
    static const int v = [] {
        const char* e = std::getenv("GSH_MC_FAC");
        return (e != nullptr && e[0] == '0') ? 0 : 1;
    }();
    return v;
}

#endif  // !GSH_MC_VARIANT_128

#ifdef GSH_MC_VARIANT_128
#define mcorr_lds_bytes_window mcorr_lds_bytes_window_t128
#define mcorr_lds_bytes_fused mcorr_lds_bytes_fused_t128
#define mcorr_lds_bytes mcorr_lds_bytes_t128
#define mcorr_launch mcorr_launch_t128
#define mcorr_launch_classes mcorr_launch_classes_t128
namespace
{
#endif
size_t mcorr_lds_bytes_window(int window_floats)
{
    const size_t tab = (static_cast<size_t>(window_floats) + 3) & ~static_cast<size_t>(3);
    return tab * sizeof(float) + (MC_WAVES * GSH_MAX_TAPS + FAC_ENTRIES) * sizeof(float2);
}

size_t mcorr_lds_bytes_fused(int max_code_len, int window_floats)
{
    const size_t one = window_floats > 0 ? static_cast<size_t>(window_floats) : static_cast<size_t>(max_code_len) + 2 * MC_MARGIN;
    const size_t tab = ((one + 3) & ~static_cast<size_t>(3)) + ((one + 3) & ~static_cast<size_t>(3));
    return tab * sizeof(float) + (MC_WAVES * GSH_MAX_TAPS + FAC_ENTRIES) * sizeof(float2);
}

size_t mcorr_lds_bytes(int max_code_len)
{
    const size_t tab = (static_cast<size_t>(max_code_len) + 2 * MC_MARGIN + 3) & ~static_cast<size_t>(3);
    return tab * sizeof(float) + (MC_WAVES * GSH_MAX_TAPS + FAC_ENTRIES) * sizeof(float2);
}

#ifdef GSH_MC_VARIANT_128
}  // namespace
#else
// work-group size of a launch: 128 threads (two waves per job) for the E/P/L whole-code flavours once the launch holds at least two rounds of them (2 x 2 560
// work-groups: a job then takes twice as long, and a chip that is not full wants the shorter jobs); GSH_MC_WG=128 / 256 forces one (A/B runs)
bool use_128(const McorrArgs& a, int max_taps, int mode)
{
    static const int forced = [] {
        const char* e = std::getenv("GSH_MC_WG");
        return e != nullptr ? std::atoi(e) : 0;
    }();
    const bool possible = mode == 0 && a.aux == nullptr && a.window_floats == 0 && max_taps >= 2 && max_taps <= 3 && a.packed == 1;
    if (!possible || forced == 256) return false;
    if (forced == 128) return true;
    return static_cast<long long>(a.n_launch) * a.splits >= 2 * 2560;
}
#endif

int mcorr_launch(const McorrArgs& a, int max_taps, int mode, int max_code_len, hipStream_t stream)
{
    if (a.n_jobs <= 0) return GSH_OK;
#ifndef GSH_MC_VARIANT_128
    if (use_128(a, max_taps, mode)) return mcorr_launch_t128(a, max_taps, mode, max_code_len, stream);
#endif
    GSH_REQUIRE(max_taps >= 1 && max_taps <= GSH_MAX_TAPS, "n_taps %d outside 1..%d", max_taps, GSH_MAX_TAPS);
    GSH_REQUIRE(a.splits >= 1, "splits must be >= 1");
    const size_t lds = a.aux != nullptr ? mcorr_lds_bytes_fused(max_code_len, a.window_floats)
                                        : (a.window_floats > 0 ? mcorr_lds_bytes_window(a.window_floats) : mcorr_lds_bytes(max_code_len));
    GSH_REQUIRE(lds <= 160 * 1024, "local code of %d samples does not fit the 160 KiB LDS", max_code_len);
    int rc;
    if (max_taps == 1)
        rc = launch_nt<1>(a, mode, lds, stream);
    else if (max_taps <= 3)
        rc = launch_nt<3>(a, mode, lds, stream);
    else if (max_taps <= 5)
        rc = launch_nt<5>(a, mode, lds, stream);
    else
        rc = launch_nt<GSH_MAX_TAPS>(a, mode, lds, stream);
    if (rc != GSH_OK) return rc;
    if (a.splits > 1)
        {
            const int total = a.n_jobs * GSH_MAX_TAPS;
            hipLaunchKernelGGL(mcorr_reduce_partials, dim3((total + 255) / 256), dim3(256), 0, stream, a.partials, a.out, a.n_jobs, a.splits);
            GSH_HIP(hipGetLastError());
        }
    return GSH_OK;
}

int mcorr_launch_classes(const McorrArgs& args, const McorrClassPlan& plan, int mode, int max_code_len, hipStream_t stream)
{
    if (args.n_jobs <= 0) return GSH_OK;
    GSH_REQUIRE(args.splits >= 1, "splits must be >= 1");
    static const int class_taps[4] = {1, 3, 5, GSH_MAX_TAPS};
    for (int c = 0; c < 4; c++)
        {
            if (plan.count[c] <= 0) continue;
            McorrArgs a = args;
            a.job_list = plan.list + plan.offset[c];
            a.n_launch = plan.count[c];
            if (!plan.aux[c]) a.aux = nullptr;
#ifndef GSH_MC_VARIANT_128
            if (use_128(a, class_taps[c], mode))
                {
                    // this class alone through the two-wave kernels (the partials, if any, are summed once below)
                    McorrClassPlan one = plan;
                    for (int o = 0; o < 4; o++)
                        if (o != c) one.count[o] = 0;
                    McorrArgs b1 = args;
                    b1.splits = args.splits;
                    const int rc1 = mcorr_launch_classes_t128(b1, one, mode, max_code_len, stream);
                    if (rc1 != GSH_OK) return rc1;
                    continue;
                }
#endif
            const size_t lds = a.aux != nullptr ? mcorr_lds_bytes_fused(max_code_len, a.window_floats)
                                                : (a.window_floats > 0 ? mcorr_lds_bytes_window(a.window_floats) : mcorr_lds_bytes(max_code_len));
            GSH_REQUIRE(lds <= 160 * 1024, "local code of %d samples does not fit the 160 KiB LDS", max_code_len);
            int rc;
            switch (class_taps[c])
                {
                case 1:
                    rc = launch_nt<1>(a, mode, lds, stream);
                    break;
                case 3:
                    rc = launch_nt<3>(a, mode, lds, stream);
                    break;
                case 5:
                    rc = launch_nt<5>(a, mode, lds, stream);
                    break;
                default:
                    rc = launch_nt<GSH_MAX_TAPS>(a, mode, lds, stream);
                    break;
                }
            if (rc != GSH_OK) return rc;
        }
    if (args.splits > 1)
        {
            const int total = args.n_jobs * GSH_MAX_TAPS;
            hipLaunchKernelGGL(mcorr_reduce_partials, dim3((total + 255) / 256), dim3(256), 0, stream, args.partials, args.out, args.n_jobs, args.splits);
            GSH_HIP(hipGetLastError());
        }
    return GSH_OK;
}
}  // namespace gsh
