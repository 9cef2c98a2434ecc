// Device-side building blocks of the multicorrelator (shared by csrc/multicorrelator.hip -- the batched open-loop
// kernel -- and csrc/tracking_loop.hip -- the closed DLL/PLL loop).  See multicorrelator.hip for the design notes and
// the reference citations.  Everything here is __device__ __forceinline__ code in namespace gsh::mcdev.
#ifndef GSH_MCORR_DEVICE_H
#define GSH_MCORR_DEVICE_H

#include "gsh_internal.h"
#include <cmath>

// Work-group size of the correlator code below.  A translation unit may define GSH_MC_THREADS (a multiple of 64, <= 1024)
// before including this header; the code then lives in its own namespace (mcdev_<threads>) so that two translation
// units with different sizes never define the same entity differently.
#ifndef GSH_MC_THREADS
#define GSH_MC_THREADS 256
#endif
#define GSH_MC_NS_CAT2(a, b) a##b
#define GSH_MC_NS_CAT(a, b) GSH_MC_NS_CAT2(a, b)
#define GSH_MC_NS GSH_MC_NS_CAT(mcdev_, GSH_MC_THREADS)

namespace gsh
{
namespace GSH_MC_NS
{
constexpr int MC_THREADS = GSH_MC_THREADS;
constexpr int MC_WAVES = MC_THREADS / 64;
constexpr int MC_MARGIN = 32;  // guard entries on each side of the LDS code table
#ifndef GSH_MC_RESEED
#define GSH_MC_RESEED 32
#endif
#ifndef GSH_MC_CVT_FLR
#define GSH_MC_CVT_FLR 1
#endif
constexpr int MC_RESEED = GSH_MC_RESEED;  // strides of 512 samples between exact NCO re-seeds
constexpr int MC_PAIRS_PER_CHUNK = MC_THREADS;  // one float4 (2 samples) per thread per chunk
constexpr double INV_TWO_PI = 0.15915494309189533576888376337251436;
constexpr double TWO_PI_D = 6.283185307179586476925286766559;

// job mode bits (gsh_corr_job::high_dyn): 0 std/std, 1 hd resampler + hd rotator,
// 2 hd resampler + std rotator (the 6-argument overload, mcorr.cc:129-144, with the flag set)
__host__ __device__ constexpr bool mode_hd_code(int mode) { return mode != 0; }
__host__ __device__ constexpr bool mode_hd_phase(int mode) { return mode == 1; }

__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    return make_float2(fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x));
}

// exp(-j*phase), phase given in double radians
__device__ __forceinline__ float2 expmj(double phase)
{
    double rev = phase * INV_TWO_PI;
    rev -= rint(rev);  // [-0.5, 0.5]
    const float r = static_cast<float>(rev * TWO_PI_D);
    float s, c;
    sincosf(r, &s, &c);
    return make_float2(c, -s);
}

// carrier phase (radians, double) of sample n.
// standard: rem + n*step (mcorr.cc:115,123: phase0 = exp(-j rem), inc = exp(-j step)).
// high dynamics: + rate*(float)((n-1)^2) for n >= 1: the rate factor computed in iteration
// n-1 from (unsigned)(n-1)*(n-1) is the one applied to sample n (K/..high_dynamic_rotator..:94-103).
template <bool HDP>
__device__ __forceinline__ double carrier_phase(float rem, float step, float rate, int n)
{
    double ph = static_cast<double>(rem) + static_cast<double>(n) * static_cast<double>(step);
    if (HDP)
        {
            if (n > 0)
                {
                    const unsigned m = static_cast<unsigned>(n - 1);
                    ph += static_cast<double>(rate) * static_cast<double>(static_cast<float>(m * m));
                }
        }
    return ph;
}

// mathematical modulo, same result as K/..resampler_32f_xn.h:75-76
__device__ __forceinline__ int wrap_chip(int k, int len)
{
    if (static_cast<unsigned>(k) >= static_cast<unsigned>(len))
        {
            k %= len;
            if (k < 0) k += len;
        }
    return k;
}

// (int)floor(x) in one VALU instruction (v_cvt_flr_i32_f32: round toward -inf, then convert)
__device__ __forceinline__ int floor_to_int(float x)
{
#if GSH_MC_CVT_FLR
    int k;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(k) : "v"(x));
    return k;
#else
    return static_cast<int>(floorf(x));
#endif
}

// raw (unwrapped) chip index, standard resampler: floor((step*(float)n + shift) - rem)
__device__ __forceinline__ int raw_chip_std(float step_x_n, float shift, float rem)
{
    return floor_to_int(__fsub_rn(__fadd_rn(step_x_n, shift), rem));
}

// raw chip index, high-dynamics resampler tap 0 expression evaluated at sample m:
// floor(((step*(float)m + rate*(float)(m*m)) + shift0) - rem), m*m in unsigned
__device__ __forceinline__ int raw_chip_hd(float step, float rate, unsigned m, float shift0, float rem)
{
    const float a = __fmul_rn(step, static_cast<float>(m));
    const float q = __fmul_rn(rate, static_cast<float>(m * m));
    return floor_to_int(__fsub_rn(__fadd_rn(__fadd_rn(a, q), shift0), rem));
}

struct JobCtx
{
    int n_total;     // job n_samples
    int n_begin;     // this work-group's segment [n_begin, n_end)
    int n_end;
    int n_first;     // sample index of pair 0's first element (n_begin or n_begin-1)
    int code_len;
    float rem_carr, phase_step, phase_rate;
    float rem_code, code_step, code_rate;
    // windowed code table (multicorrelator.hip): only the code samples k_lo..k_hi this segment can touch are staged, and `tab` is
    // indexed through k_off.  k_hi < k_lo: the whole code is staged (no window).
    int k_lo{0}, k_hi{-1};
    int k_off{MC_MARGIN};  // tab[k + k_off] is code sample k (window: -k_lo)
    // fused second correlator (AUX kernels): one more tap over the same rotated samples with ANOTHER code -- the data-component prompt that
    // track_pilot adds to a pilot channel (trk.cc:1246-1256), which the reference runs as a second pass over the window
    bool aux_on{false};
    bool aux_zero{false};   // its shift is exactly 0.0f: on the ZP path it shares the prompt tap's chip index
    float aux_shift{0.0f};
    int aux_code_len{1};
    int aux_k_off{MC_MARGIN};
    int aux_k_lo{0}, aux_k_hi{-1};
};

// One chunk = 256 pairs = 512 consecutive samples; thread `tid` owns samples n0, n0+1.
// ZP: the centre tap's shift is exactly 0.0f (the prompt of an E/P/L or VE/E/P/L/VL set): (a + 0.0f) == a, so its add is skipped.
// nf0 = (float)n0, maintained by the caller (exact: sample indices stay below 2^24).
template <int NT, int MODE, bool WRAP, bool MASKED, bool ZP = false, bool AUX = false>
__device__ __forceinline__ void process_pair(const JobCtx& c, const float2* __restrict__ base, const float* __restrict__ tab,
    const float (&sh)[NT], const int (&rot)[NT], int pair, float2 pa, float2 pb, float2 (&acc)[NT], float nf0, float2* acc_aux = nullptr)
{
    const int n0 = c.n_first + 2 * pair;
    float2 x0, x1;
    if (MASKED)
        {
            const bool v0 = (n0 >= c.n_begin) && (n0 < c.n_end);
            const bool v1 = (n0 + 1 >= c.n_begin) && (n0 + 1 < c.n_end);
            x0 = v0 ? base[2 * pair] : make_float2(0.0f, 0.0f);
            x1 = v1 ? base[2 * pair + 1] : make_float2(0.0f, 0.0f);
        }
    else
        {
            const float4 v = *reinterpret_cast<const float4*>(base + 2 * pair);
            x0 = make_float2(v.x, v.y);
            x1 = make_float2(v.z, v.w);
        }
    const float2 y0 = cmul(x0, pa);
    const float2 y1 = cmul(x1, pb);

    if (!mode_hd_code(MODE))
        {
            const float a0 = __fmul_rn(c.code_step, nf0);
            // (float)(n0 + 1): on the ZP path (selected only for windows shorter than 2^24 samples) nf0 + 1.0f is exactly that value
            const float nf1 = ZP ? __fadd_rn(nf0, 1.0f) : static_cast<float>(n0 + 1);
            const float a1 = __fmul_rn(c.code_step, nf1);
            int kz0 = 0, kz1 = 0;  // the zero-shift prompt's raw indices (ZP), shared with a zero-shift fused tap
#pragma unroll
            for (int t = 0; t < NT; t++)
                {
                    int k0, k1;
                    if (ZP && t == NT / 2)
                        {
                            k0 = floor_to_int(__fsub_rn(a0, c.rem_code));
                            k1 = floor_to_int(__fsub_rn(a1, c.rem_code));
                            kz0 = k0;
                            kz1 = k1;
                        }
                    else
                        {
                            k0 = raw_chip_std(a0, sh[t], c.rem_code);
                            k1 = raw_chip_std(a1, sh[t], c.rem_code);
                        }
                    if (WRAP)
                        {
                            k0 = wrap_chip(k0, c.code_len);
                            k1 = wrap_chip(k1, c.code_len);
                        }
                    if (MASKED)
                        {
                            // masked lanes may sit at n = -1 / n = n_end with any index: keep the lookup in range (their sample is 0, the
                            // value read only has to be finite).  In-range lanes are untouched by either form.
                            if (c.k_hi >= c.k_lo)
                                {
                                    k0 = min(max(k0, c.k_lo), c.k_hi);
                                    k1 = min(max(k1, c.k_lo), c.k_hi);
                                }
                            else
                                {
                                    k0 = wrap_chip(k0, c.code_len);
                                    k1 = wrap_chip(k1, c.code_len);
                                }
                        }
                    const float c0 = tab[k0 + c.k_off];
                    const float c1 = tab[k1 + c.k_off];
                    acc[t].x = fmaf(y0.x, c0, acc[t].x);
                    acc[t].y = fmaf(y0.y, c0, acc[t].y);
                    acc[t].x = fmaf(y1.x, c1, acc[t].x);
                    acc[t].y = fmaf(y1.y, c1, acc[t].y);
                }
            if (AUX && c.aux_on)
                {
                    // the fused correlator: same samples, same rotation, its own code table (and window) behind the first one in LDS
                    int k0, k1;
                    if (ZP && c.aux_zero)
                        {
                            k0 = kz0;
                            k1 = kz1;
                        }
                    else
                        {
                            k0 = raw_chip_std(a0, c.aux_shift, c.rem_code);
                            k1 = raw_chip_std(a1, c.aux_shift, c.rem_code);
                        }
                    if (WRAP)
                        {
                            k0 = wrap_chip(k0, c.aux_code_len);
                            k1 = wrap_chip(k1, c.aux_code_len);
                        }
                    if (MASKED)
                        {
                            if (c.aux_k_hi >= c.aux_k_lo)
                                {
                                    k0 = min(max(k0, c.aux_k_lo), c.aux_k_hi);
                                    k1 = min(max(k1, c.aux_k_lo), c.aux_k_hi);
                                }
                            else
                                {
                                    k0 = wrap_chip(k0, c.aux_code_len);
                                    k1 = wrap_chip(k1, c.aux_code_len);
                                }
                        }
                    const float c0 = tab[k0 + c.aux_k_off];
                    const float c1 = tab[k1 + c.aux_k_off];
                    acc_aux->x = fmaf(y0.x, c0, acc_aux->x);
                    acc_aux->y = fmaf(y0.y, c0, acc_aux->y);
                    acc_aux->x = fmaf(y1.x, c1, acc_aux->x);
                    acc_aux->y = fmaf(y1.y, c1, acc_aux->y);
                }
        }
    else
        {
#pragma unroll
            for (int t = 0; t < NT; t++)
                {
                    // tap t is tap 0 advanced circularly by rot[t] samples (K/..high_dynamics_resampler..:84-90)
                    int m0 = n0 + rot[t];
                    int m1 = n0 + 1 + rot[t];
                    if (m0 >= c.n_total) m0 -= c.n_total;
                    if (m1 >= c.n_total) m1 -= c.n_total;
                    if (MASKED)
                        {
                            if (m0 < 0) m0 = 0;
                            if (m1 >= c.n_total) m1 = 0;
                        }
                    const int k0 = wrap_chip(raw_chip_hd(c.code_step, c.code_rate, static_cast<unsigned>(m0), sh[0], c.rem_code), c.code_len);
                    const int k1 = wrap_chip(raw_chip_hd(c.code_step, c.code_rate, static_cast<unsigned>(m1), sh[0], c.rem_code), c.code_len);
                    const float c0 = tab[k0 + c.k_off];
                    const float c1 = tab[k1 + c.k_off];
                    acc[t].x = fmaf(y0.x, c0, acc[t].x);
                    acc[t].y = fmaf(y0.y, c0, acc[t].y);
                    acc[t].x = fmaf(y1.x, c1, acc[t].x);
                    acc[t].y = fmaf(y1.y, c1, acc[t].y);
                }
        }
}

template <int NT, int MODE, bool WRAP, bool ZP = false, bool AUX = false>
__device__ __forceinline__ void run_segment(const JobCtx& c, const float2* __restrict__ base, const float* __restrict__ tab,
    const float (&sh)[NT], const int (&rot)[NT], float2 (&acc)[NT], float2* acc_aux = nullptr)
{
    const int tid = threadIdx.x;
    const int span = c.n_end - c.n_first;          // samples covered from pair 0's first element
    const int n_pairs = (span + 1) >> 1;           // pairs touching the segment
    const int n_full = span >> 1;                  // leading pairs whose second sample is in range
    const int odd = c.n_begin - c.n_first;         // 1 when pair 0's first sample is outside
    const int n_chunks = (n_pairs + MC_PAIRS_PER_CHUNK - 1) / MC_PAIRS_PER_CHUNK;
    const int k_full_begin = odd ? 1 : 0;
    const int k_full_end = n_full / MC_PAIRS_PER_CHUNK;  // chunks [k_full_begin, k_full_end) need no masking
    constexpr bool HDP = mode_hd_phase(MODE);

    // masked head chunk (only when the window starts on an odd absolute sample)
    if (odd && n_chunks > 0)
        {
            const int pair = tid;
            if (pair < n_pairs)
                {
                    const int n0 = c.n_first + 2 * pair;
                    const float2 pa = expmj(carrier_phase<HDP>(c.rem_carr, c.phase_step, c.phase_rate, n0));
                    const float2 pb = expmj(carrier_phase<HDP>(c.rem_carr, c.phase_step, c.phase_rate, n0 + 1));
                    process_pair<NT, MODE, WRAP, true, false, AUX>(c, base, tab, sh, rot, pair, pa, pb, acc, static_cast<float>(n0), acc_aux);
                }
        }

    // unmasked body
    if (k_full_end > k_full_begin)
        {
            if (HDP)
                {
                    // chirped carrier: no constant-stride recurrence exists, evaluate per sample
                    for (int k = k_full_begin; k < k_full_end; k++)
                        {
                            const int pair = tid + k * MC_PAIRS_PER_CHUNK;
                            const int n0 = c.n_first + 2 * pair;
                            const float2 pa = expmj(carrier_phase<true>(c.rem_carr, c.phase_step, c.phase_rate, n0));
                            const float2 pb = expmj(carrier_phase<true>(c.rem_carr, c.phase_step, c.phase_rate, n0 + 1));
                            process_pair<NT, MODE, WRAP, false, false, AUX>(c, base, tab, sh, rot, pair, pa, pb, acc, static_cast<float>(n0), acc_aux);
                        }
                }
            else
                {
                    // stride rotator exp(-j * 512 * step) and sample rotator exp(-j * step), both seeded exactly
                    const float2 w = expmj(static_cast<double>(2 * MC_PAIRS_PER_CHUNK) * static_cast<double>(c.phase_step));
                    const float2 inc = expmj(static_cast<double>(c.phase_step));
                    for (int kb = k_full_begin; kb < k_full_end; kb += MC_RESEED)
                        {
                            const int cnt = min(MC_RESEED, k_full_end - kb);
                            int pair = tid + kb * MC_PAIRS_PER_CHUNK;
                            // exact re-seed of this lane's phasor; the second sample of the pair is one step further
                            float2 pa = expmj(carrier_phase<false>(c.rem_carr, c.phase_step, 0.0f, c.n_first + 2 * pair));
                            float2 pb = cmul(pa, inc);
                            // on the ZP path (windows shorter than 2^24 samples) (float)n is carried along and advanced by 2 * 256 per chunk,
                            // which is exact there; the general path converts every time, as the reference's (float)n does
                            float nf0 = static_cast<float>(c.n_first + 2 * pair);
#pragma unroll 2
                            for (int i = 0; i < cnt; i++)
                                {
                                    process_pair<NT, MODE, WRAP, false, ZP, AUX>(c, base, tab, sh, rot, pair, pa, pb, acc, nf0, acc_aux);
                                    pa = cmul(pa, w);
                                    pb = cmul(pb, w);
                                    pair += MC_PAIRS_PER_CHUNK;
                                    nf0 = ZP ? __fadd_rn(nf0, static_cast<float>(2 * MC_PAIRS_PER_CHUNK)) : static_cast<float>(c.n_first + 2 * pair);
                                }
                        }
                }
        }

    // masked tail chunks (at most two: a partially filled chunk and, when the body was empty, chunk 0)
    for (int k = max(k_full_end, k_full_begin); k < n_chunks; k++)
        {
            const int pair = tid + k * MC_PAIRS_PER_CHUNK;
            if (pair < n_pairs)
                {
                    const int n0 = c.n_first + 2 * pair;
                    const float2 pa = expmj(carrier_phase<HDP>(c.rem_carr, c.phase_step, c.phase_rate, n0));
                    const float2 pb = expmj(carrier_phase<HDP>(c.rem_carr, c.phase_step, c.phase_rate, n0 + 1));
                    process_pair<NT, MODE, WRAP, true, false, AUX>(c, base, tab, sh, rot, pair, pa, pb, acc, static_cast<float>(n0), acc_aux);
                }
        }
}


// stage a local code (+ guard bands holding the wrapped neighbours) in LDS: tab[i] = code[wrap(i - MC_MARGIN)]
__device__ __forceinline__ void stage_code_table(float* tab, const float* __restrict__ gcode, int code_len)
{
    const int tab_len = code_len + 2 * MC_MARGIN;
    for (int i = threadIdx.x; i < tab_len; i += MC_THREADS) tab[i] = gcode[wrap_chip(i - MC_MARGIN, code_len)];
}

// floats of LDS one staged code occupies (rounded to 16 bytes)
__host__ __device__ constexpr int code_table_floats(int code_len) { return (code_len + 2 * MC_MARGIN + 3) & ~3; }

// One correlation of the window [sample_offset, sample_offset + n) of `stream` by a whole work-group: the body of mcorr_kernel for
// splits == 1.  MODE as gsh_corr_job::high_dyn: 0 standard resampler + rotator, 1 the high-dynamics pair (phase_rate / code_rate used).  `tab` is a staged code table, `red` MC_WAVES *
// GSH_MAX_TAPS float2 of LDS scratch.  On return red[0..NT) holds the tap sums and every thread may read them
// (a __syncthreads() has been executed); the caller must __syncthreads() again before `red` is reused.
// AUX: one more tap with the code staged at `tab_aux` (same LDS allocation as `tab`, same length) and shift `aux_shift` is computed in the same
// pass -- the data-component prompt of track_pilot (trk.cc:1246-1256); its sum is returned in red[NT].  Standard mode only.
template <int NT, int MODE, bool AUX = false>
__device__ __forceinline__ void correlate_window(const float2* __restrict__ stream, unsigned long long sample_offset, int n_samples,
    const float* __restrict__ tab, int code_len, const float (&sh)[NT], float rem_carr, float phase_step, float phase_rate, float rem_code,
    float code_step, float code_rate, float2* __restrict__ red, const float* tab_aux = nullptr, float aux_shift = 0.0f)
{
    static_assert(!AUX || (MODE == 0 && NT < GSH_MAX_TAPS), "the fused tap exists for the standard mode and needs a free slot in `red`");
    const int tid = threadIdx.x;
    JobCtx c;
    c.n_total = n_samples;
    c.code_len = code_len;
    c.rem_carr = rem_carr;
    c.phase_step = phase_step;
    c.phase_rate = phase_rate;
    c.rem_code = rem_code;
    c.code_step = code_step;
    c.code_rate = code_rate;
    c.n_begin = 0;
    c.n_end = n_samples;
    const int odd = static_cast<int>(sample_offset & 1ULL);
    c.n_first = -odd;
    const float2* __restrict__ base = stream + (sample_offset - static_cast<unsigned long long>(odd));
    int rot[NT];
    rot[0] = 0;
    if (mode_hd_code(MODE))
        {
            // K/..high_dynamics_resampler..:82-85: shift_samples += (int)round((shift[t]-shift[t-1])/step)
            unsigned accum = 0;
#pragma unroll
            for (int t = 1; t < NT; t++)
                {
                    accum += static_cast<unsigned>(static_cast<int>(roundf(__fdiv_rn(__fsub_rn(sh[t], sh[t - 1]), code_step))));
                    rot[t] = static_cast<int>(accum);
                }
        }
    else
        {
#pragma unroll
            for (int t = 1; t < NT; t++) rot[t] = 0;
        }
    float2 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = make_float2(0.0f, 0.0f);
    float2 acc_aux = make_float2(0.0f, 0.0f);
    if (AUX)
        {
            c.aux_on = true;
            c.aux_shift = aux_shift;
            c.aux_zero = (aux_shift == 0.0f);
            c.aux_code_len = code_len;
            c.aux_k_off = static_cast<int>(tab_aux - tab) + MC_MARGIN;
        }
    if (n_samples > 0)
        {
            float smin = sh[0], smax = sh[0];
#pragma unroll
            for (int t = 1; t < NT; t++)
                {
                    smin = fminf(smin, sh[t]);
                    smax = fmaxf(smax, sh[t]);
                }
            if (AUX)
                {
                    smin = fminf(smin, aux_shift);
                    smax = fmaxf(smax, aux_shift);
                }
            const int lo = raw_chip_std(__fmul_rn(code_step, 0.0f), smin, rem_code);
            const int hi = raw_chip_std(__fmul_rn(code_step, static_cast<float>(n_samples - 1)), smax, rem_code);
            const bool fast = !mode_hd_code(MODE) && (code_step >= 0.0f) && (lo >= -MC_MARGIN) && (hi < code_len + MC_MARGIN) && (code_len >= MC_MARGIN);
            const bool zp = (NT & 1) && (sh[NT / 2] == 0.0f) && !mode_hd_code(MODE) && (n_samples < (1 << 24));
            if (fast && zp)
                run_segment<NT, MODE, false, true, AUX>(c, base, tab, sh, rot, acc, &acc_aux);
            else if (fast)
                run_segment<NT, MODE, false, false, AUX>(c, base, tab, sh, rot, acc, &acc_aux);
            else
                run_segment<NT, MODE, true, false, AUX>(c, base, tab, sh, rot, acc, &acc_aux);
        }
#pragma unroll
    for (int t = 0; t < NT; t++)
        {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1)
                {
                    acc[t].x += __shfl_down(acc[t].x, off, 64);
                    acc[t].y += __shfl_down(acc[t].y, off, 64);
                }
        }
    if (AUX)
        {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1)
                {
                    acc_aux.x += __shfl_down(acc_aux.x, off, 64);
                    acc_aux.y += __shfl_down(acc_aux.y, off, 64);
                }
        }
    const int wave = tid >> 6;
    if ((tid & 63) == 0)
        {
#pragma unroll
            for (int t = 0; t < NT; t++) red[wave * GSH_MAX_TAPS + t] = acc[t];
            if (AUX) red[wave * GSH_MAX_TAPS + NT] = acc_aux;
        }
    __syncthreads();
    constexpr int NOUT = AUX ? NT + 1 : NT;
    float2 s = make_float2(0.0f, 0.0f);
    if (tid < NOUT)
        {
#pragma unroll
            for (int w = 0; w < MC_WAVES; w++)
                {
                    s.x += red[w * GSH_MAX_TAPS + tid].x;
                    s.y += red[w * GSH_MAX_TAPS + tid].y;
                }
        }
    __syncthreads();
    if (tid < NOUT) red[tid] = s;
    __syncthreads();
}

// the standard-mode form the batched callers use
template <int NT>
__device__ __forceinline__ void correlate_window_std(const float2* __restrict__ stream, unsigned long long sample_offset, int n_samples,
    const float* __restrict__ tab, int code_len, const float (&sh)[NT], float rem_carr, float phase_step, float rem_code, float code_step,
    float2* __restrict__ red)
{
    correlate_window<NT, 0>(stream, sample_offset, n_samples, tab, code_len, sh, rem_carr, phase_step, 0.0f, rem_code, code_step, 0.0f, red);
}

// standard mode with the fused data-component tap: red[0..NT) the taps, red[NT] the fused one
template <int NT>
__device__ __forceinline__ void correlate_window_std_aux(const float2* __restrict__ stream, unsigned long long sample_offset, int n_samples,
    const float* __restrict__ tab, const float* tab_aux, float aux_shift, int code_len, const float (&sh)[NT], float rem_carr, float phase_step,
    float rem_code, float code_step, float2* __restrict__ red)
{
    correlate_window<NT, 0, true>(stream, sample_offset, n_samples, tab, code_len, sh, rem_carr, phase_step, 0.0f, rem_code, code_step, 0.0f, red, tab_aux, aux_shift);
}
}  // namespace GSH_MC_NS
namespace mcdev = GSH_MC_NS;
}  // namespace gsh
#endif
