// Device-side building blocks of the multicorrelator (shared by csrc/multicorrelator.hip -- the batched open-loop
// kernel -- and csrc/tracking_loop.hip -- the closed DLL/PLL loop).  See multicorrelator.hip for the design notes and
// the reference citations.  Everything here is __device__ __forceinline__ code in namespace gsh::mcdev.
#ifndef GSH_MCORR_DEVICE_H
#define GSH_MCORR_DEVICE_H

#include "gsh_internal.h"
#include <cmath>
#include <type_traits>

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
#ifndef GSH_MC_SCAN_ALL_AT_ONCE
#define GSH_MC_SCAN_ALL_AT_ONCE 1  // wave sums of the 1 024-thread kernel: one v_add_f32_dpp per value and step (round 4: 7.63 -> 7.50 us per period)
#endif
#ifndef GSH_MC_SINGLE_SEED
#define GSH_MC_SINGLE_SEED 1  // one transcendental evaluation per lane and window in the 1 024-thread kernel (run_segment_packed; round 4: 7.50 -> 7.43 us per period)
#endif
#ifndef GSH_MC_RESEED
#define GSH_MC_RESEED 32
#endif
#ifndef GSH_MC_PACKED
#define GSH_MC_PACKED 1
#endif
#ifndef GSH_MC_PREFETCH_BANK
#define GSH_MC_PREFETCH_BANK 1  // trips of loads in flight per lane, batched kernel
#endif
#ifndef GSH_MC_EARLY_LOADS
#define GSH_MC_EARLY_LOADS 0  // 1: run_segment_packed, closed-loop form, issues the first trips' loads ahead of the seed evaluation (measured: 0.8 % slower, profiles/ab/r05/closed_loop_notes.txt)
#endif
#ifndef GSH_MC_PREFETCH_LOOP
#define GSH_MC_PREFETCH_LOOP 1  // the same for the 1024-thread closed-loop kernel (4 until the end of round 2: same 9.15 us per period, and the twelve VGPRs
                                // of the deeper queue were what pushed long-lived constants of the loop arithmetic into scratch)
#endif
#ifndef GSH_MC_CVT_FLR
#define GSH_MC_CVT_FLR 1
#endif
#ifndef GSH_MC_PKRTZ
#define GSH_MC_PKRTZ 1  // the paired trip's eight floor() + convert as four v_cvt_pkrtz_f16_f32 over chains scaled by 2^-24 (packed_trip; 0: eight v_cvt_flr_i32_f32, the form of
                        // rounds 2 - 5).  Measured, round 6 (profiles/ab/r06/session25.txt): 176 - 182 us against 187 - 190 per launch of 12 800 jobs.  The other candidate -- the
                        // floor as the rounding of a v_pk_fma_f32 under round-toward-minus-infinity, the mode switched by two s_setreg around the four instructions -- was
                        // bit-exact too and gained 1 - 2 % (session23.txt): not kept.
#endif
#ifndef GSH_MC_RUNLEN
#define GSH_MC_RUNLEN 1  // the paired trips of a segment as counted runs (run_segment_packed); 0: the trip kind asked before every trip, rounds 2 - 6
#endif
#ifndef GSH_MC_DER_MIXED
#define GSH_MC_DER_MIXED 0  // 1: in a trip with one unsafe chunk the other chunk still pairs its taps (two more loop bodies: measured, the register
                            // allocator then spills and the launch is 45 % slower -- profiles/r02/paired_taps.txt)
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
    // sin / cos of pi * x with x = 2 rev in [-1, 1]: the float function reduces its argument exactly (no multiplication by 2 pi in double, no
    // Cody-Waite steps as in sincosf of a radian argument); the phase error is that of rounding x to float, <= 2e-7 rad
    float s, c;
    sincospif(static_cast<float>(2.0 * rev), &s, &c);
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

// index i - MARGIN of the guard-banded table, i in [0, len + 2 MARGIN): one conditional add / subtract when the code is at least a margin long
// (the general modulo costs ~30 instructions and ran for the whole wave in the first and last staging iteration)
__device__ __forceinline__ int wrap_margin(int k, int len)
{
    if (len >= MC_MARGIN) return k < 0 ? k + len : (k >= len ? k - len : k);
    return wrap_chip(k, len);
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_read0(float v)
{
    // lanes whose source is outside the row / whose row is masked off read 0.0f
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, true));
}

// inclusive prefix sum over the 64 lanes of a wave
__device__ __forceinline__ float wave_scan_incl(float v)
{
    v += dpp_read0<0x111, 0xf>(v);  // row_shr:1
    v += dpp_read0<0x112, 0xf>(v);  // row_shr:2
    v += dpp_read0<0x114, 0xf>(v);  // row_shr:4
    v += dpp_read0<0x118, 0xf>(v);  // row_shr:8
    v += dpp_read0<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v += dpp_read0<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
    return v;
}

// the same over N values at once, one v_add_f32_dpp per value and step (the compiler's rendering of the function above is v_mov_b32_dpp + add: half as many
// again).  The values take turns, so a value's next step is N instructions behind its last -- for N < 3 a wait state is inserted by hand (a DPP read of a
// VGPR needs two wait states after the VALU write; the assembler does not add them inside inline asm).
template <int N>
__device__ __forceinline__ void wave_scan_incl_n(float (&v)[N])
{
#define GSH_DPP_STEP(CTRL)                                                                                            \
    _Pragma("unroll") for (int i = 0; i < N; i++)                                                                       \
    {                                                                                                                   \
        if (N < 3) asm volatile("s_nop 1");                                                                             \
        asm volatile("v_add_f32_dpp %0, %0, %0 " CTRL " bound_ctrl:1" : "+v"(v[i]));                                     \
    }
    GSH_DPP_STEP("row_shr:1 row_mask:0xf bank_mask:0xf")
    GSH_DPP_STEP("row_shr:2 row_mask:0xf bank_mask:0xf")
    GSH_DPP_STEP("row_shr:4 row_mask:0xf bank_mask:0xf")
    GSH_DPP_STEP("row_shr:8 row_mask:0xf bank_mask:0xf")
    GSH_DPP_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
    GSH_DPP_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
#undef GSH_DPP_STEP
}

__device__ __forceinline__ float readlane_f(float v, int l)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
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
    bool packed{true};      // run_body_packed allowed (host switch gsh_bank_set_packed_body / GSH_MC_PACKED_BODY=0 for A/B runs)
    bool runs{false};       // the run-based path may be used (RUNS kernels)
    bool aux_on{false};
    bool aux_zero{false};   // its shift is exactly 0.0f: on the ZP path it shares the prompt tap's chip index
    float aux_shift{0.0f};
    int aux_code_len{1};
    int aux_k_off{MC_MARGIN};
    int aux_k_lo{0}, aux_k_hi{-1};
    // closed-loop kernel, windows of one exact seed per lane (run_segment_packed, `single`): the factors of the lanes' seeds, evaluated by two otherwise idle waves
    // beside thread 0's end of the previous period (seed_table_fill below; tracking_loop.hip) -- null: every lane evaluates its own
    const float2* seed_tab{nullptr};
    // batched kernel (multicorrelator.hip): the factors of every lane's seeds for THIS job, evaluated once per work-group by one wave beside the code staging
    // (fac_table_fill below) -- null: every lane evaluates its own two transcendentals (run_segment_packed)
    const float2* fac_tab{nullptr};
};
// factor table of the batched kernel: the seed of lane tid at re-seed r is A[r] * (WH[tid / 32] * B[tid % 32])
//   = exp(-j (rem + (n_first + 2 PPC NCH RESEED r) step)) * exp(-j 64 (tid / 32) step) * exp(-j 2 (tid % 32) step)
// 64 entries, one per lane of the filling wave: each evaluated ONCE per job and work-group instead of two evaluations per lane of every wave
// (round 6: 77 of a wave's 1 857 vector instructions per 25 000-sample job, profiles/ab/r06/session4.txt).
constexpr int FAC_B = 0;      // 32: exp(-j 2 l step)
constexpr int FAC_WH = 32;    //  8: exp(-j 64 h step), h = tid / 32 (work-groups of up to 256 threads)
constexpr int FAC_INC = 40;   //  3: exp(-j step), exp(-j 2 PPC step), exp(-j 2 NCH PPC step)
constexpr int FAC_A = 44;     // 20: the exact phasor of the chunk's first sample at re-seed r
constexpr int FAC_NA = 20;
constexpr int FAC_ENTRIES = 64;
static_assert(MC_THREADS > 256 || MC_THREADS / 32 <= FAC_INC - FAC_WH, "factor table: one WH entry per 32 threads");
// trips between exact re-seeds of run_segment_packed for NCH chunks per trip
__host__ __device__ constexpr int packed_reseed_trips(int nch) { return (GSH_MC_RESEED + nch - 1) / nch; }
// called by ONE wave (lane = its lane index); the arguments are formed as run_segment_packed forms them (double products of the float step)
template <int NCH>
__device__ __forceinline__ void fac_table_fill(float2* __restrict__ tab, float step, float rem, int n_first, int lane)
{
    const double sd = static_cast<double>(step);
    double ph;
    if (lane < FAC_WH)
        ph = static_cast<double>(2 * lane) * sd;
    else if (lane < FAC_INC)
        ph = static_cast<double>(64 * (lane - FAC_WH)) * sd;
    else if (lane < FAC_A)
        ph = (lane == FAC_INC ? 1.0 : (lane == FAC_INC + 1 ? static_cast<double>(2 * MC_THREADS) : static_cast<double>(2 * NCH * MC_THREADS))) * sd;
    else
        {
            const long long nb = static_cast<long long>(n_first) + static_cast<long long>(2 * MC_THREADS * NCH * packed_reseed_trips(NCH)) * (lane - FAC_A);
            ph = static_cast<double>(rem) + static_cast<double>(nb) * sd;
        }
    tab[lane] = expmj(ph);
}
// seed tables: exp(-j (rem + (n_first + 2 tid) step)) = A[-n_first] * W[tid / 64] * B[tid % 64]
constexpr int SEED_B = 0;     // 64: exp(-j 2 l step)
constexpr int SEED_W = 64;    // 16: exp(-j 128 v step)
constexpr int SEED_A = 80;    //  2: exp(-j (rem - odd step)), odd = 0, 1
constexpr int SEED_INC = 82;  // exp(-j step), exp(-j 2 PPC step), exp(-j 4 PPC step)
constexpr int SEED_ENTRIES = 88;
// one entry per lane of two waves (which = 0: the lane factors; 1: the wave factors, the two window parities, the three wave-uniform rotations).
// The arguments are formed as run_segment_packed forms them (double products of the float step), each evaluated once (expmj: <= 2e-7 rad).
__device__ __forceinline__ void seed_table_fill(float2* __restrict__ tab, float step, double rem, int lane, int which)
{
    asm volatile("" : "+v"(lane));  // (addresses formed here, not hoisted out of the caller's loop)
    const double sd = static_cast<double>(step);
    if (which == 0)
        tab[SEED_B + lane] = expmj(static_cast<double>(2 * lane) * sd);
    else if (lane < 21)
        {
            double ph;
            int at;
            if (lane < 16)
                {
                    ph = static_cast<double>(128 * lane) * sd;
                    at = SEED_W + lane;
                }
            else if (lane < 18)
                {
                    ph = rem + static_cast<double>(-(lane - 16)) * sd;
                    at = SEED_A + (lane - 16);
                }
            else
                {
                    ph = (lane == 18 ? 1.0 : static_cast<double>((lane == 19 ? 2 : 4) * MC_PAIRS_PER_CHUNK)) * sd;
                    at = SEED_INC + (lane - 18);
                }
            tab[at] = expmj(ph);
        }
}
// (the window's carrier phase remainder as the loop publishes it; a SEGMENT of a window that starts n_begin samples in passes rem + n_begin * step, formed in double)
__device__ __forceinline__ void seed_table_fill(float2* __restrict__ tab, float step, float rem, int lane, int which)
{
    seed_table_fill(tab, step, static_cast<double>(rem), lane, which);
}

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

// ---------------------------------------------------------------------------------------------------------------------
// Packed-FP32 body (round 2).  PMC of the round-1 kernel showed the SIMDs issuing VALU instructions for the whole launch
// (SQ_ACTIVE_INST_VALU ~ kernel time): what bounds the correlator is the NUMBER of VALU instructions per sample, 48 per pair of
// samples with three taps.  This body does the same arithmetic on FOUR samples per lane and trip -- the pair (n0, n0 + 1) of chunk A
// and the pair 2 * MC_PAIRS_PER_CHUNK samples further (chunk B), both 16-byte loads fully coalesced as before -- and spells every step
// that exists twice as one packed instruction (v_pk_mul/add/fma_f32, gfx950):
//   * chip-index chains: (step * (float)n + shift) - rem for the two samples of a pair in one v_pk_mul + two v_pk_add.  The packed
//     forms round each lane once, IEEE round-to-nearest, exactly like their scalar forms, and t + (-rem) IS t - rem; chip selection stays
//     bit-exact (tests/test_tracking_gpu.py::test_chip_selection_bit_exact);
//   * ONE carrier phasor per lane and trip instead of two: all four samples are rotated by the phasor of sample n0 (two packed
//     instructions per sample) and accumulated into four accumulator sets; the missing constant rotations -- exp(-j step) for the second
//     sample of a pair, exp(-j 2 PPC step) for chunk B -- are applied ONCE to the finished sums (the sum is linear):
//         acc = A0 + inc * A1 + w * (B0 + inc * B1);
//   * multiply-accumulate: one v_pk_fma per (sample, tap), the code value broadcast by op_sel from the pair the two look-ups fill.
// Per pair of samples and three taps: 30 VALU instructions (18 packed, 6 v_cvt_flr, 6 address) instead of 48.
// Used for the standard mode when the indices need no per-sample wrap and (float)n is exact (windows < 2^24 samples).
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2f pk_cmul(v2f x, v2f p)  // complex x * p
{
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(x), "v"(p));                                          // (xr pr, xr pi)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(x), "v"(p), "v"(t));          // (-xi pi, xi pr) + t
    return r;
}
// the same with a wave-uniform p taken straight from an SGPR pair (no per-trip copy into VGPRs)
__device__ __forceinline__ v2f pk_cmul_s(v2f x, v2f p)
{
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(x), "s"(p));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(x), "s"(p), "v"(t));
    return r;
}
// acc += y * c.lo / c.hi (the real code value broadcast to both components)
__device__ __forceinline__ void pk_fma_lo(v2f& acc, v2f y, v2f c)
{
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(y), "v"(c));
}
__device__ __forceinline__ void pk_fma_hi(v2f& acc, v2f y, v2f c)
{
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(y), "v"(c));
}
// (v.lo * k.lo, v.hi * k.lo): k is wave-uniform (an SGPR pair)
__device__ __forceinline__ v2f pk_mul_slo(v2f v, v2f k)
{
    v2f r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(r) : "v"(v), "s"(k));
    return r;
}
__device__ __forceinline__ v2f pk_add_slo(v2f v, v2f k)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(r) : "v"(v), "s"(k));
    return r;
}
__device__ __forceinline__ v2f pk_add_shi(v2f v, v2f k)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(v), "s"(k));
    return r;
}

// One trip of the packed path: NCH chunks of MC_PAIRS_PER_CHUNK pairs; lane `tid` owns the pair tid of each chunk.  The caller hands over
// the four samples (already zero where they lie outside the segment) and the sample indices the chip look-ups use (nfA / nfB: (float)n of
// the pair; for a sample outside the segment the caller passes the nearest index INSIDE it, so that its look-up stays within what is
// staged -- the chip index is monotone in n -- and no clamp is needed here: masked trips run the very same instructions).
// (the four samples arrive already rotated by the trip's phasor: the caller rotates them first and then re-uses their registers for the loads of
//  the trip PF ahead, so the prefetch queue needs no register copies)
// Paired taps (round 2).  What the instructions of a trip cost on gfx950 (profiles/ubench/valu_rate.hip, profiles/r02/valu_rate_gfx950.txt): v_fma / v_add / v_mul
// / v_and / v_add_u32 issue at full rate, every packed FP32 instruction, v_cvt_flr_i32_f32, v_fract_f32, v_cmp + v_cndmask and the three-operand
// v_lshl_add_u32 at HALF rate.  Of a trip's 120 half-cycles 68 were chip-index work: per tap and sample two adds, one convert, one address.  Two
// things cut that:
//   * constant table offset (KC): when the whole code is staged, tab[k + MARGIN] is v_lshlrev_b32 + an immediate offset of the ds_read, not v_lshl_add_u32;
//   * early from late (PA / PB): if shift_L - shift_E is exactly 1, then wherever no rounding step of the two chains changes its binade
//       fl(a + shift_E) = fl(a + shift_L) - 1  and  fl(fl(a + shift_E) - rem) = fl(fl(a + shift_L) - rem) - 1
//     hold exactly (x and x - 1 are rounded to the same quantum, and 1 is an even multiple of it: round-to-nearest-even commutes with the shift), so
//     k_E = k_L - 1: the early code value is read next to the late one and its chain -- two adds, a convert, an address per sample -- is not evaluated.
//     Whether a wave's 128 samples of a chunk satisfy the binade conditions is a property of their range of chip indices: the caller evaluates it once per
//     chunk before the loop (one lane per chunk, a ballot per wave) with margins; a wave whose samples straddle a power of two -- or reach below 1 -- takes
//     the per-tap chains for that trip as before.
// The products and their order of summation are the same either way: the outputs are bit-identical with the switch off
// (tests/test_tracking_gpu.py::test_paired_taps_are_bit_identical).
template <int NT, bool ZP, bool AUX, int NCH, bool PA = false, bool PB = false, bool KC = false>
__device__ __forceinline__ void packed_trip(const JobCtx& c, const float* __restrict__ tab, const v2f (&shp)[NT],
    v2f k_step_nrem, v2f aux_shp, bool aux_on, v2f nfA, v2f nfB, v2f yA0, v2f yA1, v2f yB0, v2f yB1, v2f (&A0)[NT], v2f (&A1)[NT], v2f (&B0)[NT],
    v2f (&B1)[NT], v2f& XA0, v2f& XA1, v2f& XB0, v2f& XB1, v2f scaled_step_nrem = (v2f){0.0f, 0.0f}, v2f scaled_shP_shL = (v2f){0.0f, 0.0f})
{
    static_assert(!(PA || PB) || NT == 3, "paired taps: early / prompt / late");
    const v2f zero = {0.0f, 0.0f};
    v2f aB = zero;
    if (NCH == 2) aB = pk_mul_slo(nfB, k_step_nrem);
    const v2f aA = pk_mul_slo(nfA, k_step_nrem);
    const int koff = KC ? MC_MARGIN : c.k_off;
    const int aux_koff = c.aux_k_off;
    typedef const __attribute__((address_space(3))) float* lds_float_ptr;
    // tab[k + off + D]: with KC the offset is a constant and the address one full-rate v_lshlrev_b32 (the compiler's own choice for k * 4 + constant is
    // v_lshl_add_u32 with a zero addend -- half rate, see above); the constant rides in the ds_read's immediate offset
    auto code_at = [&](int k, int off, auto dc, auto main_table) -> float {
        constexpr int D = decltype(dc)::value;
#ifdef GSH_EXP_NOLDS  // timing experiment only: the code value without the LDS look-up
        return __builtin_bit_cast(float, 0x3f800000 | ((k + D) << 31));
#endif
        if constexpr (KC && decltype(main_table)::value)
            {
                unsigned byte_addr;
                asm("v_lshlrev_b32 %0, 2, %1" : "=v"(byte_addr) : "v"(k));
                return reinterpret_cast<lds_float_ptr>(byte_addr)[MC_MARGIN + D];  // KC: `tab` IS LDS address 0 (the caller checks)
            }
        else
            return tab[k + off + D];
    };
    using d0 = std::integral_constant<int, 0>;
    using dm1 = std::integral_constant<int, -1>;
    auto chain = [&](v2f a, v2f sp, bool zero_shift, int& k0, int& k1) {
        v2f u;
        if (zero_shift)
            u = pk_add_shi(a, k_step_nrem);                   // a - rem
        else
            u = pk_add_shi(pk_add_slo(a, sp), k_step_nrem);   // (a + shift) - rem
        k0 = floor_to_int(u.x);
        k1 = floor_to_int(u.y);
    };
    // code values of one tap for the two samples of a pair
    auto lookup = [&](v2f a, v2f sp, bool zero_shift, int k_off, auto main_table) -> v2f {
        int k0, k1;
        chain(a, sp, zero_shift, k0, k1);
        v2f cv;
        cv.x = code_at(k0, k_off, d0{}, main_table);
        cv.y = code_at(k1, k_off, d0{}, main_table);
        return cv;
    };
    // one chunk: S0 / S1 the accumulator sets of the pair's first / second sample
    auto chunk = [&](v2f a, v2f y0, v2f y1, v2f (&S0)[NT], v2f (&S1)[NT], auto paired) {
        if constexpr (decltype(paired)::value && NT == 3)
            {
                // prompt as ever; late through its chain, early read next to it: (code[k_L - 1], code[k_L]) per sample, picked by op_sel
                const v2f cP = lookup(a, shp[1], ZP, koff, std::true_type{});
                int k0, k1;
                chain(a, shp[2], false, k0, k1);
                v2f el0, el1;
                el0.x = code_at(k0, koff, dm1{}, std::true_type{});
                el0.y = code_at(k0, koff, d0{}, std::true_type{});
                el1.x = code_at(k1, koff, dm1{}, std::true_type{});
                el1.y = code_at(k1, koff, d0{}, std::true_type{});
                pk_fma_lo(S0[0], y0, el0);
                pk_fma_lo(S1[0], y1, el1);
                pk_fma_lo(S0[1], y0, cP);
                pk_fma_hi(S1[1], y1, cP);
                pk_fma_hi(S0[2], y0, el0);
                pk_fma_hi(S1[2], y1, el1);
            }
        else
            {
#pragma unroll
                for (int t = 0; t < NT; t++)
                    {
                        const v2f cv = lookup(a, shp[t], ZP && t == NT / 2, koff, std::true_type{});
                        pk_fma_lo(S0[t], y0, cv);
                        pk_fma_hi(S1[t], y1, cv);
                    }
            }
    };
    if constexpr (PA && PB && NCH == 2 && NT == 3)
        {
            // both chunks paired (the common trip): all eight look-ups of the trip go out before the first multiply-accumulate waits for one -- left to itself the
            // scheduler, short of registers, serialises chunk A's reads, its accumulates, chunk B's reads, its accumulates: two exposed LDS round trips per trip
            struct Codes
            {
                v2f p, el0, el1;
            };
            auto codes_of = [&](v2f a) -> Codes {
                Codes d;
                d.p = lookup(a, shp[1], ZP, koff, std::true_type{});
                int k0, k1;
                chain(a, shp[2], false, k0, k1);
                d.el0.x = code_at(k0, koff, dm1{}, std::true_type{});
                d.el0.y = code_at(k0, koff, d0{}, std::true_type{});
                d.el1.x = code_at(k1, koff, dm1{}, std::true_type{});
                d.el1.y = code_at(k1, koff, d0{}, std::true_type{});
                return d;
            };
            auto accumulate = [&](const Codes& d, v2f y0, v2f y1, v2f (&S0)[NT], v2f (&S1)[NT]) {
                pk_fma_lo(S0[0], y0, d.el0);
                pk_fma_lo(S1[0], y1, d.el1);
                pk_fma_lo(S0[1], y0, d.p);
                pk_fma_hi(S1[1], y1, d.p);
                pk_fma_hi(S0[2], y0, d.el0);
                pk_fma_hi(S1[2], y1, d.el1);
            };
#if GSH_MC_PKRTZ
            Codes dA, dB;
            if constexpr (KC)
                {
                    // TWO FLOORS PER INSTRUCTION (round 6).  The eight chains of the trip end in floor() + convert (v_cvt_flr_i32_f32, half rate, one value each).  Here the
                    // chains run on constants scaled by 2^-24 -- step, shifts and -rem times a power of two: every product and sum is the unscaled one times 2^-24 bit for
                    // bit, as long as nothing leaves the normal range, which the caller's judgement of the trip vouches for (every chain value of a paired trip lies in
                    // [1, 2040)) -- and v_cvt_pkrtz_f16_f32 converts two of them at once: round toward zero (= floor, the values are positive) to half precision, whose
                    // quantum below 2^-13 is 2^-24, so the bit pattern of the half IS the integer floor(u) for u < 2048.  The pair of 16-bit patterns becomes two byte
                    // addresses with v_lshlrev_b16 (gfx9: the upper half of the destination is zeroed) and v_lshrrev_b32 by 14.
                    const v2f aAs = pk_mul_slo(nfA, scaled_step_nrem);
                    const v2f aBs = pk_mul_slo(nfB, scaled_step_nrem);
                    v2f uPA, uPB;
                    if (ZP)
                        {
                            uPA = pk_add_shi(aAs, scaled_step_nrem);
                            uPB = pk_add_shi(aBs, scaled_step_nrem);
                        }
                    else
                        {
                            uPA = pk_add_shi(pk_add_slo(aAs, scaled_shP_shL), scaled_step_nrem);
                            uPB = pk_add_shi(pk_add_slo(aBs, scaled_shP_shL), scaled_step_nrem);
                        }
                    const v2f uLA = pk_add_shi(pk_add_shi(aAs, scaled_shP_shL), scaled_step_nrem);
                    const v2f uLB = pk_add_shi(pk_add_shi(aBs, scaled_shP_shL), scaled_step_nrem);
                    auto two_floors = [](v2f u, unsigned& byte0, unsigned& byte1) {
                        unsigned pk;
                        asm("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(pk) : "v"(u.x), "v"(u.y));
                        asm("v_lshlrev_b16_e32 %0, 2, %1" : "=v"(byte0) : "v"(pk));
                        asm("v_lshrrev_b32_e32 %0, 14, %1" : "=v"(byte1) : "v"(pk));
                    };
                    auto at = [&](unsigned byte_addr, auto dc) -> float {
                        constexpr int D = decltype(dc)::value;
                        return reinterpret_cast<lds_float_ptr>(byte_addr)[MC_MARGIN + D];
                    };
                    unsigned b0, b1, b2, b3, b4, b5, b6, b7;
                    two_floors(uPA, b0, b1);
                    two_floors(uLA, b2, b3);
                    two_floors(uPB, b4, b5);
                    two_floors(uLB, b6, b7);
                    dA.p.x = at(b0, d0{});
                    dA.p.y = at(b1, d0{});
                    dA.el0.x = at(b2, dm1{});
                    dA.el0.y = at(b2, d0{});
                    dA.el1.x = at(b3, dm1{});
                    dA.el1.y = at(b3, d0{});
                    dB.p.x = at(b4, d0{});
                    dB.p.y = at(b5, d0{});
                    dB.el0.x = at(b6, dm1{});
                    dB.el0.y = at(b6, d0{});
                    dB.el1.x = at(b7, dm1{});
                    dB.el1.y = at(b7, d0{});
                }
            else
                {
                    dA = codes_of(aA);
                    dB = codes_of(aB);
                }
#else
            const Codes dA = codes_of(aA);
            const Codes dB = codes_of(aB);
#endif
            __builtin_amdgcn_sched_barrier(0);
            accumulate(dA, yA0, yA1, A0, A1);
            accumulate(dB, yB0, yB1, B0, B1);
        }
    else if constexpr (!PA && !PB && NCH == 2 && NT <= 5)
        {
            // the same for the per-tap form: 4 NT look-ups, then the accumulates
            v2f cA[NT], cB[NT];
#pragma unroll
            for (int t = 0; t < NT; t++) cA[t] = lookup(aA, shp[t], ZP && t == NT / 2, koff, std::true_type{});
#pragma unroll
            for (int t = 0; t < NT; t++) cB[t] = lookup(aB, shp[t], ZP && t == NT / 2, koff, std::true_type{});
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NT; t++)
                {
                    pk_fma_lo(A0[t], yA0, cA[t]);
                    pk_fma_hi(A1[t], yA1, cA[t]);
                }
#pragma unroll
            for (int t = 0; t < NT; t++)
                {
                    pk_fma_lo(B0[t], yB0, cB[t]);
                    pk_fma_hi(B1[t], yB1, cB[t]);
                }
        }
    else if constexpr (!PA && NCH == 1 && NT <= 5)
        {
            // one chunk per trip (the 1 024-thread closed-loop kernel): the same, 2 NT look-ups and then the accumulates
            v2f cA[NT];
#pragma unroll
            for (int t = 0; t < NT; t++) cA[t] = lookup(aA, shp[t], ZP && t == NT / 2, koff, std::true_type{});
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NT; t++)
                {
                    pk_fma_lo(A0[t], yA0, cA[t]);
                    pk_fma_hi(A1[t], yA1, cA[t]);
                }
        }
    else
        {
            chunk(aA, yA0, yA1, A0, A1, std::integral_constant<bool, PA>{});
            if (NCH == 2) chunk(aB, yB0, yB1, B0, B1, std::integral_constant<bool, PB>{});
        }
    if (AUX && aux_on)
        {
            const bool zs = ZP && c.aux_zero;
            const v2f cA = lookup(aA, aux_shp, zs, aux_koff, std::false_type{});
            pk_fma_lo(XA0, yA0, cA);
            pk_fma_hi(XA1, yA1, cA);
            if (NCH == 2)
                {
                    const v2f cB = lookup(aB, aux_shp, zs, aux_koff, std::false_type{});
                    pk_fma_lo(XB0, yB0, cB);
                    pk_fma_hi(XB1, yB1, cB);
                }
        }
}

// EARLY CODES (round 6 experiment, GSH_MC_EARLY_CODES): the both-chunks-paired trip in two halves, so that the caller can issue the trip's eight look-ups BEFORE it
// waits for the trip's samples and rotates them -- the chip-index chains need no sample, so they cover the tail of the loads' latency, and the rotations cover the
// look-ups' -- instead of wait, rotate, chains, look-ups, wait, accumulate.  Same instructions, same products, same order of summation.
struct PairedCodes
{
    v2f p, el0, el1;
};
template <bool ZP, bool KC>
__device__ __forceinline__ PairedCodes paired_codes_of(const JobCtx& c, const float* __restrict__ tab, v2f shP, v2f shL, v2f k_step_nrem, v2f nf)
{
    typedef const __attribute__((address_space(3))) float* lds_float_ptr;
    const int koff = KC ? MC_MARGIN : c.k_off;
    const v2f a = pk_mul_slo(nf, k_step_nrem);
    auto at = [&](int k, int d) -> float {
        if constexpr (KC)
            {
                unsigned byte_addr;
                asm("v_lshlrev_b32 %0, 2, %1" : "=v"(byte_addr) : "v"(k));
                return reinterpret_cast<lds_float_ptr>(byte_addr)[MC_MARGIN + d];
            }
        else
            return tab[k + koff + d];
    };
    PairedCodes d;
    {
        const v2f u = ZP ? pk_add_shi(a, k_step_nrem) : pk_add_shi(pk_add_slo(a, shP), k_step_nrem);
        d.p.x = at(floor_to_int(u.x), 0);
        d.p.y = at(floor_to_int(u.y), 0);
    }
    {
        const v2f u = pk_add_shi(pk_add_slo(a, shL), k_step_nrem);
        const int k0 = floor_to_int(u.x), k1 = floor_to_int(u.y);
        d.el0.x = at(k0, -1);
        d.el0.y = at(k0, 0);
        d.el1.x = at(k1, -1);
        d.el1.y = at(k1, 0);
    }
    return d;
}
__device__ __forceinline__ void paired_accumulate(const PairedCodes& d, v2f y0, v2f y1, v2f (&S0)[3], v2f (&S1)[3])
{
    pk_fma_lo(S0[0], y0, d.el0);
    pk_fma_lo(S1[0], y1, d.el1);
    pk_fma_lo(S0[1], y0, d.p);
    pk_fma_hi(S1[1], y1, d.p);
    pk_fma_hi(S0[2], y0, d.el0);
    pk_fma_hi(S1[2], y1, d.el1);
}

// The whole segment on the packed path -- head, body and tail are trips of ONE loop.
// Carrier phasors: a lane needs exp(-j (rem + n step)) for its own samples at every exact re-seed.  Evaluating that per lane (a double-
// precision range reduction + sincosf, ~70 VALU instructions) a dozen times per window cost as much as a third of the hot loop (PMC, round 2:
// 1 068 of the 2 556 VALU instructions of a wave were outside the loop).  Here a wave evaluates TWO transcendental phasors per lane and window:
//   L    = exp(-j 2 tid step)                      -- the lane's offset inside a chunk, constant for the whole window;
//   T[l] = one table entry per LANE: lane 0..2 the constant rotations inc = exp(-j step), w = exp(-j 2 PPC step), w2 = exp(-j 2 NCH PPC step),
//          lane 4 + r the exact phasor of re-seed r at the chunk's first sample, exp(-j (rem + step (n_first + 2 PPC NCH RESEED r)));
// a lane's seed is T[4 + r] * L (one complex product), read from the table with v_readlane.  More than 60 re-seeds: the table is refilled.
// NCH = 2 chunks per trip (four samples per lane) for the 256-thread batched kernel; NCH = 1 for the 1024-thread closed-loop kernel
// (128 VGPRs per lane).  One summation order for every tap count, so fused and unfused jobs stay bit-identical.
// MRG (round 4, the 1 024-thread closed-loop kernel): ONE accumulator set.  Every sample is rotated by its own phasor -- pa for the pair's first sample, pa inc for its
// second, pa w / pa w inc for chunk B: three more complex products per trip (six packed instructions) -- and all four products of a tap go into the same
// accumulator; the fold at the end is gone.  24 accumulator registers become 6, which is what lets a 1 024-thread work-group (128 VGPRs) run TWO chunks per
// trip: the per-trip overhead (phasor step, index step, loop control, edge tests, the wait for the look-ups) is paid once per four samples instead of once per two.
template <int NT, bool ZP, bool AUX, int NCH, int PF, bool DER = false, bool KC = false, bool MRG = false>
__device__ __forceinline__ void run_segment_packed(const JobCtx& c, const float2* __restrict__ base, const float* __restrict__ tab,
    const float (&sh)[NT], float2 (&acc)[NT], float2* acc_aux)
{
    static_assert(NCH == 1 || NCH == 2, "one or two chunks per trip");
    constexpr int PPC = MC_PAIRS_PER_CHUNK;
    constexpr int RESEED = packed_reseed_trips(NCH);     // trips between exact re-seeds
    constexpr int TBL = 60;                               // re-seed entries per table fill
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const v2f zero = {0.0f, 0.0f};
    // seed tables (c.seed_tab, the closed-loop kernel): the six reads go out first -- their LDS round trip runs under the scalar set-up below
    float2 tb_b = make_float2(1.0f, 0.0f), tb_wv = tb_b, tb_a0 = tb_b, tb_inc = tb_b, tb_w = tb_b, tb_w2 = tb_b;
    if (c.seed_tab != nullptr)  // uniform
        {
            int tl = tid;
            asm volatile("" : "+v"(tl));
            const float2* __restrict__ S = c.seed_tab;
            tb_b = S[SEED_B + (tl & 63)];
            tb_wv = S[SEED_W + (tl >> 6)];
            tb_a0 = S[SEED_A + (c.n_begin - c.n_first)];  // (0 / 1: whether the pair of the segment's first sample starts one sample before it)
            tb_inc = S[SEED_INC];
            tb_w = S[SEED_INC + 1];
            tb_w2 = S[SEED_INC + (NCH == 2 ? 2 : 1)];
        }
    const int span = c.n_end - c.n_first;
    const int n_pairs = (span + 1) >> 1;
    const int n_full = span >> 1;                       // leading pairs whose second sample is in range
    const int odd = c.n_begin - c.n_first;              // 1: pair 0's first sample is outside the segment
    const int n_trips_all = (n_pairs + NCH * PPC - 1) / (NCH * PPC);
    // One chunk per trip (the 1024-thread kernel): a wave's 128 samples of the last trip may lie wholly beyond the segment's end -- 25 000 samples are 12.2 trips
    // of 2 048, and only the first four of the sixteen waves have anything to do in the thirteenth.  Such a wave stops a trip early (its samples there
    // would all be read as zero); with the waves dealt round-robin to the SIMDs every SIMD then runs 49 wave-trips instead of 52.
    int n_trips = n_trips_all;
    if constexpr (NCH == 1 || MRG)
        {
            // (two chunks per trip: the wave's first samples of trip i are those of its chunk A, at n_first + i 2 NCH PPC + 128 wave)
            const int left = c.n_end - (c.n_first + 128 * (tid >> 6));  // samples from the wave's first slice to the end of the segment
            n_trips = left <= 0 ? 0 : min(n_trips_all, (left + 2 * NCH * PPC - 1) / (2 * NCH * PPC));
            n_trips = __builtin_amdgcn_readfirstlane(n_trips);  // wave-uniform: keep the loop control scalar
        }
    const int first_plain = odd ? 1 : 0;                // trips [first_plain, last_plain) need no masking
    const int last_plain = n_full / (NCH * PPC);
    // Loads run PF trips ahead of the arithmetic (a register queue, the trip loop unrolled by PF).  The batched kernel (many work-groups per
    // compute unit) gets by with PF = 1; the closed-loop kernel has ONE work-group per channel and ran PF = 4 until measurement showed PF = 1 to be as fast.
    // (Measured, round 2, profiles/r02/closed_loop_phases.txt: the depth hardly matters -- of the 11 us of a closed-loop period 2.4 us are thread
    // 0's loop arithmetic, 2.6 us fixed cost of the correlation phase (barriers, reductions) and 6 us the 13 trips of each wave.)
    const float2* const q0 = base + 2 * tid;  // the lane's pair of chunk 0
    constexpr int TRIP = 2 * NCH * PPC;        // samples (float2) a trip advances by
    // the four samples of trip i for this lane: 16-byte loads in the body; at the segment's edges (odd head, partial tail) the samples outside
    // [n_begin, n_end) are read as zero -- never loaded.  Edge trips go through the same queue, so their latency is hidden like the others'.
    const unsigned lane_bytes = 16u * static_cast<unsigned>(tid);  // the lane's 16 bytes inside a chunk
    auto load_plain = [&](int i, float4& va, float4& vb) {
            {
                // a wave-uniform 64-bit base plus a 32-bit lane offset: the loads take their base from SGPRs, and no per-lane pointer is carried (and advanced) in VGPRs
                const char* const ua = reinterpret_cast<const char*>(base + static_cast<long long>(i) * TRIP);
#ifdef GSH_EXP_NOLOAD  // timing experiment only (profiles/r02/mcorr_bound_experiments.txt): no sample traffic
                va = make_float4(1.0f, static_cast<float>(i), 0.5f, 0.25f);
                if (NCH == 2) vb = make_float4(0.5f, static_cast<float>(i), 1.0f, 0.25f);
                asm volatile("" : "+v"(va.x), "+v"(va.y), "+v"(va.z), "+v"(va.w));
                if (NCH == 2) asm volatile("" : "+v"(vb.x), "+v"(vb.y), "+v"(vb.z), "+v"(vb.w));
#else
                if constexpr (NCH == 2)
                    {
                        // chunk A / B at -/+ half a chunk around the middle: both inside the 13-bit immediate offset of global_load (a whole chunk, 16 PPC = 4096
                        // bytes at 256 threads, is one more than the field holds and cost a 64-bit add per trip)
                        const char* const mid = ua + 8 * PPC + lane_bytes;
                        va = *reinterpret_cast<const float4*>(mid - 8 * PPC);
                        vb = *reinterpret_cast<const float4*>(mid + 8 * PPC);
                    }
                else
                    va = *reinterpret_cast<const float4*>(ua + lane_bytes);
#endif
            }
    };
    auto load_edge = [&](int i, float4& va, float4& vb) {
        const float2* q = q0 + static_cast<long long>(i) * TRIP;
            {
                const int n0 = c.n_first + 2 * tid + i * TRIP;
                const int lo = c.n_begin, hi = c.n_end - 1;
                const bool a0 = (n0 >= lo) && (n0 <= hi), a1 = (n0 + 1 >= lo) && (n0 + 1 <= hi);
                const float2 x0 = a0 ? q[0] : make_float2(0.0f, 0.0f);
                const float2 x1 = a1 ? q[1] : make_float2(0.0f, 0.0f);
                va = make_float4(x0.x, x0.y, x1.x, x1.y);
                if (NCH == 2)
                    {
                        const int m0 = n0 + 2 * PPC;
                        const bool b0 = (m0 >= lo) && (m0 <= hi), b1 = (m0 + 1 >= lo) && (m0 + 1 <= hi);
                        const float2 z0 = b0 ? q[2 * PPC] : make_float2(0.0f, 0.0f);
                        const float2 z1 = b1 ? q[2 * PPC + 1] : make_float2(0.0f, 0.0f);
                        vb = make_float4(z0.x, z0.y, z1.x, z1.y);
                    }
            }
    };
    auto load_trip = [&](int i, float4& va, float4& vb) {
        if ((i >= first_plain) && (i < last_plain))  // uniform
            load_plain(i, va, vb);
        else
            load_edge(i, va, vb);
    };
    // The first PF trips' loads: right in front of the first trip.  (EARLY_LOADS, an A/B switch for the 1 024-thread closed-loop form: issued HERE instead, ahead of the
    // seeds' transcendental evaluation and the tables below, so that the period's first L2 round trip runs under the set-up -- measured 7.60 against 7.54 us per period:
    // the other waves of the unit already cover that latency, and the eight registers held through the set-up cost more.)
    constexpr bool EARLY_LOADS = GSH_MC_EARLY_LOADS && MRG && MC_THREADS > 256;
    float4 qa[PF], qb[PF];
    auto first_loads = [&]() {
#pragma unroll
        for (int j = 0; j < PF; j++)
            {
                qa[j] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                qb[j] = qa[j];
                if (j < n_trips) load_trip(j, qa[j], qb[j]);  // uniform
            }
    };
    if constexpr (EARLY_LOADS) first_loads();
    v2f A0[NT], A1[NT], B0[NT], B1[NT];
    v2f XA0 = zero, XA1 = zero, XB0 = zero, XB1 = zero;  // the fused tap (AUX)
#pragma unroll
    for (int t = 0; t < NT; t++) A0[t] = A1[t] = B0[t] = B1[t] = zero;
    const v2f k_step_nrem = {c.code_step, -c.rem_code};
#if GSH_MC_PKRTZ
    // the chain constants times 2^-24 for the paired trips' two-floors-per-instruction form (packed_trip); exact unless one of them is so small that its scaled value
    // would be denormal -- then no trip of this segment takes the paired form (scaled_ok, below)
    constexpr float PK_SCALE = 0x1p-24f;
    auto scales_exactly = [](float x) -> bool { return x == 0.0f || fabsf(x) >= 0x1p-100f; };
    auto uniform = [](float x) -> float { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x))); };  // (a product is a VGPR value)
    const v2f scaled_step_nrem = {uniform(c.code_step * PK_SCALE), uniform(-c.rem_code * PK_SCALE)};
    const v2f scaled_shP_shL = {uniform(sh[NT / 2] * PK_SCALE), uniform(sh[NT - 1] * PK_SCALE)};
    const bool scaled_ok = scales_exactly(c.code_step) && scales_exactly(c.rem_code) && scales_exactly(sh[NT / 2]) && scales_exactly(sh[NT - 1]);  // uniform
#else
    const v2f scaled_step_nrem = {0.0f, 0.0f}, scaled_shP_shL = {0.0f, 0.0f};
#endif
    v2f shp[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) shp[t] = (v2f){sh[t], sh[t]};
    const v2f aux_shp = {c.aux_shift, c.aux_shift};
    const v2f stride = {static_cast<float>(2 * NCH * PPC), static_cast<float>(2 * NCH * PPC)};
    const bool aux_on = AUX && c.aux_on;
    const double sd = static_cast<double>(c.phase_step);

    // MRG, windows of at most RESEED trips (one exact seed per lane and window -- every tracking window below 65 536 samples): ONE transcendental evaluation per lane instead
    // of two.  The lane evaluates its own seed exp(-j (rem + (n_first + 2 tid) step)) directly; lanes 0..2 of every wave evaluate the three wave-uniform rotations
    // instead and get their seeds from lane 3's, two samples back per lane (products with conj(inc)^2: one to three more roundings on three lanes of sixty-four).
    const bool single = GSH_MC_SINGLE_SEED && MRG && (n_trips_all <= RESEED);  // uniform
    v2f seed_direct = zero, single_inc = zero, single_w = zero, single_w2 = zero;
    if (single && c.seed_tab != nullptr)  // uniform
        {
            // the seed from its three factors (two complex products), the rotations from the table: a dozen instructions instead of a transcendental evaluation
            // per lane -- sixteen waves' worth of them were a seventh of the period's vector instructions
            // (the table's addresses are formed HERE, from a thread index the compiler cannot see through: hoisted out of the period loop as loop invariants they
            // would sit in registers the loop's constants need -- scratch)
            const float2 sdir = cmul(cmul(tb_a0, tb_wv), tb_b);
            seed_direct = (v2f){sdir.x, sdir.y};
            auto uniform = [&](float2 u) -> v2f {
                return (v2f){__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, u.x))),
                    __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, u.y)))};
            };
            single_inc = uniform(tb_inc);
            single_w = uniform(tb_w);
            single_w2 = uniform(tb_w2);
        }
    else if (single)
        {
            double ph;
            if (lane == 0)
                ph = sd;
            else if (lane == 1)
                ph = static_cast<double>(2 * PPC) * sd;
            else if (lane == 2)
                ph = static_cast<double>(2 * NCH * PPC) * sd;
            else
                ph = static_cast<double>(c.rem_carr) + static_cast<double>(c.n_first + 2 * tid) * sd;
            const float2 E = expmj(ph);
            auto lane_of = [&](int l) -> float2 {
                return make_float2(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, E.x), l)),
                    __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, E.y), l)));
            };
            const float2 i1 = lane_of(0), w1 = lane_of(1), w21 = lane_of(2), s3 = lane_of(3);
            const float2 ci = make_float2(i1.x, -i1.y);
            const float2 q = cmul(ci, ci);
            const float2 s2 = cmul(s3, q), s1 = cmul(s2, q), s0 = cmul(s1, q);
            const float2 mine = lane == 0 ? s0 : (lane == 1 ? s1 : (lane == 2 ? s2 : E));
            seed_direct = (v2f){mine.x, mine.y};
            single_inc = (v2f){i1.x, i1.y};
            single_w = (v2f){w1.x, w1.y};
            single_w2 = (v2f){w21.x, w21.y};
        }
    // batched kernel: the work-group's factor table (JobCtx::fac_tab) when it covers every re-seed of this segment
    const bool fac = !MRG && NCH == 2 && (c.fac_tab != nullptr) && (n_trips_all <= FAC_NA * RESEED);  // uniform
    float2 Lf = make_float2(1.0f, 0.0f);
    if (fac)
        {
            int tl = tid;
            asm volatile("" : "+v"(tl));
            Lf = cmul(c.fac_tab[FAC_WH + (tl >> 5)], c.fac_tab[FAC_B + (tl & 31)]);
        }
    else if (!single)
        Lf = expmj(static_cast<double>(2 * tid) * sd);  // (from the seed tables, W[tid / 64] B[tid % 64], config 4's period is 0.6 % LONGER: profiles/ab/r05/closed_loop_notes.txt)
    const v2f L = {Lf.x, Lf.y};
    auto fill_table = [&](int r0) -> float2 {  // entry of this lane for the re-seeds r0 .. r0 + TBL - 1
        double ph;
        if (lane == 0)
            ph = sd;
        else if (lane == 1)
            ph = static_cast<double>(2 * PPC) * sd;
        else if (lane == 2)
            ph = static_cast<double>(2 * NCH * PPC) * sd;
        else
            {
                const int r = r0 + max(lane - 4, 0);
                const long long nb = static_cast<long long>(c.n_first) + static_cast<long long>(2 * PPC * NCH * RESEED) * r;
                ph = static_cast<double>(c.rem_carr) + static_cast<double>(nb) * sd;
            }
        return expmj(ph);
    };
    float2 T = make_float2(1.0f, 0.0f);
    if (!single && !fac) T = fill_table(0);
    auto table = [&](int l) -> v2f {
        v2f r;
        r.x = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, T.x), l));
        r.y = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, T.y), l));
        return r;
    };
    auto fac_uniform = [&](int at) -> v2f {  // a wave-uniform table entry, in SGPRs
        const float2 u = c.fac_tab[at];
        return (v2f){__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, u.x))),
            __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, u.y)))};
    };
    v2f w2, inc_s, w_s;  // exp(-j 2 NCH PPC step), exp(-j step), exp(-j 2 PPC step): wave-uniform (MRG: used in every trip; otherwise in the fold)
    if (fac)
        {
            inc_s = fac_uniform(FAC_INC);
            w_s = fac_uniform(FAC_INC + 1);
            w2 = fac_uniform(FAC_INC + 2);
        }
    else
        {
            w2 = single ? single_w2 : table(2);
            inc_s = single ? single_inc : table(0);
            w_s = single ? single_w : table(1);
        }

    // paired taps (DER): one bit per chunk of 2 PPC samples (per WAVE: see judge), set where every value of the early and the late index chain stays inside one binade
    // (margins of 1/8 chip; see packed_trip).  Lane l judges chunk 64 m + l.  All of it happens HERE, before the accumulators exist: evaluated inside the
    // trip loop its temporaries cost the loop a dozen VGPRs and with them one wave per SIMD.  Two masks = 128 chunks; what lies
    // beyond (windows longer than 2^16 samples at 256 threads) runs the per-tap chains.
    constexpr int NM = MC_THREADS >= 128 ? 2 : 4;  // masks of 64 chunks: at least 32 768 samples of a window whatever the work-group size
    unsigned long long der_mask[NM];               // chunks 0..63, 64..127, ...
#pragma unroll
    for (int m = 0; m < NM; m++) der_mask[m] = 0ULL;
    if constexpr (DER)
        {
            // float arithmetic is enough: the margins (1/8 chip) exceed its error (< 2^-7 below the 2^16 bound) by far, and a chunk judged unsafe only loses speed
            // A WAVE judges its own 128 samples of each chunk (its lanes' pairs), not the chunk's 2 PPC: the trip form is a per-wave choice, and of the waves of a
            // chunk that straddles a power of two only the one or two that hold the crossing keep the per-tap chains.
            auto judge = [&](int first_chunk) -> unsigned long long {
                const float n_lo = static_cast<float>(c.n_first + 2 * PPC * (first_chunk + lane) + 128 * (tid >> 6));
                const float lo1 = c.code_step * n_lo + (sh[0] - 0.125f);
                const float hi1 = c.code_step * (n_lo + 127.0f) + (sh[NT - 1] + 0.125f);
                const float lo2 = lo1 - c.rem_code, hi2 = hi1 - c.rem_code;
                // (only the form that converts two chains per instruction needs the tighter bound and the scaled constants: whole-code tables, two chunks per trip --
                //  packed_trip.  The windowed tables of long codes pair their taps up to 2^16 as before.)
                constexpr bool TWO_FLOORS = GSH_MC_PKRTZ && KC && NCH == 2 && NT == 3;
                constexpr float HI = TWO_FLOORS ? 2040.0f : 65536.0f;  // the half-precision pattern of u * 2^-24 is floor(u) below 2048
                const bool one_binade_a = (lo1 >= 1.0f) && (hi1 < HI) && ((__float_as_uint(lo1) >> 23) == (__float_as_uint(hi1) >> 23));
                const bool one_binade_u = (lo2 >= 1.0f) && (hi2 < HI) && ((__float_as_uint(lo2) >> 23) == (__float_as_uint(hi2) >> 23));
#if GSH_MC_PKRTZ
                return __ballot(one_binade_a && one_binade_u && (scaled_ok || !TWO_FLOORS));
#else
                return __ballot(one_binade_a && one_binade_u);
#endif
            };
            der_mask[0] = judge(0);
#pragma unroll
            for (int m = 1; m < NM; m++)
                if (NCH * n_trips > 64 * m) der_mask[m] = judge(64 * m);  // uniform
        }
    if constexpr (!EARLY_LOADS) first_loads();
    v2f pa = zero, nfA = zero, nfB = zero;
    int until_reseed = 0, r_idx = 0, tbl0 = 0;
    // one trip; j: its slot in the load queue; FA / FB (compile time): chunk A / B reads its early tap next to the late one
    auto trip = [&](int i, auto jc, auto fa, auto fb, auto kp) {
        constexpr int j = decltype(jc)::value;
        constexpr bool FA = decltype(fa)::value, FB = decltype(fb)::value;
        constexpr bool KP = decltype(kp)::value;  // the caller knows that trip i is a plain one (first_plain <= i < last_plain)
        const int n0 = (c.n_first + i * TRIP) + 2 * tid;  // (uniform part first: used at re-seeds and edges only)
        if (until_reseed == 0)  // uniform: exact re-seed of the lane's phasor and of (float)n
            {
                if (!single && !fac && r_idx - tbl0 >= TBL)
                    {
                        tbl0 = r_idx;
                        T = fill_table(tbl0);
                    }
                if (single)
                    pa = seed_direct;
                else if (fac)
                    {
                        const float2 ar = c.fac_tab[FAC_A + r_idx];  // (uniform address: one LDS broadcast read)
                        pa = pk_cmul((v2f){ar.x, ar.y}, L);
                    }
                else
                    pa = pk_cmul(table(4 + r_idx - tbl0), L);
                nfA = (v2f){static_cast<float>(n0), static_cast<float>(n0 + 1)};
                nfB = (v2f){static_cast<float>(n0 + 2 * PPC), static_cast<float>(n0 + 2 * PPC + 1)};
                until_reseed = RESEED;
                r_idx++;
            }
        until_reseed--;
        const bool plain = KP || ((i >= first_plain) && (i < last_plain));  // uniform
        v2f ia = nfA, ib = nfB;
        if (!(FA || FB) && !plain)  // (derived trips are plain ones)
            {
                // edge trip: a sample outside the segment (already zero) is looked up at the nearest sample inside it, so that its
                // chip index stays within what is staged.  (The empty volatile asm keeps this a BRANCH: if-converted, its 16 clamp /
                // convert / select instructions ran in every trip -- a fifth of the loop's VALU work, ISA of round 2.)
                asm volatile("" ::: "memory");
                const int lo = c.n_begin, hi = c.n_end - 1;
                ia = (v2f){static_cast<float>(min(max(n0, lo), hi)), static_cast<float>(min(max(n0 + 1, lo), hi))};
                const int m0 = n0 + 2 * PPC;
                if (NCH == 2) ib = (v2f){static_cast<float>(min(max(m0, lo), hi)), static_cast<float>(min(max(m0 + 1, lo), hi))};
            }
#ifndef GSH_MC_EARLY_CODES
#define GSH_MC_EARLY_CODES 0
#endif
        constexpr bool EARLY_CODES = GSH_MC_EARLY_CODES && FA && FB && NCH == 2 && NT == 3 && !AUX && !MRG;
        PairedCodes dA, dB;
        if constexpr (EARLY_CODES)
            {
                dA = paired_codes_of<ZP, KC>(c, tab, shp[1], shp[2], k_step_nrem, ia);
                dB = paired_codes_of<ZP, KC>(c, tab, shp[1], shp[2], k_step_nrem, ib);
                __builtin_amdgcn_sched_barrier(0);
            }
        // rotate first: the samples' registers are then free for the loads of trip i + PF, which take this trip's place in the queue
        v2f yA0, yA1, yB0 = zero, yB1 = zero;
        if constexpr (MRG)
            {
                yA0 = pk_cmul((v2f){qa[j].x, qa[j].y}, pa);
                yA1 = pk_cmul((v2f){qa[j].z, qa[j].w}, pk_cmul_s(pa, inc_s));
                if (NCH == 2)
                    {
                        const v2f pb = pk_cmul_s(pa, w_s);
                        yB0 = pk_cmul((v2f){qb[j].x, qb[j].y}, pb);
                        yB1 = pk_cmul((v2f){qb[j].z, qb[j].w}, pk_cmul_s(pb, inc_s));
                    }
            }
        else
            {
                yA0 = pk_cmul((v2f){qa[j].x, qa[j].y}, pa);
                yA1 = pk_cmul((v2f){qa[j].z, qa[j].w}, pa);
                if (NCH == 2)
                    {
                        yB0 = pk_cmul((v2f){qb[j].x, qb[j].y}, pa);
                        yB1 = pk_cmul((v2f){qb[j].z, qb[j].w}, pa);
                    }
            }
#if GSH_MC_RUNLEN
        if constexpr ((FA && (FB || NCH == 1)) || KP)
            {
                // (a paired trip is a plain one: the trip PF ahead lies at or beyond the first plain trip, and one comparison says whether it is plain itself)
                if (i + PF < last_plain)  // uniform
                    load_plain(i + PF, qa[j], qb[j]);
                else if (i + PF < n_trips)
                    load_edge(i + PF, qa[j], qb[j]);
            }
        else
#endif
        if (i + PF < n_trips) load_trip(i + PF, qa[j], qb[j]);  // uniform
        if constexpr (EARLY_CODES)
            {
                if constexpr (NT == 3)
                    {
                        __builtin_amdgcn_sched_barrier(0);
                        paired_accumulate(dA, yA0, yA1, A0, A1);
                        paired_accumulate(dB, yB0, yB1, B0, B1);
                    }
            }
        else if constexpr (MRG)  // (one set: the four references name the same registers, the accumulates follow one another)
            packed_trip<NT, ZP, AUX, NCH, FA, FB, KC>(c, tab, shp, k_step_nrem, aux_shp, aux_on, ia, ib, yA0, yA1, yB0, yB1, A0, A0, A0, A0, XA0, XA0, XA0, XA0);
        else
            packed_trip<NT, ZP, AUX, NCH, FA, FB, KC>(c, tab, shp, k_step_nrem, aux_shp, aux_on, ia, ib, yA0, yA1, yB0, yB1, A0, A1, B0, B1, XA0, XA1, XB0, XB1, scaled_step_nrem,
                scaled_shP_shL);
        pa = pk_cmul_s(pa, w2);  // (w2 and the stride are wave-uniform: straight from SGPRs, not copied into VGPRs every trip)
        asm("v_pk_add_f32 %0, %0, %1" : "+v"(nfA) : "s"(stride));
        if (NCH == 2) asm("v_pk_add_f32 %0, %0, %1" : "+v"(nfB) : "s"(stride));
    };
    using std::integral_constant;
    using no = integral_constant<bool, false>;
    using yes = integral_constant<bool, true>;
    if constexpr (DER)
        {
            // Runs of trips of one kind, each kind its own loop over straight-line code.  (A branch per trip between the two forms made the
            // register allocator merge the twelve accumulator pairs with 23 copies at every join -- ISA, round 2.)
            static_assert(PF == 1, "the paired-tap trips run with a one-deep load queue");
            using j0 = integral_constant<int, 0>;
            auto flags = [&](int i) -> unsigned {  // bit 0 / 1: chunk A / B of trip i may pair its taps
#ifdef GSH_EXP_ALLSAFE  // timing experiment only: every plain trip pairs its taps (wrong chips near powers of two)
                return ((i >= first_plain) && (i < last_plain)) ? (NCH == 2 ? 3U : 1U) : 0U;
#endif
                const int ch = NCH * i;
                unsigned long long mask = der_mask[0];  // uniform selects
#pragma unroll
                for (int m = 1; m < NM; m++) mask = (ch >= 64 * m) ? der_mask[m] : mask;
                const bool plain = (i >= first_plain) && (i < last_plain) && (ch < 64 * NM);
                const unsigned bits = static_cast<unsigned>(mask >> (ch & 63)) & (NCH == 2 ? 3U : 1U);
                return plain ? bits : 0U;
            };
            constexpr unsigned ALL = (NCH == 2) ? 3U : 1U;
            int i = 0;
#if GSH_MC_RUNLEN && !GSH_MC_DER_MIXED && !defined(GSH_EXP_ALLSAFE)
            // RUN LENGTHS (round 6).  The form above asks flags(i) before EVERY trip: a dozen and a half dependent scalar instructions (mask select, shift, range tests,
            // selects) between the last vector instruction of one trip and the first of the next, in which this wave issues nothing.  Asked once per RUN instead: how many
            // trips from i on may pair their taps -- the trailing ones of the judgement mask from chunk NCH i on (both chunks of a trip: m & m >> 1 on the even bits),
            // capped by the mask word's end and by the last plain trip -- and the run is a counted loop.
            (void)flags;
            auto paired_run = [&](int i0) -> int {
                const int ch = NCH * i0;
                if (i0 < first_plain || i0 >= last_plain || ch >= 64 * NM) return 0;  // uniform
                unsigned long long mask = der_mask[0];
#pragma unroll
                for (int m = 1; m < NM; m++) mask = (ch >= 64 * m) ? der_mask[m] : mask;
                unsigned long long m = mask >> (ch & 63);  // bit 0: chunk ch; zeros come in from the top, so a run ends at the word's end at the latest
                constexpr unsigned long long UNITS = (NCH == 2) ? 0x5555555555555555ULL : ~0ULL;
                if (NCH == 2) m &= (m >> 1);
                const unsigned long long stop = ~m & UNITS;  // lowest set bit: the first trip that may not pair
                const int run = (stop != 0ULL) ? static_cast<int>(__builtin_ctzll(stop)) / NCH : 64 / NCH;
                return min(run, last_plain - i0);
            };
            // (the shape of the loops is the one below -- a run of paired trips, then ONE trip of the other kind -- because that is the shape whose joins the register
            //  allocator gets through without copies: an if / else of the two kinds put two dozen v_mov_b64 and a scratch slot at every end of a run.  A run that
            //  ends at a mask word's end is followed by one per-tap trip it did not need: one in 32.)
            while (i < n_trips)
                {
                    int run = paired_run(i);
                    while (run > 0)
                        {
                            trip(i, j0{}, yes{}, integral_constant<bool, NCH == 2>{}, no{});
                            i++;
                            run--;
                        }
                    if (i >= n_trips) break;
                    trip(i, j0{}, no{}, no{}, no{});
                    i++;
                }
            (void)ALL;
#else
            while (i < n_trips)
                {
                    unsigned f = flags(i);
                    while (f == ALL)
                        {
                            trip(i, j0{}, yes{}, integral_constant<bool, NCH == 2>{}, no{});
                            i++;
                            if (i >= n_trips) break;
                            f = flags(i);
                        }
                    if (i >= n_trips) break;
#if GSH_MC_DER_MIXED
                    if (NCH == 2 && f == 1U)
                        trip(i, j0{}, yes{}, no{}, no{});
                    else if (NCH == 2 && f == 2U)
                        trip(i, j0{}, no{}, yes{}, no{});
                    else
#endif
                        trip(i, j0{}, no{}, no{}, no{});
                    i++;
                }
#endif
        }
#if GSH_MC_RUNLEN
    else if constexpr (PF == 1)
        {
            // the trips of a segment in three counted ranges -- at most one masked trip at the head, the plain trips, the masked tail -- so that a plain trip neither asks
            // what it is nor what the trip after it is with more than one comparison (see the paired runs above)
            using j0 = integral_constant<int, 0>;
            int i = 0;
            if (first_plain > 0 && n_trips > 0)
                {
                    trip(0, j0{}, no{}, no{}, no{});
                    i = 1;
                }
            for (; i < last_plain; i++) trip(i, j0{}, no{}, no{}, yes{});
            for (; i < n_trips; i++) trip(i, j0{}, no{}, no{}, no{});
        }
#endif
    else
        {
            for (int i0 = 0; i0 < n_trips; i0 += PF)
                {
                    auto slot = [&](auto jc) -> bool {
                        const int i = i0 + decltype(jc)::value;
                        if (i >= n_trips) return false;  // uniform
                        trip(i, jc, no{}, no{}, no{});
                        return true;
                    };
                    bool go = slot(integral_constant<int, 0>{});
                    if constexpr (PF > 1) go = go && slot(integral_constant<int, 1>{});
                    if constexpr (PF > 2) go = go && slot(integral_constant<int, 2>{});
                    if constexpr (PF > 3) go = go && slot(integral_constant<int, 3>{});
                    static_assert(PF <= 4, "load queue: at most four trips");
                    (void)go;
                }
        }
    if constexpr (MRG)
        {
#pragma unroll
            for (int t = 0; t < NT; t++)
                {
                    acc[t].x += A0[t].x;
                    acc[t].y += A0[t].y;
                }
            if (aux_on)
                {
                    acc_aux->x += XA0.x;
                    acc_aux->y += XA0.y;
                }
            return;
        }
    // fold: acc += A0 + inc * A1 + w * (B0 + inc * B1)
    const float2 inc = make_float2(inc_s.x, inc_s.y), w = make_float2(w_s.x, w_s.y);
#pragma unroll
    for (int t = 0; t < NT; t++)
        {
            const float2 a1 = cmul(make_float2(A1[t].x, A1[t].y), inc);
            const float2 b1 = cmul(make_float2(B1[t].x, B1[t].y), inc);
            const float2 b = cmul(make_float2(B0[t].x + b1.x, B0[t].y + b1.y), w);
            acc[t].x += (A0[t].x + a1.x) + b.x;
            acc[t].y += (A0[t].y + a1.y) + b.y;
        }
    if (aux_on)
        {
            const float2 a1 = cmul(make_float2(XA1.x, XA1.y), inc);
            const float2 b1 = cmul(make_float2(XB1.x, XB1.y), inc);
            const float2 b = cmul(make_float2(XB0.x + b1.x, XB0.y + b1.y), w);
            acc_aux->x += (XA0.x + a1.x) + b.x;
            acc_aux->y += (XA0.y + a1.y) + b.y;
        }
}

// ---------------------------------------------------------------------------------------------------------------------
// Run-based path (round 2, experiment -> see DESIGN section 3).  The packed trips above spend two thirds of their VALU issue on evaluating
// the float32 chip-index chain once per tap and SAMPLE, although at 25 Msps the chip index of a tap changes only every 24th sample.  Here the
// chain is evaluated only where it changes:
//   * a wave takes a "super-trip" of S = 64 R consecutive samples (R = 8: 512); coalesced 16-byte loads put them into the wave's LDS buffer
//     in natural order (phase 1); each lane then owns the RUN of R consecutive samples i R .. i R + R - 1, forms the prefix sums
//     q_p = sum_{p' <= p} x_p' w^p' of its run (w^p = exp(-j p step), wave-uniform) and writes them back in place; the run totals, rotated
//     by the lane's phasor ph_i, are prefix-scanned across the wave (DPP) -> E_i, total T (phase 2).  P[n] = E_i + ph_i q_{p-1} is then the
//     carrier-wiped prefix sum of the super-trip up to (not including) sample n = a + i R + p;
//   * the sum a tap wants, sum_n code[idx_t(n)] y[n], over the super-trip is  code[k_first] T + sum_k (code[k] - code[k-1]) (T - P[n*_k])
//     over the chip boundaries n*_k = min{n : idx_t(n) >= k} inside it.  A boundary is found EXACTLY: an estimate from the real-valued
//     formula, then the reference's own float32 chain floor((step (float)n + shift) - rem) evaluated at n* - 1 and n* until
//     idx(n* - 1) < k <= idx(n*) holds (the chain is monotone in n: every rounding is).  One lane per (tap, boundary): 64 / NT lanes per tap,
//     more boundaries than that are taken in further passes (phase 3).
// The chip selection is therefore the reference's, sample for sample; the accumulators differ from the per-sample order of summation only
// by float rounding of the prefix differences (~ ulp of the sum over 512 samples per boundary).
// Needs: standard mode, no per-sample wrap (indices inside what is staged), code_step > 0, windows below 2^24 samples; per wave
// RUNS_LDS_FLOATS floats of LDS scratch.
template <int R>
struct RunsLayout
{
    static constexpr int S = 64 * R;          // samples per super-trip
    static constexpr int STRIDE = R + 2;      // float2 per run in LDS: 16-byte aligned rows, conflict-free ds_read_b128 across 8 lanes (R = 8: 20 words)
    static constexpr int Y_F2 = 64 * STRIDE;  // float2
    static constexpr int FLOATS = 2 * (Y_F2 + 64 + 64);  // y / q buffer, (E_i, ph_i) per run
};

template <int NT, int R>
__device__ __forceinline__ void run_segment_runs(const JobCtx& c, const float2* __restrict__ base, const float* __restrict__ tab, const float (&sh)[NT],
    float2 (&acc)[NT], float* __restrict__ wave_lds)
{
    using LY = RunsLayout<R>;
    constexpr int S = LY::S, STRIDE = LY::STRIDE;
    constexpr int LT = 64 / NT;  // lanes per tap in the boundary phase
    static_assert(R >= 4 && R <= 16 && (R & (R - 1)) == 0, "run length: 4, 8 or 16");
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    float2* const ybuf = reinterpret_cast<float2*>(wave_lds);
    float4* const epbuf = reinterpret_cast<float4*>(ybuf + LY::Y_F2);  // per run: (E_i, ph_i)
    const int a0 = c.n_first;
    const int n_st = (c.n_end - a0 + S - 1) / S;  // super-trips of the segment
    const double sd = static_cast<double>(c.phase_step);

    // ---- per-window constants of this lane: two transcendental phasors per lane and window
    const float2 Lf = expmj(static_cast<double>(R * lane) * sd);  // the lane's run offset inside a super-trip
    const v2f Li = {Lf.x, Lf.y};
    // table T: lanes 0..47 hold the exact phasor of the first sample of one of this wave's next 48 super-trips, lane 48 + p holds w^p = exp(-j p step)
    constexpr int SEEDS = 48;
    auto fill_table = [&](int it0) -> float2 {
        double phs;
        if (lane >= SEEDS)
            phs = static_cast<double>(min(lane - SEEDS, R - 1)) * sd;
        else
            {
                const long long nb = static_cast<long long>(a0) + static_cast<long long>(S) * (static_cast<long long>(wave) + static_cast<long long>(MC_WAVES) * (it0 + lane));
                phs = static_cast<double>(c.rem_carr) + static_cast<double>(nb) * sd;
            }
        return expmj(phs);
    };
    float2 Tb = fill_table(0);
    v2f wp[R];
#pragma unroll
    for (int p = 0; p < R; p++) wp[p] = (v2f){readlane_f(Tb.x, SEEDS + p), readlane_f(Tb.y, SEEDS + p)};
    const int my_t = lane / LT;           // >= NT: idle in the boundary phase
    const int my_j = lane - my_t * LT;
    float my_sh = sh[0];
#pragma unroll
    for (int t = 1; t < NT; t++) my_sh = (my_t == t) ? sh[t] : my_sh;
    const float inv_step = __fdiv_rn(1.0f, c.code_step);
    const float est_off = __fsub_rn(c.rem_code, my_sh);  // n ~ (k + rem - shift) / step
    auto chip_of = [&](int n) -> int { return raw_chip_std(__fmul_rn(c.code_step, static_cast<float>(n)), my_sh, c.rem_code); };
    float2 bacc = make_float2(0.0f, 0.0f);

    // the super-trip's samples for this lane: R / 2 coalesced 16-byte loads (samples outside the segment are zero); issued one super-trip ahead
    auto load_st = [&](int m, float4 (&v)[R / 2]) {
        const int a = a0 + m * S;
        const float2* __restrict__ src = base + static_cast<long long>(m) * S;
        const bool plain = (a >= c.n_begin) && (a + S <= c.n_end);  // uniform
#pragma unroll
        for (int q = 0; q < R / 2; q++)
            {
                const int s0 = 128 * q + 2 * lane;
                if (plain)
                    v[q] = *reinterpret_cast<const float4*>(src + s0);
                else
                    {
                        const int n0 = a + s0;
                        const bool v0 = (n0 >= c.n_begin) && (n0 < c.n_end), v1 = (n0 + 1 >= c.n_begin) && (n0 + 1 < c.n_end);
                        const float2 x0 = v0 ? src[s0] : make_float2(0.0f, 0.0f);
                        const float2 x1 = v1 ? src[s0 + 1] : make_float2(0.0f, 0.0f);
                        v[q] = make_float4(x0.x, x0.y, x1.x, x1.y);
                    }
            }
    };
    float4 nxt[R / 2];
#pragma unroll
    for (int q = 0; q < R / 2; q++) nxt[q] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (wave < n_st) load_st(wave, nxt);  // uniform

    int it = 0, it0 = 0;
    for (int m = wave; m < n_st; m += MC_WAVES, it++)
        {
            if (it - it0 >= SEEDS)  // uniform: long windows refill the table
                {
                    it0 = it;
                    Tb = fill_table(it0);
                }
            const v2f U = {readlane_f(Tb.x, it - it0), readlane_f(Tb.y, it - it0)};
            const int a = a0 + m * S;
            const int b = min(a + S, c.n_end);       // samples [n_lo, b) of this super-trip are inside the segment
            const int n_lo = max(a, c.n_begin);

            // ---- phase 1: natural order in LDS; then the next super-trip's loads go out and fly during phases 2 and 3
#pragma unroll
            for (int q = 0; q < R / 2; q++)
                {
                    const int s0 = 128 * q + 2 * lane;
                    *reinterpret_cast<float4*>(ybuf + (s0 / R) * STRIDE + (s0 % R)) = nxt[q];
                }
            asm volatile("" ::: "memory");  // LDS operations of one wave execute in order; keep the compiler from moving the reads up
            if (m + MC_WAVES < n_st) load_st(m + MC_WAVES, nxt);  // uniform

            // ---- phase 2: run prefixes in place, run totals scanned across the wave
            float2 T = make_float2(0.0f, 0.0f);
#ifndef GSH_RUNS_SKIP2
            {
                v2f x[R];
#pragma unroll
                for (int u = 0; u < R / 2; u++)
                    {
                        const float4 v = *reinterpret_cast<const float4*>(ybuf + lane * STRIDE + 2 * u);
                        x[2 * u] = (v2f){v.x, v.y};
                        x[2 * u + 1] = (v2f){v.z, v.w};
                    }
#pragma unroll
                for (int p = 1; p < R; p++) x[p] = x[p - 1] + pk_cmul(x[p], wp[p]);
#pragma unroll
                for (int u = 0; u < R / 2; u++)
                    *reinterpret_cast<float4*>(ybuf + lane * STRIDE + 2 * u) = make_float4(x[2 * u].x, x[2 * u].y, x[2 * u + 1].x, x[2 * u + 1].y);
                const v2f ph = pk_cmul(U, Li);
                const v2f tot = pk_cmul(x[R - 1], ph);
                const float2 incl = make_float2(wave_scan_incl(tot.x), wave_scan_incl(tot.y));
                epbuf[lane] = make_float4(incl.x - tot.x, incl.y - tot.y, ph.x, ph.y);
                T = make_float2(readlane_f(incl.x, 63), readlane_f(incl.y, 63));
            }
#endif
            asm volatile("" ::: "memory");

            // ---- phase 3: the chip boundaries of this lane's tap inside (n_lo, b)
#ifdef GSH_RUNS_SKIP3
            if (false)
#else
            if (my_t < NT)
#endif
                {
                    const int k_first = chip_of(n_lo);
                    const int count = chip_of(b - 1) - k_first;
                    if (my_j == 0)
                        {
                            const float c0 = tab[k_first + c.k_off];
                            bacc.x = fmaf(c0, T.x, bacc.x);
                            bacc.y = fmaf(c0, T.y, bacc.y);
                        }
                    for (int j = my_j; j < count; j += LT)
                        {
                            const int k = k_first + 1 + j;
                            // estimate, then the exact chain: idx(n - 1) < k <= idx(n), n in (n_lo, b - 1]
                            int n = static_cast<int>(ceilf(__fmul_rn(__fadd_rn(static_cast<float>(k), est_off), inv_step)));
                            n = min(max(n, n_lo + 1), b - 1);
                            while (n > n_lo + 1 && chip_of(n - 1) >= k) n--;
                            while (n < b - 1 && chip_of(n) < k) n++;
                            const int rel = n - a;
                            const int run = rel / R, pos = rel % R;
                            const float4 ep = epbuf[run];
                            const float2 qv = ybuf[run * STRIDE + max(pos - 1, 0)];
                            float2 P = make_float2(ep.x, ep.y);
                            if (pos > 0)
                                {
                                    const float2 r = cmul(qv, make_float2(ep.z, ep.w));
                                    P.x += r.x;
                                    P.y += r.y;
                                }
                            const float d = tab[k + c.k_off] - tab[k - 1 + c.k_off];
                            bacc.x = fmaf(d, T.x - P.x, bacc.x);
                            bacc.y = fmaf(d, T.y - P.y, bacc.y);
                        }
                }
            asm volatile("" ::: "memory");  // the next super-trip overwrites the buffers
        }
#pragma unroll
    for (int t = 0; t < NT; t++)
        {
            acc[t].x += (my_t == t) ? bacc.x : 0.0f;
            acc[t].y += (my_t == t) ? bacc.y : 0.0f;
        }
}

// KC: the whole code is staged behind a MARGIN-entry guard band (c.k_off == MC_MARGIN) at LDS address 0: look-ups use a constant offset
// PAIRK: zero-shift-prompt E/P/L jobs read their early tap next to the late one (the caller checked the job: mcorr_pair_eligible, multicorrelator.h)
template <int NT, int MODE, bool WRAP, bool ZP = false, bool AUX = false, bool KC = false, bool PAIRK = false>
__device__ __forceinline__ void run_segment(const JobCtx& c, const float2* __restrict__ base, const float* __restrict__ tab,
    const float (&sh)[NT], const int (&rot)[NT], float2 (&acc)[NT], float2* acc_aux = nullptr)
{
#if GSH_MC_PACKED
    if (!WRAP && MODE == 0 && c.packed && c.n_total < (1 << 24))
        {
#ifndef GSH_MC_MERGED
#define GSH_MC_MERGED 1
#endif
#ifndef GSH_MC_MERGED_BANK
#define GSH_MC_MERGED_BANK 0  // 1: the batched 256-thread kernel too (A/B switch: profiles/ab/r04/mcorr_merged_ab.txt -- slower there, the trip is bound by instruction issue)
#endif
            constexpr bool MRG = GSH_MC_MERGED && (MC_THREADS > 256 || GSH_MC_MERGED_BANK);  // one accumulator set (run_segment_packed): the 1 024-thread closed-loop kernel
#ifdef GSH_MC_NCH
            constexpr int NCH = (NT <= 3 && !AUX) ? GSH_MC_NCH : 1;
#else
#ifdef GSH_MC_NCH_WIDE  /* (A/B: two chunks per trip for the five-tap and pilot + data flavours of the merged form too) */
            constexpr int NCH = (MC_THREADS <= 256 || MRG) ? 2 : 1;
#else
            constexpr int NCH = (MC_THREADS <= 256 || (MRG && NT <= 3 && !AUX)) ? 2 : 1;
#endif
#endif
#ifdef GSH_MC_PREFETCH
            constexpr int PF = GSH_MC_PREFETCH;
#else
            constexpr int PF = (MC_THREADS <= 256) ? GSH_MC_PREFETCH_BANK : GSH_MC_PREFETCH_LOOP;
#endif
            if constexpr (PAIRK && NT == 3 && ZP && !AUX && PF == 1)
                {
                    // early read next to late (packed_trip); the caller vouches for the job: shifts exactly one chip apart, code running forward
                    run_segment_packed<NT, ZP, AUX, NCH, PF, true, KC, MRG>(c, base, tab, sh, acc, acc_aux);
                    return;
                }
            run_segment_packed<NT, ZP, AUX, NCH, PF, false, KC, MRG>(c, base, tab, sh, acc, acc_aux);
            return;
        }
#endif
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
                    const int k_body = k_full_begin;
                    // stride rotator exp(-j * 512 * step) and sample rotator exp(-j * step), both seeded exactly
                    const float2 w = expmj(static_cast<double>(2 * MC_PAIRS_PER_CHUNK) * static_cast<double>(c.phase_step));
                    const float2 inc = expmj(static_cast<double>(c.phase_step));
                    for (int kb = k_body; kb < k_full_end; kb += MC_RESEED)
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
#if defined(GSH_TRK_PROFILE) && GSH_TRK_PROFILE == 2
#define GSH_CW_STAMP(i)                                  \
    do                                                   \
        {                                                \
            if (threadIdx.x == 0) cw_stamp[i] = clock64(); \
        }                                                \
    while (0)
__shared__ long long cw_stamp[8];  // phase stamps of the last correlate_window call, thread 0's view (profiling builds only)
#else
#define GSH_CW_STAMP(i) \
    do                  \
        {               \
        }               \
    while (0)
#endif

// SUM = false: return after the first barrier, with one row of per-wave partial sums at red[GSH_MAX_TAPS * (1 + wave) + tap] and nothing in red[0..NOUT): the caller
// adds the rows itself (in wave order, as below, to get the same sums) in whichever waves need the result -- two barriers and one LDS round trip less per call.
struct NoHook
{
    __device__ __forceinline__ void operator()() const {}
};
// hook: called by every thread after its wave's partial sums are in LDS and before the barrier that ends the correlation -- a wave that has finished its
// trips early idles there (the oldest wave of a SIMD gets the issue slots first), which makes it the place for work nobody should wait for
// (tracking_loop.hip, live mode: the look-out for new samples).
template <int NT, int MODE, bool AUX = false, bool PAIRK = false, bool SUM = true, typename Hook = NoHook>
__device__ __forceinline__ void correlate_window(const float2* __restrict__ stream, unsigned long long sample_offset, int n_samples,
    const float* __restrict__ tab, int code_len, const float (&sh)[NT], float rem_carr, float phase_step, float phase_rate, float rem_code,
    float code_step, float code_rate, float2* __restrict__ red, const float* tab_aux = nullptr, float aux_shift = 0.0f, Hook hook = Hook(), const float2* seed_tab = nullptr,
    int seg_begin = 0, int seg_end = -1)
{
    // seg_begin / seg_end (round 6): correlate only the samples [seg_begin, seg_end) of the window -- one of several work-groups that share a window
    // (tracking_loop.hip, cooperating work-groups); the sample indices the chip look-ups and the phasors use stay those of the whole window.  A seed table handed
    // in must then have been formed for rem_carr + seg_begin * phase_step.
    static_assert(!AUX || (MODE == 0 && NT < GSH_MAX_TAPS), "the fused tap exists for the standard mode and needs a free slot in `red`");
    const int tid = threadIdx.x;
    GSH_CW_STAMP(0);
    JobCtx c;
    c.n_total = n_samples;
    c.code_len = code_len;
    c.rem_carr = rem_carr;
    c.phase_step = phase_step;
    c.phase_rate = phase_rate;
    c.rem_code = rem_code;
    c.code_step = code_step;
    c.code_rate = code_rate;
    if (seg_end < 0) seg_end = n_samples;
    c.n_begin = seg_begin;
    c.n_end = seg_end;
    const unsigned long long seg_offset = sample_offset + static_cast<unsigned long long>(seg_begin);
    const int odd = static_cast<int>(seg_offset & 1ULL);
    c.n_first = seg_begin - odd;
    c.seed_tab = seed_tab;
    const float2* __restrict__ base = stream + (seg_offset - static_cast<unsigned long long>(odd));
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
    GSH_CW_STAMP(1);
    if (seg_end > seg_begin)
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
            const int lo = raw_chip_std(__fmul_rn(code_step, static_cast<float>(seg_begin)), smin, rem_code);
            const int hi = raw_chip_std(__fmul_rn(code_step, static_cast<float>(seg_end - 1)), smax, rem_code);
            const bool fast = !mode_hd_code(MODE) && (code_step >= 0.0f) && (lo >= -MC_MARGIN) && (hi < code_len + MC_MARGIN) && (code_len >= MC_MARGIN);
            const bool zp = (NT & 1) && (sh[NT / 2] == 0.0f) && !mode_hd_code(MODE) && (n_samples < (1 << 24));
            if (fast && zp)
                run_segment<NT, MODE, false, true, AUX, false, PAIRK>(c, base, tab, sh, rot, acc, &acc_aux);  // PAIRK: the caller vouches for mcorr_pair_eligible taps
            else if (fast)
                run_segment<NT, MODE, false, false, AUX>(c, base, tab, sh, rot, acc, &acc_aux);
            else
                run_segment<NT, MODE, true, false, AUX>(c, base, tab, sh, rot, acc, &acc_aux);
        }
    GSH_CW_STAMP(2);
    // wave sums in DPP steps (the last lane holds them), one row of partials per wave behind the output row, one LDS step over the waves.
    // Outputs (red[0..NOUT)) and partials (red[GSH_MAX_TAPS ..)) do not overlap, so a call needs two barriers, not three: the caller reads the
    // outputs and passes a barrier of its own before the next call writes them again.
    // (the per-value form here: in the 1024-thread kernel the all-at-once v_add_f32_dpp form of the batched kernel keeps a dozen values live through 72
    //  instructions at the most crowded point of the function and sends registers to scratch)
#if GSH_MC_SCAN_ALL_AT_ONCE
    // (round 4: with one accumulator set the registers are there -- one v_add_f32_dpp per value and step instead of v_mov_b32_dpp + v_add_f32)
    if constexpr (mode_hd_code(MODE) == false && NT <= 3)
        {
            constexpr int NV = 2 * NT + (AUX ? 2 : 0);
            float v[NV];
#pragma unroll
            for (int t = 0; t < NT; t++)
                {
                    v[2 * t] = acc[t].x;
                    v[2 * t + 1] = acc[t].y;
                }
            if (AUX)
                {
                    v[2 * NT] = acc_aux.x;
                    v[2 * NT + 1] = acc_aux.y;
                }
            wave_scan_incl_n<NV>(v);
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = make_float2(v[2 * t], v[2 * t + 1]);
            if (AUX) acc_aux = make_float2(v[2 * NT], v[2 * NT + 1]);
        }
    else
#endif
        {
#pragma unroll
            for (int t = 0; t < NT; t++)
                {
                    acc[t].x = wave_scan_incl(acc[t].x);
                    acc[t].y = wave_scan_incl(acc[t].y);
                }
            if (AUX)
                {
                    acc_aux.x = wave_scan_incl(acc_aux.x);
                    acc_aux.y = wave_scan_incl(acc_aux.y);
                }
        }
    const int wave = tid >> 6;
    float2* const part = red + GSH_MAX_TAPS;
    if ((tid & 63) == 63)
        {
#pragma unroll
            for (int t = 0; t < NT; t++) part[wave * GSH_MAX_TAPS + t] = acc[t];
            if (AUX) part[wave * GSH_MAX_TAPS + NT] = acc_aux;
        }
    GSH_CW_STAMP(3);
    hook();
    __syncthreads();
    GSH_CW_STAMP(4);
    if constexpr (!SUM) return;
    constexpr int NOUT = AUX ? NT + 1 : NT;
    if (tid < NOUT)
        {
            float2 s = make_float2(0.0f, 0.0f);
#pragma unroll
            for (int w = 0; w < MC_WAVES; w++)
                {
                    s.x += part[w * GSH_MAX_TAPS + tid].x;
                    s.y += part[w * GSH_MAX_TAPS + tid].y;
                }
            red[tid] = s;
        }
    __syncthreads();
    GSH_CW_STAMP(5);
}

// After a correlate_window<..., SUM = false>: the sums of taps 0 .. NOUT-1, formed by lanes 0 .. NOUT-1 of the calling wave in the order the SUM = true
// form uses (wave 0's partial first) and handed to every lane of the wave as wave-uniform values.  Any wave may call it; reads only.
template <int NOUT>
__device__ __forceinline__ void sum_wave_partials(const float2* __restrict__ red, float2 (&out)[NOUT])
{
    const float2* const part = red + GSH_MAX_TAPS;
    const int lane = threadIdx.x & 63;
    const int t = lane < NOUT ? lane : 0;
    float2 s = make_float2(0.0f, 0.0f);
    if constexpr (MC_WAVES == 16)
        {
            // All sixteen rows are fetched before the first addition.  Left to itself the compiler (at the register limit of the 1 024-thread kernel, scheduling for
            // fewer live values everywhere) emitted read - wait - add eight times over: eight dependent LDS round trips, ~600 clocks of every period of the closed loop.
            typedef float f4 __attribute__((ext_vector_type(4)));
            f4 q[8];
#pragma unroll
            for (int i = 0; i < 8; i++)
                {
                    const float2 u = part[(2 * i) * GSH_MAX_TAPS + t], v = part[(2 * i + 1) * GSH_MAX_TAPS + t];
                    q[i] = f4{u.x, u.y, v.x, v.y};
                }
            asm volatile("" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]));  // (every row in a register here)
#pragma unroll
            for (int i = 0; i < 8; i++)  // the same additions in the same order
                {
                    s.x += q[i][0];
                    s.y += q[i][1];
                    s.x += q[i][2];
                    s.y += q[i][3];
                }
        }
    else
        {
#pragma unroll
            for (int w = 0; w < MC_WAVES; w++)
                {
                    s.x += part[w * GSH_MAX_TAPS + t].x;
                    s.y += part[w * GSH_MAX_TAPS + t].y;
                }
        }
#pragma unroll
    for (int k = 0; k < NOUT; k++)
        {
            out[k].x = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s.x), k));
            out[k].y = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s.y), k));
        }
}

// the standard-mode form the batched callers use.  PAIRK: E/P/L with the taps exactly one chip apart and the code running forward (mcorr_pair_eligible,
// multicorrelator.h) -- the early tap is read next to the late one (packed_trip); the sums are bit-identical either way
template <int NT, bool PAIRK = false, bool SUM = true, typename Hook = NoHook>
__device__ __forceinline__ void correlate_window_std(const float2* __restrict__ stream, unsigned long long sample_offset, int n_samples,
    const float* __restrict__ tab, int code_len, const float (&sh)[NT], float rem_carr, float phase_step, float rem_code, float code_step,
    float2* __restrict__ red, Hook hook = Hook(), const float2* seed_tab = nullptr, int seg_begin = 0, int seg_end = -1)
{
    correlate_window<NT, 0, false, PAIRK, SUM, Hook>(stream, sample_offset, n_samples, tab, code_len, sh, rem_carr, phase_step, 0.0f, rem_code, code_step, 0.0f, red, nullptr, 0.0f, hook, seed_tab,
        seg_begin, seg_end);
}

// standard mode with the fused data-component tap: red[0..NT) the taps, red[NT] the fused one
template <int NT, bool SUM = true, typename Hook = NoHook>
__device__ __forceinline__ void correlate_window_std_aux(const float2* __restrict__ stream, unsigned long long sample_offset, int n_samples,
    const float* __restrict__ tab, const float* tab_aux, float aux_shift, int code_len, const float (&sh)[NT], float rem_carr, float phase_step,
    float rem_code, float code_step, float2* __restrict__ red, Hook hook = Hook(), const float2* seed_tab = nullptr, int seg_begin = 0, int seg_end = -1)
{
    correlate_window<NT, 0, true, false, SUM, Hook>(stream, sample_offset, n_samples, tab, code_len, sh, rem_carr, phase_step, 0.0f, rem_code, code_step, 0.0f, red, tab_aux, aux_shift, hook, seed_tab,
        seg_begin, seg_end);
}
}  // namespace GSH_MC_NS
namespace mcdev = GSH_MC_NS;
}  // namespace gsh
#endif
