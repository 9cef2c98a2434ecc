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
#include "multicorrelator.h"
#include <cmath>

namespace gsh
{
namespace
{
constexpr int MC_THREADS = 256;
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
};

// One chunk = 256 pairs = 512 consecutive samples; thread `tid` owns samples n0, n0+1.
template <int NT, int MODE, bool WRAP, bool MASKED>
__device__ __forceinline__ void process_pair(const JobCtx& c, const float2* __restrict__ base, const float* __restrict__ tab,
    const float (&sh)[NT], const int (&rot)[NT], int pair, float2 pa, float2 pb, float2 (&acc)[NT])
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
            const float a0 = __fmul_rn(c.code_step, static_cast<float>(n0));
            const float a1 = __fmul_rn(c.code_step, static_cast<float>(n0 + 1));
#pragma unroll
            for (int t = 0; t < NT; t++)
                {
                    int k0 = raw_chip_std(a0, sh[t], c.rem_code);
                    int k1 = raw_chip_std(a1, sh[t], c.rem_code);
                    if (WRAP)
                        {
                            k0 = wrap_chip(k0, c.code_len);
                            k1 = wrap_chip(k1, c.code_len);
                        }
                    if (MASKED)
                        {
                            // masked lanes may sit at n = -1 / n = n_end with any index: keep the lookup in range
                            k0 = wrap_chip(k0, c.code_len);
                            k1 = wrap_chip(k1, c.code_len);
                        }
                    const float c0 = tab[k0 + MC_MARGIN];
                    const float c1 = tab[k1 + MC_MARGIN];
                    acc[t].x = fmaf(y0.x, c0, acc[t].x);
                    acc[t].y = fmaf(y0.y, c0, acc[t].y);
                    acc[t].x = fmaf(y1.x, c1, acc[t].x);
                    acc[t].y = fmaf(y1.y, c1, acc[t].y);
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
                    const float c0 = tab[k0 + MC_MARGIN];
                    const float c1 = tab[k1 + MC_MARGIN];
                    acc[t].x = fmaf(y0.x, c0, acc[t].x);
                    acc[t].y = fmaf(y0.y, c0, acc[t].y);
                    acc[t].x = fmaf(y1.x, c1, acc[t].x);
                    acc[t].y = fmaf(y1.y, c1, acc[t].y);
                }
        }
}

template <int NT, int MODE, bool WRAP>
__device__ __forceinline__ void run_segment(const JobCtx& c, const float2* __restrict__ base, const float* __restrict__ tab,
    const float (&sh)[NT], const int (&rot)[NT], float2 (&acc)[NT])
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
                    process_pair<NT, MODE, WRAP, true>(c, base, tab, sh, rot, pair, pa, pb, acc);
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
                            process_pair<NT, MODE, WRAP, false>(c, base, tab, sh, rot, pair, pa, pb, acc);
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
#pragma unroll 2
                            for (int i = 0; i < cnt; i++)
                                {
                                    process_pair<NT, MODE, WRAP, false>(c, base, tab, sh, rot, pair, pa, pb, acc);
                                    pa = cmul(pa, w);
                                    pb = cmul(pb, w);
                                    pair += MC_PAIRS_PER_CHUNK;
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
                    process_pair<NT, MODE, WRAP, true>(c, base, tab, sh, rot, pair, pa, pb, acc);
                }
        }
}

template <int NT, int MODE>
__global__ __launch_bounds__(MC_THREADS) void mcorr_kernel(McorrArgs a)
{
    extern __shared__ __align__(16) float lds[];
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int job = static_cast<int>(lb) / a.splits;
    const int split = static_cast<int>(lb) - job * a.splits;
    const gsh_corr_job& J = a.jobs[job];
    const int tid = threadIdx.x;

    JobCtx c;
    c.n_total = J.n_samples;
    c.code_len = a.code_lens[J.code_slot];
    c.rem_carr = J.rem_carr_phase_rad;
    c.phase_step = J.phase_step_rad;
    c.phase_rate = J.phase_rate_step_rad;
    c.rem_code = J.rem_code_phase_chips;
    c.code_step = J.code_phase_step_chips;
    c.code_rate = J.code_phase_rate_step_chips;

    // ---- stage the local code (+ guard bands holding the wrapped neighbours) in LDS
    float* tab = lds;
    const int tab_len = c.code_len + 2 * MC_MARGIN;
    {
        const float* __restrict__ gcode = a.codes + static_cast<size_t>(J.code_slot) * a.code_stride;
        for (int i = tid; i < tab_len; i += MC_THREADS) tab[i] = gcode[wrap_chip(i - MC_MARGIN, c.code_len)];
    }
    float2* red = reinterpret_cast<float2*>(lds + ((tab_len + 3) & ~3));

    // ---- this work-group's slice of the window
    int seg = (c.n_total + a.splits - 1) / a.splits;
    seg = (seg + 1) & ~1;
    c.n_begin = min(split * seg, c.n_total);
    c.n_end = min(c.n_total, c.n_begin + seg);
    const unsigned long long abs0 = J.sample_offset + static_cast<unsigned long long>(c.n_begin);
    const int odd = static_cast<int>(abs0 & 1ULL);
    c.n_first = c.n_begin - odd;
    const float2* __restrict__ base = a.stream + (abs0 - static_cast<unsigned long long>(odd));  // 16-byte aligned

    float sh[NT];
    int rot[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) sh[t] = (t < J.n_taps) ? J.shifts_chips[t] : 0.0f;
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

    float2 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = make_float2(0.0f, 0.0f);

    __syncthreads();  // code table visible

    if (c.n_end > c.n_begin)
        {
            bool fast = false;
            if (!mode_hd_code(MODE))
                {
                    // the raw index is monotone in n and in the shift (rounding is monotone) when step >= 0:
                    // bound it over the segment and skip the per-sample wrap when it stays inside the guard bands
                    float smin = sh[0], smax = sh[0];
#pragma unroll
                    for (int t = 1; t < NT; t++)
                        {
                            smin = fminf(smin, sh[t]);
                            smax = fmaxf(smax, sh[t]);
                        }
                    const int lo = raw_chip_std(__fmul_rn(c.code_step, static_cast<float>(c.n_begin)), smin, c.rem_code);
                    const int hi = raw_chip_std(__fmul_rn(c.code_step, static_cast<float>(c.n_end - 1)), smax, c.rem_code);
                    fast = (c.code_step >= 0.0f) && (lo >= -MC_MARGIN) && (hi < c.code_len + MC_MARGIN) && (c.code_len >= MC_MARGIN);
                }
            if (fast)
                run_segment<NT, MODE, false>(c, base, tab, sh, rot, acc);
            else
                run_segment<NT, MODE, true>(c, base, tab, sh, rot, acc);
        }

    // ---- integrate-and-dump: wave64 shuffle tree, then one LDS step over the 4 waves
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
    const int wave = tid >> 6;
    if ((tid & 63) == 0)
        {
#pragma unroll
            for (int t = 0; t < NT; t++) red[wave * GSH_MAX_TAPS + t] = acc[t];
        }
    __syncthreads();
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
    const dim3 grid(static_cast<unsigned>(a.n_jobs) * static_cast<unsigned>(a.splits));
    const dim3 block(MC_THREADS);
    switch (mode)
        {
        case 0:
            hipLaunchKernelGGL((mcorr_kernel<NT, 0>), grid, block, lds, stream, a);
            break;
        case 1:
            hipLaunchKernelGGL((mcorr_kernel<NT, 1>), grid, block, lds, stream, a);
            break;
        case 2:
            hipLaunchKernelGGL((mcorr_kernel<NT, 2>), grid, block, lds, stream, a);
            break;
        default:
            return set_error(GSH_ERR_INVALID, "unknown correlator mode %d", mode);
        }
    GSH_HIP(hipGetLastError());
    return GSH_OK;
}
}  // namespace

size_t mcorr_lds_bytes(int max_code_len)
{
    const size_t tab = (static_cast<size_t>(max_code_len) + 2 * MC_MARGIN + 3) & ~static_cast<size_t>(3);
    return tab * sizeof(float) + MC_WAVES * GSH_MAX_TAPS * sizeof(float2);
}

int mcorr_launch(const McorrArgs& a, int max_taps, int mode, int max_code_len, hipStream_t stream)
{
    if (a.n_jobs <= 0) return GSH_OK;
    GSH_REQUIRE(max_taps >= 1 && max_taps <= GSH_MAX_TAPS, "n_taps %d outside 1..%d", max_taps, GSH_MAX_TAPS);
    GSH_REQUIRE(a.splits >= 1, "splits must be >= 1");
    const size_t lds = mcorr_lds_bytes(max_code_len);
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
}  // namespace gsh
