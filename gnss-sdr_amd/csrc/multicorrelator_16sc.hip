// The 16-bit correlator family (the row's tail): Cpu_Multicorrelator_16sc (T/cpu_multicorrelator_16sc.cc) over
// K/volk_gnsssdr_16ic_xn_resampler_16ic_xn.h:60-78 and K/volk_gnsssdr_16ic_x2_rotator_dot_prod_16ic_xn.h:66-102 (the generic protokernels), bit for bit.
//
// What the reference computes per call, n ascending (samples, code and sums are complex int16):
//     y[n]   = int16( rintf( float(x[n]) * phi[n] ) )                  float32 complex product, round to nearest even, low 16 bits
//     phi    = phi / hypotf(phi)   at n = 0, 256, 512, ...             AFTER phi[n] has been used
//     phi    = phi * inc                                               float32, one rounding per operation
//     acc[t] = sat16( acc[t] + int16( y[n] * code[k_t(n)] ) )          low 16 bits of the complex integer product, every addition saturates
// with k_t(n) the float32 chip index of the 32f resampler.  Two things here are sequential by construction and decide the layout:
//   * phi[n] is a float32 recurrence with its roundings -- there is no closed form that reproduces them.  rot16_kernel walks it with ONE LANE PER JOB (lanes
//     0 .. J-1 of a wave, 64 steps at a time, phasors left in LDS) and then lets all 64 lanes rotate those 64 samples of each of the J jobs (coalesced 4-byte
//     loads of x, coalesced stores of y into a scratch row per job).  J is chosen so that the launch has about one wave per SIMD: the chain's instructions are
//     issued once per wave whatever J is, the rotation's once per job.
//   * saturating sums are not associative -- but the MAPS x -> clamp(x + a, lo, hi) are closed under composition (g2 o g1 = clamp(x + a1 + a2,
//     clamp(lo1 + a2, lo2, hi2), clamp(hi1 + a2, lo2, hi2))), composition is associative, and a + 65 535 already sends every int16 to the upper bound, so the
//     offset can be kept within +-65 535 without changing the map.  corr16_kernel gives every lane ONE contiguous run of the window (staged through LDS in
//     tiles so that global loads stay coalesced), composes the run's maps in order, and folds the 256 lanes' maps in lane order once per job; the job's sum is
//     the composed map at 0.  Same result as the reference's loop for any input, including ones that saturate.
// The phasors phi[0] = (cos r, -sin r) and inc = exp(-j s) are formed on the HOST with the C library (exactly the expressions of
// cpu_multicorrelator_16sc.cc:89-93), because the recurrence amplifies nothing but also forgives nothing: a last-bit difference in inc is a different result.
#include "gsh_internal.h"
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdlib>
#include <new>
#include <vector>

namespace gsh
{
namespace
{
struct Job16Dev  // one reference call, phasors formed
{
    unsigned long long sample_offset;
    int n_samples;
    int code_slot;
    float ph_re, ph_im, inc_re, inc_im;
    float rem_code, code_step;
    int n_taps;
    int pad;
    float shifts[GSH_MAX_TAPS];
};
static_assert(sizeof(Job16Dev) == 80, "job table layout");

constexpr int ROT_BLOCK = 64;             // samples of a job rotated per step (one per lane)
constexpr int PH_ROW = ROT_BLOCK + 1;     // phasors of one job in LDS, padded
constexpr int C16_THREADS = 256;
constexpr int C16_R = 32;                 // samples of a lane's run staged per tile
constexpr int C16_ROW = C16_R + 1;        // padded: lane t reads word t * 33 + p -> bank (t + p) % 32

// glibc's hypotf: the square root, in double, of the exact sum of the two exact squares, rounded to float
__device__ __forceinline__ float hypot_as_libm(float a, float b)
{
    const double s = __dadd_rn(__dmul_rn(static_cast<double>(a), static_cast<double>(a)), __dmul_rn(static_cast<double>(b), static_cast<double>(b)));
    return static_cast<float>(__dsqrt_rn(s));
}

// one IEEE float32 operation each, as single instructions the compiler cannot pair into packed ones (it otherwise forms v_pk_mul / v_pk_add pairs that compute both
// the sum and the difference and move half of each away).  Measured: no faster -- a lone wave issues a dependent instruction every ~9 clocks either way, ~62 clocks
// per step of the chain, which is what bounds a single call (25 000 steps ~ 0.7 ms); kept because the six operations are then exactly the six the reference performs.
__device__ __forceinline__ float mul1(float a, float b)
{
    float r;
    asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float add1(float a, float b)
{
    float r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float sub1(float a, float b)
{
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ int low16(int v) { return static_cast<int>(static_cast<short>(v)); }
__device__ __forceinline__ int clamp16(int v) { return min(max(v, -32768), 32767); }
__device__ __forceinline__ int med(int v, int lo, int hi) { return min(max(v, lo), hi); }

// ---- kernel 1: the phasor chain and the rotated, rounded samples ---------------------------------------------------------------------------------------
// JW: jobs per wave, a compile-time bound so that the block's samples of all JW jobs are LOADED BEFORE the chain is walked (registers, one short2 per job) --
// the chain's 64 dependent steps then cover the loads' latency instead of every job's load being waited for in turn.
template <int JW>
__global__ __launch_bounds__(ROT_BLOCK) void rot16_kernel(const Job16Dev* __restrict__ jobs, int n_jobs, const short2* __restrict__ stream, short2* __restrict__ rot,
    unsigned long long rot_stride)
{
    __shared__ float2 ph[JW * PH_ROW];
    const int lane = threadIdx.x;
    const int j0 = static_cast<int>(blockIdx.x) * JW;
    const int J = min(JW, n_jobs - j0);
    float pr = 1.0f, pi = 0.0f, ir = 1.0f, ii = 0.0f;
    int n_mine = 0;
    if (lane < J)
        {
            const Job16Dev& jb = jobs[j0 + lane];
            pr = jb.ph_re;
            pi = jb.ph_im;
            ir = jb.inc_re;
            ii = jb.inc_im;
            n_mine = jb.n_samples;
        }
    int n_max = n_mine;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) n_max = max(n_max, __shfl_xor(n_max, off));
    float2* const my_row = ph + (lane < JW ? lane : 0) * PH_ROW;
    // the samples of block b + 64 are requested before block b's chain is walked and used one trip later: a trip's arithmetic covers the loads' latency
    short2 x_next[JW];
    auto request = [&](int b) {
        const int n = b + lane;
#pragma unroll
        for (int j = 0; j < JW; j++)
            {
                x_next[j] = make_short2(0, 0);
                if (j < J)
                    {
                        const Job16Dev& jb = jobs[j0 + j];  // uniform
                        if (n < jb.n_samples) x_next[j] = stream[jb.sample_offset + static_cast<unsigned long long>(n)];
                    }
            }
    };
    request(0);
    for (int b = 0; b < n_max; b += ROT_BLOCK)
        {
            short2 x[JW];
            const int n = b + lane;
#pragma unroll
            for (int j = 0; j < JW; j++) x[j] = x_next[j];
            if (b + ROT_BLOCK < n_max) request(b + ROT_BLOCK);
            if (b < n_mine)  // (lanes >= J have n_mine = 0)
                {
                    // phi[b]: used, then renormalised where b is a multiple of 256, then advanced (K/..16ic_x2_rotator_dot_prod_16ic_xn.h:77-92)
                    my_row[0] = make_float2(pr, pi);
                    if ((b & 255) == 0)
                        {
                            const float h = hypot_as_libm(pr, pi);
                            pr = __fdiv_rn(pr, h);
                            pi = __fdiv_rn(pi, h);
                        }
                    {
                        const float nr = sub1(mul1(pr, ir), mul1(pi, ii)), ni = add1(mul1(pr, ii), mul1(pi, ir));
                        pr = nr;
                        pi = ni;
                    }
#pragma unroll 9
                    for (int s = 1; s < ROT_BLOCK; s++)
                        {
                            my_row[s] = make_float2(pr, pi);
                            const float nr = sub1(mul1(pr, ir), mul1(pi, ii)), ni = add1(mul1(pr, ii), mul1(pi, ir));
                            pr = nr;
                            pi = ni;
                        }
                }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < JW; j++)
                if (j < J)
                    {
                        const Job16Dev& jb = jobs[j0 + j];  // uniform
                        if (n < jb.n_samples)
                            {
                                const float2 p = ph[j * PH_ROW + lane];
                                const float xr = static_cast<float>(x[j].x), xi = static_cast<float>(x[j].y);
                                const float yr = __fsub_rn(__fmul_rn(xr, p.x), __fmul_rn(xi, p.y)), yi = __fadd_rn(__fmul_rn(xr, p.y), __fmul_rn(xi, p.x));
                                short2 y;
                                y.x = static_cast<short>(static_cast<int>(rintf(yr)));
                                y.y = static_cast<short>(static_cast<int>(rintf(yi)));
                                rot[static_cast<unsigned long long>(j0 + j) * rot_stride + static_cast<unsigned long long>(n)] = y;
                            }
                    }
            __syncthreads();
        }
}

// ---- kernel 2: chip selection, integer products, saturating sums as composed clamp maps ------------------------------------------------------------------
struct Map16  // x -> clamp(x + a, lo, hi), -32768 <= lo <= hi <= 32767, |a| <= 65535 after canon()
{
    int a, lo, hi;
};
__device__ __forceinline__ Map16 map_identity() { return Map16{0, -32768, 32767}; }
__device__ __forceinline__ void map_push(Map16& g, int p)  // the map of one more saturating addition of p, applied after g
{
    g.a += p;
    g.lo = clamp16(g.lo + p);
    g.hi = clamp16(g.hi + p);
}
__device__ __forceinline__ Map16 map_then(const Map16& g1, const Map16& g2)  // g2 o g1
{
    Map16 r;
    r.a = med(g1.a + g2.a, -65535, 65535);
    r.lo = med(g1.lo + g2.a, g2.lo, g2.hi);
    r.hi = med(g1.hi + g2.a, g2.lo, g2.hi);
    return r;
}
__device__ __forceinline__ Map16 map_canon(Map16 g)
{
    g.a = med(g.a, -65535, 65535);
    return g;
}

// K/..resampler..:73-76 -- floor((step * n + shift) - rem) in float32, then into [0, L).  `base` (a multiple of L, kept per tap by the caller) follows the index
// along the lane's run: consecutive samples are at most a chip or so apart, so one conditional step either way brings the remainder home; anything further off
// (the first sample of a run, a huge step) takes the division.
__device__ __forceinline__ int chip16(float a, float shift, float rem, int L, int& base)
{
    const float c = __fsub_rn(__fadd_rn(a, shift), rem);
    int k;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(k) : "v"(c));  // (int)floor(c)
    int r = k - base;
    const int up = r >= L ? L : 0;
    r -= up;
    base += up;
    const int down = r < 0 ? L : 0;
    r += down;
    base -= down;
    if (static_cast<unsigned>(r) >= static_cast<unsigned>(L))
        {
            r = ((k % L) + L) % L;
            base = k - r;
        }
    return r;
}

template <int NT>
__global__ __launch_bounds__(C16_THREADS) void corr16_kernel(const Job16Dev* __restrict__ jobs, const short2* __restrict__ codes, const int* __restrict__ code_lens,
    int max_code_len, const short2* __restrict__ rot, unsigned long long rot_stride, short2* __restrict__ out)
{
    extern __shared__ int lds16[];
    short2* const code = reinterpret_cast<short2*>(lds16);
    short2* const tile = code + max_code_len;
    const int tid = threadIdx.x;
    const Job16Dev& jb = jobs[blockIdx.x];
    const int n = jb.n_samples;
    const int L = code_lens[jb.code_slot];
    const int n_taps = jb.n_taps;
    {
        const short2* src = codes + static_cast<size_t>(jb.code_slot) * static_cast<size_t>(max_code_len);
        for (int i = tid; i < L; i += C16_THREADS) code[i] = src[i];
    }
    const float step = jb.code_step, rem = jb.rem_code;
    float sh[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) sh[t] = jb.shifts[t < n_taps ? t : 0];
    Map16 gr[NT], gi[NT];
    int base[NT];
#pragma unroll
    for (int t = 0; t < NT; t++)
        {
            gr[t] = gi[t] = map_identity();
            base[t] = 0;
        }
    const int Q = (n + C16_THREADS - 1) / C16_THREADS;  // lane t owns the samples [t Q, min(n, (t + 1) Q))
    const short2* const src = rot + static_cast<unsigned long long>(blockIdx.x) * rot_stride;
    for (int k0 = 0; k0 < Q; k0 += C16_R)
        {
            __syncthreads();  // the previous tile has been consumed (first trip: the code is staged)
#pragma unroll 8
            for (int i = 0; i < C16_R; i++)
                {
                    const int e = tid + C16_THREADS * i;
                    const int seg = e / C16_R, pos = e % C16_R;
                    const int q = k0 + pos;
                    const int nn = seg * Q + q;
                    short2 v = make_short2(0, 0);
                    if (q < Q && nn < n) v = src[nn];
                    tile[seg * C16_ROW + pos] = v;
                }
            __syncthreads();
            const int first = tid * Q + k0;
            const int cnt = min(min(C16_R, Q - k0), n - first);
            for (int p = 0; p < cnt; p++)
                {
                    const short2 w = tile[tid * C16_ROW + p];
                    const int wr = w.x, wi = w.y;
                    const float a = __fmul_rn(step, static_cast<float>(static_cast<unsigned>(first + p)));
#pragma unroll
                    for (int t = 0; t < NT; t++)
                        if (t < n_taps)
                            {
                                const short2 c = code[chip16(a, sh[t], rem, L, base[t])];
                                const int cr = c.x, ci = c.y;
                                map_push(gr[t], low16(wr * cr - wi * ci));
                                map_push(gi[t], low16(wr * ci + wi * cr));
                            }
                }
        }
    // the lanes' maps in lane order (lane t's samples come before lane t + 1's): a tree over the wave, then the four waves in order
#pragma unroll
    for (int t = 0; t < NT; t++)
        {
            gr[t] = map_canon(gr[t]);
            gi[t] = map_canon(gi[t]);
        }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1)
        {
#pragma unroll
            for (int t = 0; t < NT; t++)
                {
                    Map16 o;
                    o.a = __shfl_down(gr[t].a, off);
                    o.lo = __shfl_down(gr[t].lo, off);
                    o.hi = __shfl_down(gr[t].hi, off);
                    gr[t] = map_then(gr[t], o);  // (only lanes that are multiples of 2 off hold a meaningful map afterwards; the others are never read by one that does)
                    o.a = __shfl_down(gi[t].a, off);
                    o.lo = __shfl_down(gi[t].lo, off);
                    o.hi = __shfl_down(gi[t].hi, off);
                    gi[t] = map_then(gi[t], o);
                }
        }
    __syncthreads();  // the tile is free
    int* const fold = reinterpret_cast<int*>(tile);  // [wave][tap][6]
    if ((tid & 63) == 0)
        {
#pragma unroll
            for (int t = 0; t < NT; t++)
                {
                    int* f = fold + ((tid >> 6) * NT + t) * 6;
                    f[0] = gr[t].a;
                    f[1] = gr[t].lo;
                    f[2] = gr[t].hi;
                    f[3] = gi[t].a;
                    f[4] = gi[t].lo;
                    f[5] = gi[t].hi;
                }
        }
    __syncthreads();
    if (tid < GSH_MAX_TAPS)
        {
            short2 res = make_short2(0, 0);
            if (tid < n_taps && tid < NT)
                {
                    Map16 r = map_identity(), i = map_identity();
                    for (int w = 0; w < C16_THREADS / 64; w++)
                        {
                            const int* f = fold + (w * NT + tid) * 6;
                            r = map_then(r, Map16{f[0], f[1], f[2]});
                            i = map_then(i, Map16{f[3], f[4], f[5]});
                        }
                    res.x = static_cast<short>(med(r.a, r.lo, r.hi));  // the composed map at 0
                    res.y = static_cast<short>(med(i.a, i.lo, i.hi));
                }
            out[static_cast<size_t>(blockIdx.x) * GSH_MAX_TAPS + tid] = res;
        }
}
}  // namespace
}  // namespace gsh

// ---- host side ---------------------------------------------------------------------------------------------------------------------------------------
struct gsh_bank16
{
    int device{0};
    hipStream_t stream{nullptr};
    int n_slots{0};
    int max_code_len{0};
    short2* d_codes{nullptr};
    int* d_code_lens{nullptr};
    std::vector<int> h_code_lens;
    short2* d_stream_owned{nullptr};
    size_t stream_owned_cap{0};
    const short2* d_stream{nullptr};
    unsigned long long stream_len{0};
    gsh::Job16Dev* d_jobs{nullptr};
    short2* d_out{nullptr};
    int jobs_cap{0};
    short2* d_rot{nullptr};
    size_t rot_cap{0};  // samples
    std::vector<gsh::Job16Dev> h_jobs;
    std::vector<short2> h_out;
    int n_jobs{0};
    int max_taps{0};
    unsigned long long rot_stride{0};
    hipEvent_t ev0{nullptr}, ev1{nullptr};
};

namespace
{
using gsh::set_error;

int bank16_launch(gsh_bank16* b)
{
    if (b->n_jobs <= 0) return GSH_OK;
    int sims = 1024;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, b->device) == hipSuccess && prop.multiProcessorCount > 0) sims = 4 * prop.multiProcessorCount;
    }
    // jobs per wave: about GSH_M16_WAVES_PER_SIMD waves per SIMD (the chain's instructions are issued once per wave, so few, well-filled waves; a second wave per SIMD
    // fills the first one's dependent-issue gaps)
    static const int per_simd = [] { const char* e = std::getenv("GSH_M16_WAVES_PER_SIMD"); const int v = e != nullptr ? std::atoi(e) : 0; return v > 0 ? v : 2; }();
    const int want = (b->n_jobs + sims * per_simd - 1) / (sims * per_simd);
    int jpw = 1;
    while (jpw < want && jpw < 16) jpw *= 2;  // (beyond 16 the jobs' scalars no longer fit the SGPR file: more waves per SIMD instead)
    const dim3 blk(gsh::ROT_BLOCK);
#define GSH_ROT16(JW)                                                                                                                                              \
    hipLaunchKernelGGL((gsh::rot16_kernel<JW>), dim3(static_cast<unsigned>((b->n_jobs + JW - 1) / JW)), blk, 0, b->stream, b->d_jobs, b->n_jobs, b->d_stream, b->d_rot, \
        b->rot_stride)
    switch (jpw)
        {
        case 1: GSH_ROT16(1); break;
        case 2: GSH_ROT16(2); break;
        case 4: GSH_ROT16(4); break;
        case 8: GSH_ROT16(8); break;
        default: GSH_ROT16(16); break;
        }
#undef GSH_ROT16
    const size_t lds_b = (static_cast<size_t>(b->max_code_len) + static_cast<size_t>(gsh::C16_THREADS) * gsh::C16_ROW) * sizeof(short2);
    const dim3 grid(static_cast<unsigned>(b->n_jobs)), block(gsh::C16_THREADS);
    if (b->max_taps <= 1)
        hipLaunchKernelGGL((gsh::corr16_kernel<1>), grid, block, lds_b, b->stream, b->d_jobs, b->d_codes, b->d_code_lens, b->max_code_len, b->d_rot, b->rot_stride, b->d_out);
    else if (b->max_taps <= 3)
        hipLaunchKernelGGL((gsh::corr16_kernel<3>), grid, block, lds_b, b->stream, b->d_jobs, b->d_codes, b->d_code_lens, b->max_code_len, b->d_rot, b->rot_stride, b->d_out);
    else if (b->max_taps <= 5)
        hipLaunchKernelGGL((gsh::corr16_kernel<5>), grid, block, lds_b, b->stream, b->d_jobs, b->d_codes, b->d_code_lens, b->max_code_len, b->d_rot, b->rot_stride, b->d_out);
    else
        hipLaunchKernelGGL((gsh::corr16_kernel<GSH_MAX_TAPS>), grid, block, lds_b, b->stream, b->d_jobs, b->d_codes, b->d_code_lens, b->max_code_len, b->d_rot, b->rot_stride,
            b->d_out);
    GSH_HIP(hipGetLastError());
    return GSH_OK;
}
}  // namespace

extern "C"
{
    int gsh_bank16_create(int device, int n_code_slots, int max_code_length, gsh_bank16_t** out)
    {
        GSH_REQUIRE(out != nullptr, "null out pointer");
        *out = nullptr;
        GSH_REQUIRE(n_code_slots >= 1 && n_code_slots <= 4096, "n_code_slots %d outside 1..4096", n_code_slots);
        // both kernels' LDS: the code (4 bytes per chip) beside a 33 KB tile; 160 KB per compute unit
        GSH_REQUIRE(max_code_length >= 1 && max_code_length <= 30000, "max_code_length %d outside 1..30000", max_code_length);
        const int rc = gsh::use_device(device);
        if (rc != GSH_OK) return rc;
        gsh_bank16* b = new (std::nothrow) gsh_bank16();
        if (b == nullptr) return set_error(GSH_ERR_HIP, "out of host memory");
        b->device = device;
        b->n_slots = n_code_slots;
        b->max_code_len = max_code_length;
        b->h_code_lens.assign(static_cast<size_t>(n_code_slots), 0);
        auto fail = [&](hipError_t e, const char* what) {
            const int r = gsh::hip_fail(e, what, __FILE__, __LINE__);
            gsh_bank16_destroy(b);
            return r;
        };
        hipError_t e;
        if ((e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking)) != hipSuccess) return fail(e, "hipStreamCreate");
        if ((e = hipMalloc(&b->d_codes, sizeof(short2) * static_cast<size_t>(n_code_slots) * static_cast<size_t>(max_code_length))) != hipSuccess) return fail(e, "hipMalloc(codes)");
        if ((e = hipMalloc(&b->d_code_lens, sizeof(int) * static_cast<size_t>(n_code_slots))) != hipSuccess) return fail(e, "hipMalloc(code_lens)");
        if ((e = hipMemset(b->d_code_lens, 0, sizeof(int) * static_cast<size_t>(n_code_slots))) != hipSuccess) return fail(e, "hipMemset");
        if ((e = hipEventCreate(&b->ev0)) != hipSuccess) return fail(e, "hipEventCreate");
        if ((e = hipEventCreate(&b->ev1)) != hipSuccess) return fail(e, "hipEventCreate");
        {
            const size_t lds_b = (static_cast<size_t>(max_code_length) + static_cast<size_t>(gsh::C16_THREADS) * gsh::C16_ROW) * sizeof(short2);
            if (lds_b > 48 * 1024)
                {
                    const int bytes = static_cast<int>(lds_b);
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gsh::corr16_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gsh::corr16_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gsh::corr16_kernel<5>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gsh::corr16_kernel<GSH_MAX_TAPS>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
                }
        }
        *out = b;
        return GSH_OK;
    }

    void gsh_bank16_destroy(gsh_bank16_t* b)
    {
        if (b == nullptr) return;
        (void)hipSetDevice(b->device);
        if (b->stream) (void)hipStreamSynchronize(b->stream);
        if (b->d_codes) (void)hipFree(b->d_codes);
        if (b->d_code_lens) (void)hipFree(b->d_code_lens);
        if (b->d_stream_owned) (void)hipFree(b->d_stream_owned);
        if (b->d_jobs) (void)hipFree(b->d_jobs);
        if (b->d_out) (void)hipFree(b->d_out);
        if (b->d_rot) (void)hipFree(b->d_rot);
        if (b->ev0) (void)hipEventDestroy(b->ev0);
        if (b->ev1) (void)hipEventDestroy(b->ev1);
        if (b->stream) (void)hipStreamDestroy(b->stream);
        delete b;
    }

    int gsh_bank16_set_code(gsh_bank16_t* b, int slot, const int16_t* code_iq, int code_length)
    {
        GSH_REQUIRE(b != nullptr && code_iq != nullptr, "null argument");
        GSH_REQUIRE(slot >= 0 && slot < b->n_slots, "slot %d outside 0..%d", slot, b->n_slots - 1);
        GSH_REQUIRE(code_length >= 1 && code_length <= b->max_code_len, "code_length %d outside 1..%d", code_length, b->max_code_len);
        GSH_HIP(hipSetDevice(b->device));
        GSH_HIP(hipStreamSynchronize(b->stream));
        GSH_HIP(hipMemcpy(b->d_codes + static_cast<size_t>(slot) * static_cast<size_t>(b->max_code_len), code_iq, sizeof(short2) * static_cast<size_t>(code_length), hipMemcpyHostToDevice));
        GSH_HIP(hipMemcpy(b->d_code_lens + slot, &code_length, sizeof(int), hipMemcpyHostToDevice));
        b->h_code_lens[static_cast<size_t>(slot)] = code_length;
        return GSH_OK;
    }

    int gsh_bank16_set_stream_host(gsh_bank16_t* b, const int16_t* iq, uint64_t n_samples)
    {
        GSH_REQUIRE(b != nullptr && (iq != nullptr || n_samples == 0), "null argument");
        GSH_HIP(hipSetDevice(b->device));
        GSH_HIP(hipStreamSynchronize(b->stream));
        if (n_samples > b->stream_owned_cap || b->d_stream_owned == nullptr)
            {
                if (b->d_stream_owned) GSH_HIP(hipFree(b->d_stream_owned));
                b->d_stream_owned = nullptr;
                b->stream_owned_cap = 0;
                const size_t cap = std::max<size_t>(static_cast<size_t>(n_samples), 64);  // (an empty stream is a stream: a call over zero samples answers zeros, as the reference's loop does)
                GSH_HIP(hipMalloc(&b->d_stream_owned, sizeof(short2) * cap));
                b->stream_owned_cap = cap;
            }
        if (n_samples > 0) GSH_HIP(hipMemcpy(b->d_stream_owned, iq, sizeof(short2) * static_cast<size_t>(n_samples), hipMemcpyHostToDevice));
        b->d_stream = b->d_stream_owned;
        b->stream_len = n_samples;
        return GSH_OK;
    }

    int gsh_bank16_set_stream_device(gsh_bank16_t* b, const void* device_iq, uint64_t n_samples)
    {
        GSH_REQUIRE(b != nullptr && device_iq != nullptr, "null argument");
        GSH_REQUIRE((reinterpret_cast<uintptr_t>(device_iq) & 3u) == 0, "the device stream must be 4-byte aligned");
        b->d_stream = static_cast<const short2*>(device_iq);
        b->stream_len = n_samples;
        return GSH_OK;
    }

    int gsh_bank16_upload_jobs(gsh_bank16_t* b, const gsh_corr16_job* jobs, int n_jobs)
    {
        GSH_REQUIRE(b != nullptr && (jobs != nullptr || n_jobs == 0), "null argument");
        GSH_REQUIRE(n_jobs >= 0 && n_jobs <= (1 << 22), "n_jobs %d outside 0..%d", n_jobs, 1 << 22);
        GSH_REQUIRE(b->d_stream != nullptr || n_jobs == 0, "no sample stream attached");
        b->n_jobs = 0;
        int max_n = 0, max_taps = 0;
        b->h_jobs.resize(static_cast<size_t>(n_jobs));
        for (int i = 0; i < n_jobs; i++)
            {
                const gsh_corr16_job& j = jobs[i];
                GSH_REQUIRE(j.n_samples >= 0 && j.n_samples <= (1 << 28), "job %d: n_samples %d", i, j.n_samples);
                GSH_REQUIRE(j.sample_offset <= b->stream_len && static_cast<uint64_t>(j.n_samples) <= b->stream_len - j.sample_offset,
                    "job %d: window [%llu, +%d) leaves the stream (%llu samples)", i, static_cast<unsigned long long>(j.sample_offset), j.n_samples, b->stream_len);
                GSH_REQUIRE(j.code_slot >= 0 && j.code_slot < b->n_slots && b->h_code_lens[static_cast<size_t>(j.code_slot)] > 0, "job %d: code slot %d is not set", i, j.code_slot);
                GSH_REQUIRE(j.n_taps >= 1 && j.n_taps <= GSH_MAX_TAPS, "job %d: n_taps %d outside 1..%d", i, j.n_taps, GSH_MAX_TAPS);
                bool finite = std::isfinite(j.rem_carr_phase_rad) && std::isfinite(j.phase_step_rad) && std::isfinite(j.rem_code_phase_chips) && std::isfinite(j.code_phase_step_chips);
                for (int t = 0; t < j.n_taps; t++) finite = finite && std::isfinite(j.shifts_chips[t]);
                GSH_REQUIRE(finite, "job %d: a parameter is not finite", i);
                // every chip index must be an int (the reference's cast is undefined otherwise): |step * n + shift - rem| < 2^30
                {
                    double shift_max = 0.0;
                    for (int t = 0; t < j.n_taps; t++) shift_max = std::max(shift_max, std::fabs(static_cast<double>(j.shifts_chips[t])));
                    const double worst = std::fabs(static_cast<double>(j.code_phase_step_chips)) * static_cast<double>(j.n_samples) + std::fabs(static_cast<double>(j.rem_code_phase_chips)) + shift_max;
                    GSH_REQUIRE(worst < 1073741824.0, "job %d: chip indices beyond 2^30", i);
                }
                gsh::Job16Dev& d = b->h_jobs[static_cast<size_t>(i)];
                d.sample_offset = j.sample_offset;
                d.n_samples = j.n_samples;
                d.code_slot = j.code_slot;
                // cpu_multicorrelator_16sc.cc:89-93, the same expressions on the same C library
                d.ph_re = std::cos(j.rem_carr_phase_rad);
                d.ph_im = -std::sin(j.rem_carr_phase_rad);
                const std::complex<float> inc = std::exp(std::complex<float>(0.0f, -j.phase_step_rad));
                d.inc_re = inc.real();
                d.inc_im = inc.imag();
                d.rem_code = j.rem_code_phase_chips;
                d.code_step = j.code_phase_step_chips;
                d.n_taps = j.n_taps;
                d.pad = 0;
                for (int t = 0; t < GSH_MAX_TAPS; t++) d.shifts[t] = t < j.n_taps ? j.shifts_chips[t] : 0.0f;
                max_n = std::max(max_n, j.n_samples);
                max_taps = std::max(max_taps, j.n_taps);
            }
        if (n_jobs == 0) return GSH_OK;
        GSH_HIP(hipSetDevice(b->device));
        GSH_HIP(hipStreamSynchronize(b->stream));
        if (n_jobs > b->jobs_cap)
            {
                if (b->d_jobs) GSH_HIP(hipFree(b->d_jobs));
                if (b->d_out) GSH_HIP(hipFree(b->d_out));
                b->d_jobs = nullptr;
                b->d_out = nullptr;
                b->jobs_cap = 0;
                GSH_HIP(hipMalloc(&b->d_jobs, sizeof(gsh::Job16Dev) * static_cast<size_t>(n_jobs)));
                GSH_HIP(hipMalloc(&b->d_out, sizeof(short2) * GSH_MAX_TAPS * static_cast<size_t>(n_jobs)));
                b->jobs_cap = n_jobs;
            }
        const unsigned long long stride = (static_cast<unsigned long long>(std::max(max_n, 1)) + 63ull) & ~63ull;
        const size_t need = static_cast<size_t>(stride) * static_cast<size_t>(n_jobs);
        if (need > b->rot_cap)
            {
                if (b->d_rot) GSH_HIP(hipFree(b->d_rot));
                b->d_rot = nullptr;
                b->rot_cap = 0;
                GSH_HIP(hipMalloc(&b->d_rot, sizeof(short2) * need));
                b->rot_cap = need;
            }
        b->rot_stride = stride;
        GSH_HIP(hipMemcpyAsync(b->d_jobs, b->h_jobs.data(), sizeof(gsh::Job16Dev) * static_cast<size_t>(n_jobs), hipMemcpyHostToDevice, b->stream));
        GSH_HIP(hipStreamSynchronize(b->stream));
        b->n_jobs = n_jobs;
        b->max_taps = max_taps;
        return GSH_OK;
    }

    int gsh_bank16_launch(gsh_bank16_t* b)
    {
        GSH_REQUIRE(b != nullptr, "null handle");
        GSH_HIP(hipSetDevice(b->device));
        return bank16_launch(b);
    }

    int gsh_bank16_read_outputs(gsh_bank16_t* b, int16_t* out_iq, int n_jobs)
    {
        GSH_REQUIRE(b != nullptr && (out_iq != nullptr || n_jobs == 0), "null argument");
        GSH_REQUIRE(n_jobs >= 0 && n_jobs <= b->n_jobs, "n_jobs %d outside 0..%d", n_jobs, b->n_jobs);
        GSH_HIP(hipSetDevice(b->device));
        GSH_HIP(hipStreamSynchronize(b->stream));
        if (n_jobs > 0) GSH_HIP(hipMemcpy(out_iq, b->d_out, sizeof(short2) * GSH_MAX_TAPS * static_cast<size_t>(n_jobs), hipMemcpyDeviceToHost));
        return GSH_OK;
    }

    int gsh_bank16_correlate(gsh_bank16_t* b, const gsh_corr16_job* jobs, int n_jobs, int16_t* out_iq)
    {
        int rc = gsh_bank16_upload_jobs(b, jobs, n_jobs);
        if (rc != GSH_OK) return rc;
        rc = gsh_bank16_launch(b);
        if (rc != GSH_OK) return rc;
        return gsh_bank16_read_outputs(b, out_iq, n_jobs);
    }

    int gsh_bank16_time_launches(gsh_bank16_t* b, int reps, float* avg_ms)
    {
        GSH_REQUIRE(b != nullptr && avg_ms != nullptr, "null argument");
        GSH_REQUIRE(reps >= 1 && reps <= 100000, "reps %d outside 1..100000", reps);
        GSH_REQUIRE(b->n_jobs > 0, "no job table uploaded");
        GSH_HIP(hipSetDevice(b->device));
        int rc = bank16_launch(b);  // warm-up
        if (rc != GSH_OK) return rc;
        GSH_HIP(hipEventRecord(b->ev0, b->stream));
        for (int r = 0; r < reps; r++)
            {
                rc = bank16_launch(b);
                if (rc != GSH_OK) return rc;
            }
        GSH_HIP(hipEventRecord(b->ev1, b->stream));
        GSH_HIP(hipEventSynchronize(b->ev1));
        float ms = 0.0f;
        GSH_HIP(hipEventElapsedTime(&ms, b->ev0, b->ev1));
        *avg_ms = ms / static_cast<float>(reps);
        return GSH_OK;
    }
}

// ---- gsh_mcorr16_*: the reference class's surface, one call = one job ------------------------------------------------------------------------------------
struct gsh_mcorr16
{
    int device{0};
    gsh_bank16* bank{nullptr};
    int max_len{0};
    int n_correlators{0};
    int code_len{0};
    float* shifts{nullptr};     // borrowed (cpu_multicorrelator_16sc.cc:49)
    int16_t* corr_out{nullptr};        // borrowed (:60)
    const int16_t* sig_in{nullptr};    // borrowed (:59)
    std::vector<int16_t> code_copy;
};

extern "C"
{
    int gsh_mcorr16_create(int device, gsh_mcorr16_t** out)
    {
        GSH_REQUIRE(out != nullptr, "null out pointer");
        *out = nullptr;
        const int rc = gsh::use_device(device);
        if (rc != GSH_OK) return rc;
        gsh_mcorr16* h = new (std::nothrow) gsh_mcorr16();
        if (h == nullptr) return set_error(GSH_ERR_HIP, "out of host memory");
        h->device = device;
        *out = h;
        return GSH_OK;
    }

    void gsh_mcorr16_destroy(gsh_mcorr16_t* h)
    {
        if (h == nullptr) return;
        if (h->bank) gsh_bank16_destroy(h->bank);
        delete h;
    }

    int gsh_mcorr16_init(gsh_mcorr16_t* h, int max_signal_length_samples, int n_correlators)
    {
        GSH_REQUIRE(h != nullptr, "null handle");
        GSH_REQUIRE(max_signal_length_samples >= 1 && max_signal_length_samples <= (1 << 28), "max_signal_length_samples %d", max_signal_length_samples);
        GSH_REQUIRE(n_correlators >= 1 && n_correlators <= GSH_MAX_TAPS, "n_correlators %d outside 1..%d", n_correlators, GSH_MAX_TAPS);
        h->max_len = max_signal_length_samples;
        h->n_correlators = n_correlators;
        return GSH_OK;
    }

    int gsh_mcorr16_set_local_code_and_taps(gsh_mcorr16_t* h, int code_length_chips, const int16_t* local_code_in_iq, float* shifts_chips)
    {
        GSH_REQUIRE(h != nullptr && local_code_in_iq != nullptr && shifts_chips != nullptr, "null argument");
        GSH_REQUIRE(code_length_chips >= 1 && code_length_chips <= 30000, "code_length_chips %d outside 1..30000", code_length_chips);
        if (h->bank == nullptr || code_length_chips > h->code_len)
            {
                if (h->bank) gsh_bank16_destroy(h->bank);
                h->bank = nullptr;
                const int rc = gsh_bank16_create(h->device, 1, code_length_chips, &h->bank);
                if (rc != GSH_OK) return rc;
            }
        const int rc = gsh_bank16_set_code(h->bank, 0, local_code_in_iq, code_length_chips);
        if (rc != GSH_OK) return rc;
        h->code_len = code_length_chips;
        h->shifts = shifts_chips;
        return GSH_OK;
    }

    int gsh_mcorr16_set_input_output_vectors(gsh_mcorr16_t* h, int16_t* corr_out_iq, const int16_t* sig_in_iq)
    {
        GSH_REQUIRE(h != nullptr && corr_out_iq != nullptr && sig_in_iq != nullptr, "null argument");
        h->corr_out = corr_out_iq;
        h->sig_in = sig_in_iq;
        return GSH_OK;
    }

    int gsh_mcorr16_carrier_wipeoff_multicorrelator_resampler(gsh_mcorr16_t* h, float rem_carrier_phase_in_rad, float phase_step_rad, float rem_code_phase_chips,
        float code_phase_step_chips, int signal_length_samples)
    {
        GSH_REQUIRE(h != nullptr, "null handle");
        if (h->n_correlators <= 0) return set_error(GSH_ERR_STATE, "init() has not been called");
        if (h->bank == nullptr || h->shifts == nullptr) return set_error(GSH_ERR_STATE, "set_local_code_and_taps() has not been called");
        if (h->corr_out == nullptr || h->sig_in == nullptr) return set_error(GSH_ERR_STATE, "set_input_output_vectors() has not been called");
        GSH_REQUIRE(signal_length_samples >= 0 && signal_length_samples <= h->max_len, "signal_length_samples %d outside what init() sized (%d)", signal_length_samples, h->max_len);
        int rc = gsh_bank16_set_stream_host(h->bank, h->sig_in, static_cast<uint64_t>(signal_length_samples));
        if (rc != GSH_OK) return rc;
        gsh_corr16_job j{};
        j.sample_offset = 0;
        j.n_samples = signal_length_samples;
        j.code_slot = 0;
        j.rem_carr_phase_rad = rem_carrier_phase_in_rad;
        j.phase_step_rad = phase_step_rad;
        j.rem_code_phase_chips = rem_code_phase_chips;
        j.code_phase_step_chips = code_phase_step_chips;
        j.n_taps = h->n_correlators;
        for (int t = 0; t < h->n_correlators; t++) j.shifts_chips[t] = h->shifts[t];  // re-read at every call: the caller may have moved the taps
        int16_t out[2 * GSH_MAX_TAPS];
        rc = gsh_bank16_correlate(h->bank, &j, 1, out);
        if (rc != GSH_OK) return rc;
        for (int t = 0; t < 2 * h->n_correlators; t++) h->corr_out[t] = out[t];
        return GSH_OK;
    }

    int gsh_mcorr16_free(gsh_mcorr16_t* h)
    {
        GSH_REQUIRE(h != nullptr, "null handle");
        if (h->bank) gsh_bank16_destroy(h->bank);
        h->bank = nullptr;
        h->code_len = 0;
        h->n_correlators = 0;
        return GSH_OK;
    }
}
