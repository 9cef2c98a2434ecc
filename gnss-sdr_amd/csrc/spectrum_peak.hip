// Largest line of a zero-padded spectrum: the fine-Doppler step of gnss-sdr's pcps_acquisition_fine_doppler_cc
// (src/algorithms/acquisition/gnuradio_blocks/pcps_acquisition_fine_doppler_cc.cc:316-389, estimate_Doppler): ten code periods of
// signal are multiplied by the aligned code replica (code wipe-off, :348), zero-padded eightfold, transformed (:351), and the bin with
// the largest |X|^2 (:354-358) gives the carrier frequency.
//
// One transform of up to ~2 M points through the four-step plan of pcps_fft.hip (product and zero padding on load, spectrum left in
// the plan's permuted [k1][k2] layout, k = k1 + n1 * k2), then a two-level arg-max over |X|^2 that maps positions back to k and keeps
// the LOWEST k among equal maxima (volk_gnsssdr_32f_index_max_32u scans upward with '>').
#include "pcps_fft.h"
#include <vector>

namespace gsh
{
namespace
{
constexpr int SP_THREADS = 256;

__global__ __launch_bounds__(SP_THREADS) void sp_multiply_kernel(const float2* x, const float2* __restrict__ w, float2* y, unsigned n)  // y may be x
{
    const unsigned i = blockIdx.x * SP_THREADS + threadIdx.x;
    if (i >= n) return;
    const float2 a = x[i], b = w[i];
    // volk_32fc_x2_multiply_32fc: (ar br - ai bi, ar bi + ai br)
    y[i] = make_float2(__fsub_rn(__fmul_rn(a.x, b.x), __fmul_rn(a.y, b.y)), __fadd_rn(__fmul_rn(a.x, b.y), __fmul_rn(a.y, b.x)));
}

struct Cand
{
    float v;
    unsigned k;
};

__device__ __forceinline__ void take(Cand& a, float v, unsigned k)
{
    if (v > a.v || (v == a.v && k < a.k))
        {
            a.v = v;
            a.k = k;
        }
}

// stage 1: every work-group reduces a slice of the permuted spectrum; stage 2 (one work-group) reduces the slices
__global__ __launch_bounds__(SP_THREADS) void sp_argmax_kernel(const float2* __restrict__ spec, unsigned n, unsigned n1, unsigned n2, Cand* __restrict__ part)
{
    __shared__ Cand red[SP_THREADS];
    Cand c{-1.0f, 0xFFFFFFFFu};
    for (unsigned p = blockIdx.x * SP_THREADS + threadIdx.x; p < n; p += gridDim.x * SP_THREADS)
        {
            const float2 v = spec[p];
            const float m = __fadd_rn(__fmul_rn(v.x, v.x), __fmul_rn(v.y, v.y));  // volk_32fc_magnitude_squared_32f
            const unsigned k1 = p / n2, k2 = p - k1 * n2;                         // position p holds X[k1 + n1 * k2]
            take(c, m, k1 + n1 * k2);
        }
    red[threadIdx.x] = c;
    __syncthreads();
    for (int off = SP_THREADS / 2; off > 0; off >>= 1)
        {
            if (threadIdx.x < off) take(red[threadIdx.x], red[threadIdx.x + off].v, red[threadIdx.x + off].k);
            __syncthreads();
        }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(SP_THREADS) void sp_final_kernel(const Cand* __restrict__ part, unsigned n_part, Cand* __restrict__ out)
{
    __shared__ Cand red[SP_THREADS];
    Cand c{-1.0f, 0xFFFFFFFFu};
    for (unsigned i = threadIdx.x; i < n_part; i += SP_THREADS) take(c, part[i].v, part[i].k);
    red[threadIdx.x] = c;
    __syncthreads();
    for (int off = SP_THREADS / 2; off > 0; off >>= 1)
        {
            if (threadIdx.x < off) take(red[threadIdx.x], red[threadIdx.x + off].v, red[threadIdx.x + off].k);
            __syncthreads();
        }
    if (threadIdx.x == 0) *out = red[0];
}
}  // namespace
}  // namespace gsh

extern "C"
{
    int gsh_spectrum_peak(int device, const float* x_iq, const float* w_iq, uint32_t n, uint32_t fft_size, uint32_t* index, float* peak)
    {
        GSH_REQUIRE(x_iq != nullptr && index != nullptr, "null argument");
        GSH_REQUIRE(n >= 1 && n <= fft_size, "n %u outside 1..fft_size %u", n, fft_size);
        GSH_REQUIRE(fft_size >= 4 && fft_size <= (1u << 24), "fft_size %u outside 4..2^24", fft_size);
        int rc = gsh::use_device(device);
        if (rc != GSH_OK) return rc;
        gsh::FftPlan plan;
        rc = gsh::plan_create(static_cast<int>(fft_size), &plan);
        if (rc != GSH_OK) return rc;
        float2 *d_x = nullptr, *d_w = nullptr, *d_tmp = nullptr, *d_spec = nullptr;
        gsh::Cand *d_part = nullptr, *d_out = nullptr;
        hipStream_t s = nullptr;
        const unsigned n_part = 1024;
        auto cleanup = [&]() {
            if (s) (void)hipStreamSynchronize(s);
            if (d_x) (void)hipFree(d_x);
            if (d_w) (void)hipFree(d_w);
            if (d_tmp) (void)hipFree(d_tmp);
            if (d_spec) (void)hipFree(d_spec);
            if (d_part) (void)hipFree(d_part);
            if (d_out) (void)hipFree(d_out);
            if (s) (void)hipStreamDestroy(s);
            gsh::plan_destroy(&plan);
        };
#define SP_HIP(call)                                                           \
    do                                                                         \
        {                                                                      \
            hipError_t e__ = (call);                                           \
            if (e__ != hipSuccess)                                             \
                {                                                              \
                    cleanup();                                                 \
                    return gsh::hip_fail(e__, #call, __FILE__, __LINE__);      \
                }                                                              \
        }                                                                      \
    while (0)
        SP_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        SP_HIP(hipMalloc(&d_x, sizeof(float2) * n));
        SP_HIP(hipMalloc(&d_tmp, sizeof(float2) * fft_size));
        SP_HIP(hipMalloc(&d_spec, sizeof(float2) * fft_size));
        SP_HIP(hipMalloc(&d_part, sizeof(gsh::Cand) * n_part));
        SP_HIP(hipMalloc(&d_out, sizeof(gsh::Cand)));
        SP_HIP(hipMemcpyAsync(d_x, x_iq, sizeof(float2) * n, hipMemcpyHostToDevice, s));
        if (w_iq != nullptr)
            {
                SP_HIP(hipMalloc(&d_w, sizeof(float2) * n));
                SP_HIP(hipMemcpyAsync(d_w, w_iq, sizeof(float2) * n, hipMemcpyHostToDevice, s));
                hipLaunchKernelGGL(gsh::sp_multiply_kernel, dim3((n + gsh::SP_THREADS - 1) / gsh::SP_THREADS), dim3(gsh::SP_THREADS), 0, s, d_x, d_w, d_x, n);
                SP_HIP(hipGetLastError());
            }
        rc = gsh::fft_forward(plan, d_x, 0, static_cast<int>(n), 0, nullptr, 1.0, d_tmp, d_spec, 1, s);  // zero padding on load (:324)
        if (rc != GSH_OK)
            {
                cleanup();
                return rc;
            }
        hipLaunchKernelGGL(gsh::sp_argmax_kernel, dim3(n_part), dim3(gsh::SP_THREADS), 0, s, d_spec, fft_size, static_cast<unsigned>(plan.n1), static_cast<unsigned>(plan.n2), d_part);
        SP_HIP(hipGetLastError());
        hipLaunchKernelGGL(gsh::sp_final_kernel, dim3(1), dim3(gsh::SP_THREADS), 0, s, d_part, n_part, d_out);
        SP_HIP(hipGetLastError());
        gsh::Cand h{};
        SP_HIP(hipMemcpyAsync(&h, d_out, sizeof(h), hipMemcpyDeviceToHost, s));
        SP_HIP(hipStreamSynchronize(s));
#undef SP_HIP
        cleanup();
        *index = h.k;
        if (peak != nullptr) *peak = h.v;
        return GSH_OK;
    }
}
