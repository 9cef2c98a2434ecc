// Notch filters on the device: the arithmetic of gnss-sdr's Notch (src/algorithms/input_filter/gnuradio_blocks/notch_cc.cc:33-140, "notch.cc")
// and NotchLite (.../notch_lite_cc.cc:30-150, "lite.cc") input filters -- continuous-wave interference excision in front of the channels.
//
// The stream is cut into segments of `length` samples.  Per segment the blocks keep a small sequential state: while the noise floor is being
// estimated (fewer than n_segments_est segments since the last reset, filter not engaged) the segment passes and its spectral noise floor -- FFT,
// power spectrum in dB, mean of the bins within 15 dB of the mean -- updates a running estimate (notch.cc:74-84); afterwards a segment whose energy
// over that estimate exceeds the chi-squared threshold is filtered by  out[n] = in[n] - z0 in[n-1] + p z0 out[n-1]  (notch.cc:97-102), where Notch
// takes z0 = exp(j arg(in[n] conj(in[n-1]))) per sample and NotchLite one z0 per n_segments_coeff filtered segments from the phase steps at the
// segment's two ends (lite.cc:104-112); out[n-1] starts from 0 whenever the filter engages.
//
// Launches per call:
//   notch_segment_kernel   per segment: energy (float |x|^2 terms summed in double) and -- only when this call can reach an estimating segment --
//                          the floor value: length-point DFT (double accumulation), 10 log10 |X|^2, the reference's two sequential float passes;
//   notch_decide_kernel    one thread: the blocks' state machine over the segments -> per segment mode / engage flag / (lite) z0;
//   notch_compose_kernel   the one-pole recurrence is the affine map out[n] = a[n] + b[n] out[n-1] with b = 0 outside filtered segments and where the
//   notch_carry_kernel     filter engages: each thread composes the maps of its chunk, one thread chains the chunk composites (the state's last
//   notch_apply_kernel     output is the first carry), each thread replays its chunk from its carry and writes the outputs.
// Floating-point parity with the sequential reference: the per-sample expression is formed in the reference's order; chaining chunk composites
// re-associates the recurrence (differences ~1e-7 relative, amplified by 1 / (1 - p)).
#include "gsh_internal.h"
#include <algorithm>
#include <cmath>
#include <new>

struct gsh_notch
{
    int device{0};
    hipStream_t stream{nullptr};
    int length{32};
    int n_segments_est{12500};
    int n_segments_reset{5000000};
    int n_segments_coeff{0};  // 0: Notch (coefficient per sample); >= 1: NotchLite
    float thres{0.0f};
    float p_c_factor{0.9f};
    float* d_energy{nullptr};
    float* d_floor{nullptr};          // sig2lin per segment (notch.cc:82)
    unsigned char* d_mode{nullptr};   // 0 estimate + copy, 1 filter, 2 copy; bit 7: the filter engages here (last_out = 0)
    float2* d_z0{nullptr};            // lite: coefficient per segment
    size_t seg_capacity{0};
    float2* d_comp{nullptr};          // per chunk: (B, A) composites, then carries
    size_t chunk_capacity{0};
    struct State
    {
        float noise_pow_est;
        int n_segments;
        int filter_state;
        int n_segments_coeff;
        float2 last_out;
        float2 z0;
    };
    State* d_state{nullptr};
    State h_state{};  // shadow of the state after the last call (decides whether a call can reach an estimating segment)
};

namespace gsh
{
namespace
{
constexpr int NF_THREADS = 256;
constexpr int NF_CHUNK = 128;  // samples one thread chains sequentially

__device__ __forceinline__ float2 cmul_ref(float2 a, float2 b)  // std::complex<float> product, one rounding per operation
{
    return make_float2(__fsub_rn(__fmul_rn(a.x, b.x), __fmul_rn(a.y, b.y)), __fadd_rn(__fmul_rn(a.x, b.y), __fmul_rn(a.y, b.x)));
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(__fadd_rn(a.x, b.x), __fadd_rn(a.y, b.y)); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(__fsub_rn(a.x, b.x), __fsub_rn(a.y, b.y)); }

// z0 = exp(j atan2(Im c, Re c)), c = cur conj(prev)   (notch.cc:95-99; volk_32fc_x2_multiply_conjugate_32fc, volk_32fc_s32f_atan2_32f, std::exp)
__device__ __forceinline__ float2 phase_step(float2 cur, float2 prev)
{
    const float cr = __fadd_rn(__fmul_rn(cur.x, prev.x), __fmul_rn(cur.y, prev.y));
    const float ci = __fsub_rn(__fmul_rn(cur.y, prev.x), __fmul_rn(cur.x, prev.y));
    const float ang = atan2f(ci, cr);
    float s, c;
    sincosf(ang, &s, &c);
    return make_float2(c, s);
}
__device__ __forceinline__ float angle_step(float2 cur, float2 prev)
{
    const float cr = __fadd_rn(__fmul_rn(cur.x, prev.x), __fmul_rn(cur.y, prev.y));
    const float ci = __fsub_rn(__fmul_rn(cur.y, prev.x), __fmul_rn(cur.x, prev.y));
    return atan2f(ci, cr);
}

// x points at item 0 (the sample in front of the first one processed); segment s covers items 1 + s L .. s L + L
__global__ __launch_bounds__(NF_THREADS) void notch_segment_kernel(const float2* __restrict__ x, int length, unsigned long long n_seg, float* __restrict__ energy,
    float* __restrict__ floor_lin, int with_floor, int seg_per_wg)
{
    extern __shared__ float lds[];
    float2* tw = reinterpret_cast<float2*>(lds);                        // length twiddles
    float2* xs = tw + length;                                           // seg_per_wg * length samples
    float* db = reinterpret_cast<float*>(xs + static_cast<size_t>(seg_per_wg) * length);  // seg_per_wg * length dB values
    const unsigned long long seg0 = static_cast<unsigned long long>(blockIdx.x) * seg_per_wg;
    const int segs = static_cast<int>(min(static_cast<unsigned long long>(seg_per_wg), n_seg - seg0));
    const int count = segs * length;
    const float2* __restrict__ cur = x + 1 + seg0 * length;
    for (int i = threadIdx.x; i < count; i += NF_THREADS) xs[i] = cur[i];
    if (with_floor)
        for (int i = threadIdx.x; i < length; i += NF_THREADS)
            {
                float s, c;
                sincospif(2.0f * static_cast<float>(i) / static_cast<float>(length), &s, &c);
                tw[i] = make_float2(c, -s);
            }
    __syncthreads();
    // energy: real part of volk_32fc_x2_conjugate_dot_prod_32fc(in, in) (notch.cc:87-88), float terms, double sum
    for (int s = threadIdx.x; s < segs; s += NF_THREADS)
        {
            double e = 0.0;
            for (int k = 0; k < length; k++)
                {
                    const float2 v = xs[s * length + (k + s) % length];
                    e += static_cast<double>(__fadd_rn(__fmul_rn(v.x, v.x), __fmul_rn(v.y, v.y)));
                }
            energy[seg0 + s] = static_cast<float>(e);
        }
    if (!with_floor) return;
    // spectrum: bin k of segment s (notch.cc:77-79); 10 log10 through a base-2 logarithm like the VOLK kernel
    for (int i = threadIdx.x; i < count; i += NF_THREADS)
        {
            const int s = i / length, k = i - s * length;
            double ar = 0.0, ai = 0.0;
            int m = 0;
            for (int n = 0; n < length; n++)
                {
                    const float2 v = xs[s * length + n], w = tw[m];
                    ar += static_cast<double>(v.x) * w.x - static_cast<double>(v.y) * w.y;
                    ai += static_cast<double>(v.x) * w.y + static_cast<double>(v.y) * w.x;
                    m += k;
                    if (m >= length) m -= length;
                }
            const float re = static_cast<float>(ar), im = static_cast<float>(ai);
            const float p = __fadd_rn(__fmul_rn(re, re), __fmul_rn(im, im));
            db[i] = p > 0.0f ? __fmul_rn(3.01029995663981209120f, log2f(p)) : __fmul_rn(3.01029995663981209120f, -127.0f);
        }
    __syncthreads();
    // volk_32f_s32f_calc_spectral_noise_floor_32f(&sig2dB, power_spect, 15.0, length) and notch.cc:82, sequential float sums as the kernel forms them
    for (int s = threadIdx.x; s < segs; s += NF_THREADS)
        {
            const float* d = db + s * length;
            float sum = 0.0f;
            for (int k = 0; k < length; k++) sum = __fadd_rn(sum, d[k]);
            const float mean_amp = __fadd_rn(__fdiv_rn(sum, static_cast<float>(length)), 15.0f);
            sum = 0.0f;
            int kept = length;
            for (int k = 0; k < length; k++)
                {
                    if (d[k] <= mean_amp)
                        sum = __fadd_rn(sum, d[k]);
                    else
                        kept--;
                }
            const float sig2db = kept == 0 ? mean_amp : __fdiv_rn(sum, static_cast<float>(kept));
            floor_lin[seg0 + s] = __fdiv_rn(powf(10.0f, __fdiv_rn(sig2db, 10.0f)), static_cast<float>(2 * length));
        }
}

// notch.cc:72-118 / lite.cc:81-136, one segment after the other
__global__ void notch_decide_kernel(const float2* __restrict__ x, const float* __restrict__ energy, const float* __restrict__ floor_lin, int have_floor,
    unsigned long long n_seg, int length, unsigned char* __restrict__ mode, float2* __restrict__ z0_seg, gsh_notch::State* st, int n_segments_est, int n_segments_reset,
    int n_segments_coeff_reset, float thres, int* need_floor)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    gsh_notch::State s = *st;
    const bool lite = n_segments_coeff_reset > 0;
    for (unsigned long long i = 0; i < n_seg; i++)
        {
            unsigned char m;
            if ((s.n_segments < n_segments_est) && (s.filter_state == 0))
                {
                    if (!have_floor)
                        {
                            *need_floor = 1;  // the host sized this call without the floor values: it repeats the call with them (nothing was written yet)
                            return;
                        }
                    s.noise_pow_est = __fdiv_rn(__fadd_rn(__fmul_rn(static_cast<float>(s.n_segments), s.noise_pow_est), floor_lin[i]), static_cast<float>(s.n_segments + 1));
                    m = 0;
                }
            else
                {
                    if (__fdiv_rn(energy[i], s.noise_pow_est) > thres)
                        {
                            m = 1;
                            if (s.filter_state == 0)
                                {
                                    s.filter_state = 1;
                                    m |= 0x80;  // last_out_ = 0
                                    s.n_segments_coeff = 0;
                                }
                            if (lite)
                                {
                                    if (s.n_segments_coeff == 0)
                                        {
                                            const float2* cur = x + 1 + i * length;
                                            const float a1 = angle_step(cur[1], cur[0]);
                                            const float a2 = angle_step(cur[length - 1], cur[length - 2]);
                                            const float ang = __fdiv_rn(__fadd_rn(a1, a2), 2.0f);
                                            float sn, cs;
                                            sincosf(ang, &sn, &cs);
                                            s.z0 = make_float2(cs, sn);
                                        }
                                    z0_seg[i] = s.z0;
                                    s.n_segments_coeff = (s.n_segments_coeff + 1) % n_segments_coeff_reset;
                                }
                        }
                    else
                        {
                            if (s.n_segments > n_segments_reset) s.n_segments = 0;
                            s.filter_state = 0;
                            m = 2;
                        }
                }
            mode[i] = m;
            s.n_segments++;
        }
    *st = s;  // last_out / (Notch) z0 are completed by the apply kernel
}

// the affine map of sample n (output index): out = a + b * out_prev
__device__ __forceinline__ void sample_map(const float2* __restrict__ x, unsigned long long n, int length, const unsigned char* __restrict__ mode,
    const float2* __restrict__ z0_seg, bool lite, float p, float2& a, float2& b, float2* z_out = nullptr)
{
    const unsigned long long s = n / length;
    const unsigned char m = mode[s];
    const float2 cur = x[n + 1], prev = x[n];
    if ((m & 0x7f) != 1)
        {
            a = cur;
            b = make_float2(0.0f, 0.0f);
            return;
        }
    const float2 z = lite ? z0_seg[s] : phase_step(cur, prev);
    if (z_out) *z_out = z;
    a = csub(cur, cmul_ref(z, prev));                                 // in[n] - z0 in[n-1]
    b = cmul_ref(make_float2(p, 0.0f), z);                            // p_c_factor_ * z_0_ (complex product as the block forms it)
    if ((m & 0x80) && (n - s * length == 0)) b = make_float2(0.0f, 0.0f);  // the filter engages: last_out_ = 0
}

__global__ __launch_bounds__(NF_THREADS) void notch_compose_kernel(const float2* __restrict__ x, unsigned long long n_total, int length,
    const unsigned char* __restrict__ mode, const float2* __restrict__ z0_seg, int lite, float p, float2* __restrict__ comp)
{
    const unsigned long long c = static_cast<unsigned long long>(blockIdx.x) * NF_THREADS + threadIdx.x;
    const unsigned long long n0 = c * NF_CHUNK;
    if (n0 >= n_total) return;
    const unsigned long long n1 = min(n_total, n0 + NF_CHUNK);
    float2 B = make_float2(1.0f, 0.0f), A = make_float2(0.0f, 0.0f);
    for (unsigned long long n = n0; n < n1; n++)
        {
            float2 a, b;
            sample_map(x, n, length, mode, z0_seg, lite != 0, p, a, b);
            A = cadd(a, cmul_ref(b, A));
            B = cmul_ref(b, B);
        }
    comp[2 * c] = B;
    comp[2 * c + 1] = A;
}

// carries: carry[c] = value of out just before chunk c
__global__ void notch_carry_kernel(float2* __restrict__ comp, unsigned long long n_chunks, const gsh_notch::State* st)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float2 carry = st->last_out;
    for (unsigned long long c = 0; c < n_chunks; c++)
        {
            const float2 B = comp[2 * c], A = comp[2 * c + 1];
            comp[2 * c] = carry;
            carry = cadd(A, cmul_ref(B, carry));
        }
}

__global__ __launch_bounds__(NF_THREADS) void notch_apply_kernel(const float2* x, float2* y, unsigned long long n_total, int length,  // y may not alias x (x[n] is read after y[n - 1] is written)
    const unsigned char* __restrict__ mode, const float2* __restrict__ z0_seg, int lite, float p, const float2* __restrict__ comp, gsh_notch::State* st)
{
    const unsigned long long c = static_cast<unsigned long long>(blockIdx.x) * NF_THREADS + threadIdx.x;
    const unsigned long long n0 = c * NF_CHUNK;
    if (n0 >= n_total) return;
    const unsigned long long n1 = min(n_total, n0 + NF_CHUNK);
    float2 out = comp[2 * c];
    float2 z_last = make_float2(0.0f, 0.0f);
    bool any = false;
    for (unsigned long long n = n0; n < n1; n++)
        {
            float2 a, b, z = make_float2(0.0f, 0.0f);
            sample_map(x, n, length, mode, z0_seg, lite != 0, p, a, b, &z);
            const bool filt = (mode[n / length] & 0x7f) == 1;
            if (filt)
                {
                    out = cadd(a, cmul_ref(b, out));
                    z_last = z;
                    any = true;
                }
            y[n] = filt ? out : a;
            // (a copied segment leaves last_out_ as it was; it is cleared when the filter engages again, so the stale value is never used)
        }
    if (n1 == n_total)
        {
            // state for the next call: the last filtered output of this call, if any, else what it was
            if (any) st->last_out = out;
            if (!lite && any) st->z0 = z_last;
        }
}
}  // namespace
}  // namespace gsh

extern "C"
{
    int gsh_notch_create(int device, float pfa, float p_c_factor, int32_t length, int32_t n_segments_est, int32_t n_segments_reset, int32_t n_segments_coeff,
        gsh_notch_t** out)
    {
        GSH_REQUIRE(out != nullptr, "null argument");
        *out = nullptr;
        GSH_REQUIRE(pfa > 0.0f && pfa < 1.0f, "pfa %g outside (0, 1)", static_cast<double>(pfa));
        GSH_REQUIRE(length >= 2 && length <= 1024, "length %d outside 2..1024", length);
        GSH_REQUIRE(n_segments_est >= 0 && n_segments_reset >= 0 && n_segments_coeff >= 0, "negative segment count");
        GSH_REQUIRE(std::isfinite(p_c_factor), "p_c_factor is not finite");
        int rc = gsh::use_device(device);
        if (rc != GSH_OK) return rc;
        gsh_notch* p = new (std::nothrow) gsh_notch();
        GSH_REQUIRE(p != nullptr, "out of host memory");
        p->device = device;
        p->length = length;
        p->n_segments_est = n_segments_est;
        p->n_segments_reset = n_segments_reset;
        p->n_segments_coeff = n_segments_coeff;
        p->p_c_factor = p_c_factor;
        // notch.cc:54-55: thres_ = quantile(complement(chi_squared(2 * length), pfa)) = 2 * gamma_p_inv(length, 1 - pfa), in float
        p->thres = static_cast<float>(2.0 * gsh::gamma_p_inv(static_cast<double>(length), 1.0 - static_cast<double>(pfa)));
        auto fail = [&](hipError_t e, const char* what) {
            gsh::hip_fail(e, what, __FILE__, __LINE__);
            gsh_notch_destroy(p);
            return GSH_ERR_HIP;
        };
        hipError_t e;
        if ((e = hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking)) != hipSuccess) return fail(e, "hipStreamCreate");
        if ((e = hipMalloc(&p->d_state, sizeof(gsh_notch::State) + sizeof(int))) != hipSuccess) return fail(e, "hipMalloc(state)");
        if ((e = hipMemset(p->d_state, 0, sizeof(gsh_notch::State) + sizeof(int))) != hipSuccess) return fail(e, "hipMemset(state)");
        *out = p;
        return GSH_OK;
    }

    void gsh_notch_destroy(gsh_notch_t* p)
    {
        if (!p) return;
        (void)hipSetDevice(p->device);
        if (p->stream) (void)hipStreamSynchronize(p->stream);
        if (p->d_energy) (void)hipFree(p->d_energy);
        if (p->d_floor) (void)hipFree(p->d_floor);
        if (p->d_mode) (void)hipFree(p->d_mode);
        if (p->d_z0) (void)hipFree(p->d_z0);
        if (p->d_comp) (void)hipFree(p->d_comp);
        if (p->d_state) (void)hipFree(p->d_state);
        if (p->stream) (void)hipStreamDestroy(p->stream);
        delete p;
    }

    float gsh_notch_threshold(const gsh_notch_t* p) { return p ? p->thres : 0.0f; }

    int gsh_notch_process_device(gsh_notch_t* p, const void* device_in_iq, uint64_t n_items, void* device_out_iq, uint64_t* n_done)
    {
        GSH_REQUIRE(p != nullptr && n_done != nullptr, "null argument");
        *n_done = 0;
        if (n_items == 0) return GSH_OK;
        GSH_REQUIRE(device_in_iq != nullptr && device_out_iq != nullptr && device_in_iq != device_out_iq, "null buffer, or in place (the filter reads in[n - 1] after out[n - 1] is written)");
        GSH_HIP(hipSetDevice(p->device));
        // notch.cc:71-72: in++; while ((index_out + length_) < noutput_items)
        const uint64_t L = static_cast<uint64_t>(p->length);
        const uint64_t n_seg = (n_items > L) ? (n_items - 1) / L : 0;
        if (n_seg == 0) return GSH_OK;
        if (p->seg_capacity < n_seg)
            {
                for (void* q : {static_cast<void*>(p->d_energy), static_cast<void*>(p->d_floor), static_cast<void*>(p->d_mode), static_cast<void*>(p->d_z0)})
                    if (q) (void)hipFree(q);
                p->d_energy = p->d_floor = nullptr;
                p->d_mode = nullptr;
                p->d_z0 = nullptr;
                p->seg_capacity = 0;
                GSH_HIP(hipMalloc(&p->d_energy, sizeof(float) * n_seg));
                GSH_HIP(hipMalloc(&p->d_floor, sizeof(float) * n_seg));
                GSH_HIP(hipMalloc(&p->d_mode, n_seg));
                GSH_HIP(hipMalloc(&p->d_z0, sizeof(float2) * n_seg));
                p->seg_capacity = n_seg;
            }
        const uint64_t n = n_seg * L;
        const uint64_t n_chunks = (n + gsh::NF_CHUNK - 1) / gsh::NF_CHUNK;
        if (p->chunk_capacity < n_chunks)
            {
                if (p->d_comp) (void)hipFree(p->d_comp);
                p->d_comp = nullptr;
                p->chunk_capacity = 0;
                GSH_HIP(hipMalloc(&p->d_comp, sizeof(float2) * 2 * n_chunks));
                p->chunk_capacity = n_chunks;
            }
        const float2* x = static_cast<const float2*>(device_in_iq);
        int* d_need = reinterpret_cast<int*>(p->d_state + 1);
        // the floor values cost a length-point DFT per segment: formed only when this call can reach an estimating segment (the state machine asks for a
        // repeat with them in the one case the shadow cannot foresee: the filter disengaging inside the call while the counter is still below n_segments_est)
        const gsh_notch::State& h = p->h_state;
        bool with_floor = (h.n_segments < p->n_segments_est) || (static_cast<uint64_t>(h.n_segments) + n_seg > static_cast<uint64_t>(p->n_segments_reset));
        const int seg_per_wg = std::max(1, 2048 / p->length);
        const unsigned blocks_s = static_cast<unsigned>((n_seg + seg_per_wg - 1) / seg_per_wg);
        const size_t lds = sizeof(float2) * p->length + (sizeof(float2) + sizeof(float)) * static_cast<size_t>(seg_per_wg) * p->length;
        for (int attempt = 0; attempt < 2; attempt++)
            {
                GSH_HIP(hipMemsetAsync(d_need, 0, sizeof(int), p->stream));
                hipLaunchKernelGGL(gsh::notch_segment_kernel, dim3(blocks_s), dim3(gsh::NF_THREADS), lds, p->stream, x, p->length, static_cast<unsigned long long>(n_seg),
                    p->d_energy, p->d_floor, with_floor ? 1 : 0, seg_per_wg);
                GSH_HIP(hipGetLastError());
                hipLaunchKernelGGL(gsh::notch_decide_kernel, dim3(1), dim3(64), 0, p->stream, x, p->d_energy, p->d_floor, with_floor ? 1 : 0,
                    static_cast<unsigned long long>(n_seg), p->length, p->d_mode, p->d_z0, p->d_state, p->n_segments_est, p->n_segments_reset, p->n_segments_coeff, p->thres, d_need);
                GSH_HIP(hipGetLastError());
                if (with_floor) break;
                int need = 0;
                GSH_HIP(hipMemcpyAsync(&need, d_need, sizeof(int), hipMemcpyDeviceToHost, p->stream));
                GSH_HIP(hipStreamSynchronize(p->stream));
                if (!need) break;
                with_floor = true;
            }
        const unsigned blocks_c = static_cast<unsigned>((n_chunks + gsh::NF_THREADS - 1) / gsh::NF_THREADS);
        const int lite = p->n_segments_coeff > 0 ? 1 : 0;
        hipLaunchKernelGGL(gsh::notch_compose_kernel, dim3(blocks_c), dim3(gsh::NF_THREADS), 0, p->stream, x, static_cast<unsigned long long>(n), p->length, p->d_mode,
            p->d_z0, lite, p->p_c_factor, p->d_comp);
        GSH_HIP(hipGetLastError());
        hipLaunchKernelGGL(gsh::notch_carry_kernel, dim3(1), dim3(64), 0, p->stream, p->d_comp, static_cast<unsigned long long>(n_chunks), p->d_state);
        GSH_HIP(hipGetLastError());
        hipLaunchKernelGGL(gsh::notch_apply_kernel, dim3(blocks_c), dim3(gsh::NF_THREADS), 0, p->stream, x, static_cast<float2*>(device_out_iq),
            static_cast<unsigned long long>(n), p->length, p->d_mode, p->d_z0, lite, p->p_c_factor, p->d_comp, p->d_state);
        GSH_HIP(hipGetLastError());
        GSH_HIP(hipMemcpyAsync(&p->h_state, p->d_state, sizeof(gsh_notch::State), hipMemcpyDeviceToHost, p->stream));
        GSH_HIP(hipStreamSynchronize(p->stream));
        *n_done = n;
        return GSH_OK;
    }

    int gsh_notch_get_state(gsh_notch_t* p, float* noise_pow_est, int32_t* n_segments, int32_t* filter_state, float* last_out_iq, int32_t* n_segments_coeff, float* z0_iq)
    {
        GSH_REQUIRE(p != nullptr, "null handle");
        const gsh_notch::State& s = p->h_state;
        if (noise_pow_est) *noise_pow_est = s.noise_pow_est;
        if (n_segments) *n_segments = s.n_segments;
        if (filter_state) *filter_state = s.filter_state;
        if (last_out_iq)
            {
                last_out_iq[0] = s.last_out.x;
                last_out_iq[1] = s.last_out.y;
            }
        if (n_segments_coeff) *n_segments_coeff = s.n_segments_coeff;
        if (z0_iq)
            {
                z0_iq[0] = s.z0.x;
                z0_iq[1] = s.z0.y;
            }
        return GSH_OK;
    }
}
