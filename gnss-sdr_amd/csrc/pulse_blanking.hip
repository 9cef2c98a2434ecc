// Pulse blanking on the device: the arithmetic of gnss-sdr's pulse_blanking_cc
// (src/algorithms/input_filter/gnuradio_blocks/pulse_blanking_cc.cc:33-106), the input filter that zeroes `length`-sample segments
// whose energy exceeds a chi-squared threshold over an estimated noise floor (pulsed interference, e.g. DME in the L5 / E5a band).
//
// The stream is tiled into consecutive segments of `length` samples starting at sample 0 (the block consumes whole segments only and
// gets the remainder back on its next call, :66 and :104, so the tiling does not depend on how the scheduler cuts the stream).  Per
// segment the reference keeps a small sequential state (:69-95): while fewer than n_segments_est segments have been seen since the last
// reset and the previous segment was not blanked, the segment updates a running mean of the noise power and passes; otherwise it is
// blanked when energy / noise > thres and passes when not (resetting the segment counter after n_segments_reset).
//
// Three launches per call: segment energies (data parallel), the state machine over the segments (one thread: it is the reference's
// sequential recurrence, ~20 operations per segment), and the masked copy (data parallel).  Energies are sums of the float |x|^2 the
// reference forms, added in double (its float VOLK accumulator has no defined lane order).  HBM-bound: 8 B read twice + 8 B written per sample.
#include "gsh_internal.h"
#include <cmath>
#include <new>

struct gsh_pulse_blanking
{
    int device{0};
    hipStream_t stream{nullptr};
    int length{32};
    int n_segments_est{12500};
    int n_segments_reset{5000000};
    float thres{0.0f};
    float* d_energy{nullptr};      // per-segment energies of the current call
    unsigned char* d_mask{nullptr};
    size_t seg_capacity{0};
    struct State
    {
        float noise_power_estimation;
        int n_segments;
        int last_filtered;
        int n_deg_fred;
    };
    State* d_state{nullptr};
};

namespace gsh
{
namespace
{
constexpr int PB_THREADS = 256;

// one thread per segment would read with a stride of `length` samples; instead a work-group stages a tile of |x|^2 in LDS with
// coalesced reads and every thread then sums one segment out of LDS
__global__ __launch_bounds__(PB_THREADS) void pb_energy_kernel(const float2* __restrict__ x, int length, unsigned long long n_seg, float* __restrict__ energy,
    int seg_per_wg)
{
    extern __shared__ float tile[];
    const unsigned long long seg0 = static_cast<unsigned long long>(blockIdx.x) * seg_per_wg;
    const unsigned long long segs = min(static_cast<unsigned long long>(seg_per_wg), n_seg - seg0);
    const unsigned long long base = seg0 * length;
    const unsigned long long count = segs * length;
    for (unsigned long long i = threadIdx.x; i < count; i += PB_THREADS)
        {
            const float2 v = x[base + i];
            tile[i] = __fadd_rn(__fmul_rn(v.x, v.x), __fmul_rn(v.y, v.y));  // volk_32fc_magnitude_squared_32f, :63
        }
    __syncthreads();
    for (unsigned long long s = threadIdx.x; s < segs; s += PB_THREADS)
        {
            double e = 0.0;
            const float* t = tile + s * length;
            for (int k = 0; k < length; k++) e += static_cast<double>(t[(k + static_cast<int>(s)) % length]);  // rotated start: spreads the LDS banks
            energy[seg0 + s] = static_cast<float>(e);
        }
}

// pulse_blanking_cc.cc:69-95, one segment after the other
__global__ void pb_decide_kernel(const float* __restrict__ energy, unsigned long long n_seg, unsigned char* __restrict__ mask, gsh_pulse_blanking::State* st,
    int n_segments_est, int n_segments_reset, float thres)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float noise = st->noise_power_estimation;
    int n_segments = st->n_segments;
    bool last_filtered = st->last_filtered != 0;
    const float n_deg = static_cast<float>(st->n_deg_fred);
    for (unsigned long long s = 0; s < n_seg; s++)
        {
            const float segment_energy = energy[s];
            unsigned char blank = 0;
            if ((n_segments < n_segments_est) && (last_filtered == false))
                {
                    noise = __fdiv_rn(__fadd_rn(__fmul_rn(static_cast<float>(n_segments), noise), __fdiv_rn(segment_energy, n_deg)), static_cast<float>(n_segments + 1));
                }
            else
                {
                    if (__fdiv_rn(segment_energy, noise) > thres)
                        {
                            blank = 1;
                            last_filtered = true;
                        }
                    else
                        {
                            last_filtered = false;
                            if (n_segments > n_segments_reset) n_segments = 0;
                        }
                }
            mask[s] = blank;
            n_segments++;
        }
    st->noise_power_estimation = noise;
    st->n_segments = n_segments;
    st->last_filtered = last_filtered ? 1 : 0;
}

__global__ __launch_bounds__(PB_THREADS) void pb_apply_kernel(const float2* x, float2* y, const unsigned char* __restrict__ mask, int length,  // x may alias y (in place)
    unsigned long long n)
{
    const unsigned long long stride = static_cast<unsigned long long>(gridDim.x) * PB_THREADS;
    for (unsigned long long i = static_cast<unsigned long long>(blockIdx.x) * PB_THREADS + threadIdx.x; i < n; i += stride)
        y[i] = mask[i / length] ? make_float2(0.0f, 0.0f) : x[i];
}
}  // namespace
}  // namespace gsh

extern "C"
{
    int gsh_pb_create(int device, float pfa, int32_t length, int32_t n_segments_est, int32_t n_segments_reset, gsh_pb_t** out)
    {
        GSH_REQUIRE(out != nullptr, "null argument");
        *out = nullptr;
        GSH_REQUIRE(pfa > 0.0f && pfa < 1.0f, "pfa %g outside (0, 1)", static_cast<double>(pfa));
        GSH_REQUIRE(length >= 1 && length <= 4096, "length %d outside 1..4096", length);
        GSH_REQUIRE(n_segments_est >= 0 && n_segments_reset >= 0, "negative segment count");
        int rc = gsh::use_device(device);
        if (rc != GSH_OK) return rc;
        gsh_pulse_blanking* p = new (std::nothrow) gsh_pulse_blanking();
        GSH_REQUIRE(p != nullptr, "out of host memory");
        p->device = device;
        p->length = length;
        p->n_segments_est = n_segments_est;
        p->n_segments_reset = n_segments_reset;
        // :48-49: thres_ = quantile(complement(chi_squared(2 * length), pfa)) = 2 * gamma_p_inv(length, 1 - pfa), in float
        p->thres = static_cast<float>(2.0 * gsh::gamma_p_inv(static_cast<double>(length), 1.0 - static_cast<double>(pfa)));
        auto fail = [&](hipError_t e, const char* what) {
            gsh::hip_fail(e, what, __FILE__, __LINE__);
            gsh_pb_destroy(p);
            return GSH_ERR_HIP;
        };
        hipError_t e;
        if ((e = hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking)) != hipSuccess) return fail(e, "hipStreamCreate");
        if ((e = hipMalloc(&p->d_state, sizeof(gsh_pulse_blanking::State))) != hipSuccess) return fail(e, "hipMalloc(state)");
        const gsh_pulse_blanking::State s0{0.0f, 0, 0, 2 * length};  // :38-44
        if ((e = hipMemcpy(p->d_state, &s0, sizeof(s0), hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "hipMemcpy(state)");
        *out = p;
        return GSH_OK;
    }

    void gsh_pb_destroy(gsh_pb_t* p)
    {
        if (!p) return;
        (void)hipSetDevice(p->device);
        if (p->stream) (void)hipStreamSynchronize(p->stream);
        if (p->d_energy) (void)hipFree(p->d_energy);
        if (p->d_mask) (void)hipFree(p->d_mask);
        if (p->d_state) (void)hipFree(p->d_state);
        if (p->stream) (void)hipStreamDestroy(p->stream);
        delete p;
    }

    float gsh_pb_threshold(const gsh_pb_t* p) { return p ? p->thres : 0.0f; }

    int gsh_pb_process_device(gsh_pb_t* p, const void* device_in_iq, uint64_t n_items, void* device_out_iq, uint64_t* n_done)
    {
        GSH_REQUIRE(p != nullptr && n_done != nullptr, "null argument");
        *n_done = 0;
        if (n_items == 0) return GSH_OK;
        GSH_REQUIRE(device_in_iq != nullptr && device_out_iq != nullptr, "null buffer");
        GSH_HIP(hipSetDevice(p->device));
        // :66: while ((sample_index + length_) < noutput_items)  -- a segment that ends exactly on the last item waits for the next call
        const uint64_t L = static_cast<uint64_t>(p->length);
        const uint64_t n_seg = (n_items > L) ? (n_items - 1) / L : 0;
        if (n_seg == 0) return GSH_OK;
        if (p->seg_capacity < n_seg)
            {
                if (p->d_energy) (void)hipFree(p->d_energy);
                if (p->d_mask) (void)hipFree(p->d_mask);
                p->d_energy = nullptr;
                p->d_mask = nullptr;
                p->seg_capacity = 0;
                GSH_HIP(hipMalloc(&p->d_energy, sizeof(float) * n_seg));
                GSH_HIP(hipMalloc(&p->d_mask, n_seg));
                p->seg_capacity = n_seg;
            }
        const int seg_per_wg = std::max(1, 8192 / p->length);  // 32 KB of LDS per work-group
        const unsigned blocks_e = static_cast<unsigned>((n_seg + seg_per_wg - 1) / seg_per_wg);
        hipLaunchKernelGGL(gsh::pb_energy_kernel, dim3(blocks_e), dim3(gsh::PB_THREADS), sizeof(float) * static_cast<size_t>(seg_per_wg) * p->length, p->stream,
            static_cast<const float2*>(device_in_iq), p->length, static_cast<unsigned long long>(n_seg), p->d_energy, seg_per_wg);
        GSH_HIP(hipGetLastError());
        hipLaunchKernelGGL(gsh::pb_decide_kernel, dim3(1), dim3(64), 0, p->stream, p->d_energy, static_cast<unsigned long long>(n_seg), p->d_mask, p->d_state,
            p->n_segments_est, p->n_segments_reset, p->thres);
        GSH_HIP(hipGetLastError());
        const uint64_t n = n_seg * L;
        const unsigned blocks_a = static_cast<unsigned>(std::min<uint64_t>((n + gsh::PB_THREADS - 1) / gsh::PB_THREADS, 4096));
        hipLaunchKernelGGL(gsh::pb_apply_kernel, dim3(blocks_a), dim3(gsh::PB_THREADS), 0, p->stream, static_cast<const float2*>(device_in_iq),
            static_cast<float2*>(device_out_iq), p->d_mask, p->length, static_cast<unsigned long long>(n));
        GSH_HIP(hipGetLastError());
        GSH_HIP(hipStreamSynchronize(p->stream));
        *n_done = n;
        return GSH_OK;
    }

    int gsh_pb_get_state(gsh_pb_t* p, float* noise_power_estimation, int32_t* n_segments, int32_t* last_filtered)
    {
        GSH_REQUIRE(p != nullptr, "null handle");
        GSH_HIP(hipSetDevice(p->device));
        gsh_pulse_blanking::State s{};
        GSH_HIP(hipMemcpy(&s, p->d_state, sizeof(s), hipMemcpyDeviceToHost));
        if (noise_power_estimation) *noise_power_estimation = s.noise_power_estimation;
        if (n_segments) *n_segments = s.n_segments;
        if (last_filtered) *last_filtered = s.last_filtered;
        return GSH_OK;
    }
}
