// Frequency-translating decimating FIR filter on the device: what gnss-sdr's input filter adapters instantiate from GNU Radio
// (src/algorithms/input_filter/adapters/freq_xlating_fir_filter.cc:115-161: gr::filter::freq_xlating_fir_filter_{ccf,fcf,scf}::make(decimation,
// taps, IF, sampling_frequency); fir_filter.cc: gr::filter::fir_filter_ccf, the same with IF = 0 and decimation 1).  GNU Radio is not vendored
// in the reference, so the definition is restated from its documentation: the band around `center_freq` is moved to 0 Hz, filtered with the
// real low-pass taps h[0..K), and every D-th result is kept:
//     y[m] = sum_{k<K} h[k] * x[mD - k] * exp(-j 2 pi f_c (mD - k) / f_s),      x[n] = 0 for n < 0 (a fresh GNU Radio buffer is zero history)
// (GNU Radio rotates the taps and de-rotates the decimated output instead -- the same sum).  The taps themselves come from the adapter
// (pm_remez / firdes on the host, once); this is the per-sample part.  The translation phase is computed exactly per sample (double
// revolutions reduced mod 1, like the acquisition wipe-off), not by a float recurrence.
//
// HBM-bound streaming kernel: 8 bytes in per input sample + 8 / D out; K MACs per output sample.  One work-group produces a tile of outputs:
// it stages the translated inputs its tile needs (tile * D + K - 1 samples) in LDS once, then every thread forms its outputs from LDS with
// the taps read as wave-uniform LDS broadcasts.
#include "gsh_internal.h"
#include <cmath>
#include <new>
#include <vector>

struct gsh_fir
{
    int device{0};
    int n_taps{0};
    int decimation{1};
    int in_kind{0};          // 0 complex64, 1 real float32, 2 real int16, 3 real int8 (ccf / fcf / scf with short or byte items)
    double rev_per_sample{0.0};  // f_c / f_s
    float* d_taps{nullptr};
    float2* d_hist{nullptr};     // the last n_taps - 1 (untranslated, converted) input samples of the stream so far
    unsigned long long n_in_total{0};   // input samples consumed so far
    unsigned long long n_out_total{0};  // outputs produced so far
    hipStream_t stream{nullptr};
};

namespace gsh
{
namespace
{
constexpr int FIR_THREADS = 256;
constexpr int FIR_TILE = 1024;   // outputs per work-group
constexpr int FIR_MAX_TAPS = 1024;
constexpr int FIR_MAX_SPAN = 12 * 1024;  // staged inputs per work-group (96 KiB of LDS as float2)

struct FirArgs
{
    const void* in;              // n_in new items
    const float2* hist;          // n_taps - 1 previous samples (converted, untranslated)
    const float* taps;
    float2* out;
    float2* hist_out;            // receives the new tail (may alias nothing else)
    unsigned long long in0;      // absolute index of in[0]
    unsigned long long out0;     // absolute index of out[0]
    unsigned long long n_in, n_out;
    double rev_per_sample;
    int n_taps, decimation, in_kind, tile;
};

__device__ __forceinline__ float2 load_item(const void* in, unsigned long long i, int kind)
{
    switch (kind)
        {
        case 0:
            return static_cast<const float2*>(in)[i];
        case 1:
            return make_float2(static_cast<const float*>(in)[i], 0.0f);
        case 2:
            return make_float2(static_cast<float>(static_cast<const short*>(in)[i]), 0.0f);
        default:
            return make_float2(static_cast<float>(static_cast<const signed char*>(in)[i]), 0.0f);
        }
}

// sample with absolute index n (may lie in the history or before the stream), translated to baseband
__device__ __forceinline__ float2 fetch_translated(const FirArgs& a, long long n)
{
    if (n < 0) return make_float2(0.0f, 0.0f);
    float2 x;
    const long long rel = n - static_cast<long long>(a.in0);
    if (rel >= 0)
        x = load_item(a.in, static_cast<unsigned long long>(rel), a.in_kind);
    else
        x = a.hist[(a.n_taps - 1) + rel];  // hist[K-1-1] is sample in0 - 1
    if (a.rev_per_sample != 0.0)
        {
            double rev = a.rev_per_sample * static_cast<double>(n);
            rev -= rint(rev);
            float s, c;
            sincospif(static_cast<float>(2.0 * rev), &s, &c);
            x = make_float2(x.x * c + x.y * s, x.y * c - x.x * s);  // x * exp(-j 2 pi rev)
        }
    return x;
}

__global__ __launch_bounds__(FIR_THREADS) void fir_kernel(FirArgs a)
{
    extern __shared__ __align__(16) float2 lds[];
    float* ltaps = reinterpret_cast<float*>(lds);           // n_taps floats (rounded up to an even count)
    float2* lx = lds + ((a.n_taps + 1) >> 1);
    const int K = a.n_taps, D = a.decimation;
    for (unsigned long long tile0 = static_cast<unsigned long long>(blockIdx.x) * a.tile; tile0 < a.n_out; tile0 += static_cast<unsigned long long>(gridDim.x) * a.tile)
        {
            const int cnt = static_cast<int>(min(static_cast<unsigned long long>(a.tile), a.n_out - tile0));
            // output m (absolute) needs inputs m D - K + 1 .. m D
            const long long first = static_cast<long long>((a.out0 + tile0) * D) - (K - 1);
            const int span = (cnt - 1) * D + K;
            __syncthreads();
            for (int i = threadIdx.x; i < K; i += FIR_THREADS) ltaps[i] = a.taps[i];
            for (int i = threadIdx.x; i < span; i += FIR_THREADS) lx[i] = fetch_translated(a, first + i);
            __syncthreads();
            for (int j = threadIdx.x; j < cnt; j += FIR_THREADS)
                {
                    // y = sum_k h[k] x[mD - k]; x[mD - k] sits at lx[j D + (K - 1) - k]
                    const float2* xp = lx + j * D + (K - 1);
                    float re = 0.0f, im = 0.0f;
                    for (int k = 0; k < K; k++)
                        {
                            const float h = ltaps[k];
                            const float2 v = xp[-k];
                            re = fmaf(h, v.x, re);
                            im = fmaf(h, v.y, im);
                        }
                    a.out[tile0 + j] = make_float2(re, im);
                }
        }
}

// keep the last K - 1 input samples (converted, untranslated) for the next call
__global__ void fir_history_kernel(FirArgs a)
{
    const int K1 = a.n_taps - 1;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < K1; i += gridDim.x * blockDim.x)
        {
            // new hist[i] = sample (in0 + n_in) - K1 + i
            const long long rel = static_cast<long long>(a.n_in) - K1 + i;
            float2 v;
            if (rel >= 0)
                v = load_item(a.in, static_cast<unsigned long long>(rel), a.in_kind);
            else
                v = (K1 + rel >= 0) ? a.hist[K1 + rel] : make_float2(0.0f, 0.0f);
            a.hist_out[i] = v;
        }
}
}  // namespace
}  // namespace gsh

extern "C"
{
    int gsh_fir_create(int device, const float* taps, int n_taps, int decimation, double center_freq_hz, double sampling_freq_hz, int input_kind,
        gsh_fir_t** out)
    {
        GSH_REQUIRE(out != nullptr && taps != nullptr, "null argument");
        *out = nullptr;
        GSH_REQUIRE(n_taps >= 1 && n_taps <= gsh::FIR_MAX_TAPS, "n_taps %d outside 1..%d", n_taps, gsh::FIR_MAX_TAPS);
        GSH_REQUIRE(decimation >= 1 && decimation <= 64, "decimation %d outside 1..64", decimation);
        GSH_REQUIRE(sampling_freq_hz > 0.0, "sampling frequency must be positive");
        GSH_REQUIRE(input_kind >= 0 && input_kind <= 3, "input_kind %d outside 0..3", input_kind);
        int rc = gsh::use_device(device);
        if (rc != GSH_OK) return rc;
        gsh_fir* f = new (std::nothrow) gsh_fir();
        GSH_REQUIRE(f != nullptr, "out of host memory");
        f->device = device;
        f->n_taps = n_taps;
        f->decimation = decimation;
        f->in_kind = input_kind;
        f->rev_per_sample = center_freq_hz / sampling_freq_hz;
        auto fail = [&](hipError_t e, const char* what) {
            gsh::hip_fail(e, what, __FILE__, __LINE__);
            gsh_fir_destroy(f);
            return GSH_ERR_HIP;
        };
        hipError_t e;
        if ((e = hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking)) != hipSuccess) return fail(e, "hipStreamCreate");
        if ((e = hipMalloc(&f->d_taps, sizeof(float) * n_taps)) != hipSuccess) return fail(e, "hipMalloc(taps)");
        if ((e = hipMemcpy(f->d_taps, taps, sizeof(float) * n_taps, hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "hipMemcpy(taps)");
        if ((e = hipMalloc(&f->d_hist, sizeof(float2) * 2 * static_cast<size_t>(n_taps))) != hipSuccess) return fail(e, "hipMalloc(hist)");
        if ((e = hipMemset(f->d_hist, 0, sizeof(float2) * 2 * static_cast<size_t>(n_taps))) != hipSuccess) return fail(e, "hipMemset(hist)");
        *out = f;
        return GSH_OK;
    }

    void gsh_fir_destroy(gsh_fir_t* f)
    {
        if (!f) return;
        (void)hipSetDevice(f->device);
        if (f->stream) (void)hipStreamSynchronize(f->stream);
        if (f->d_taps) (void)hipFree(f->d_taps);
        if (f->d_hist) (void)hipFree(f->d_hist);
        if (f->stream) (void)hipStreamDestroy(f->stream);
        delete f;
    }

    int gsh_fir_process_device(gsh_fir_t* f, const void* device_in, uint64_t n_in, void* device_out, uint64_t max_out, uint64_t* n_out, void* hip_stream)
    {
        GSH_REQUIRE(f != nullptr && n_out != nullptr, "null argument");
        *n_out = 0;
        GSH_REQUIRE(n_in == 0 || device_in != nullptr, "null input");
        GSH_HIP(hipSetDevice(f->device));
        hipStream_t s = hip_stream ? static_cast<hipStream_t>(hip_stream) : f->stream;
        // outputs m with m D < n_in_total + n_in and m >= n_out_total (output m uses inputs up to m D)
        const unsigned long long end_in = f->n_in_total + n_in;
        const unsigned long long m_end = (end_in + f->decimation - 1) / f->decimation;  // first m with m D >= end_in
        unsigned long long count = m_end > f->n_out_total ? m_end - f->n_out_total : 0ull;
        GSH_REQUIRE(count <= max_out, "the block yields %llu outputs, the destination holds %llu", count, static_cast<unsigned long long>(max_out));
        GSH_REQUIRE(count == 0 || device_out != nullptr, "null output");
        gsh::FirArgs a;
        a.in = device_in;
        a.hist = f->d_hist;
        a.hist_out = f->d_hist + f->n_taps;  // double buffer inside one allocation
        a.taps = f->d_taps;
        a.out = static_cast<float2*>(device_out);
        a.in0 = f->n_in_total;
        a.out0 = f->n_out_total;
        a.n_in = n_in;
        a.n_out = count;
        a.rev_per_sample = f->rev_per_sample;
        a.n_taps = f->n_taps;
        a.decimation = f->decimation;
        a.in_kind = f->in_kind;
        int tile = gsh::FIR_TILE;
        while (tile > 64 && (tile - 1) * f->decimation + f->n_taps > gsh::FIR_MAX_SPAN) tile >>= 1;
        a.tile = tile;
        if (count > 0)
            {
                const size_t span = static_cast<size_t>(tile - 1) * f->decimation + f->n_taps;
                const size_t lds = sizeof(float2) * (span + ((f->n_taps + 1) >> 1));
                unsigned long long blocks = (count + tile - 1) / tile;
                if (blocks > 256ull * 8ull) blocks = 256ull * 8ull;
                if (lds > 64 * 1024)
                    GSH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gsh::fir_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
                gsh::fir_kernel<<<dim3(static_cast<unsigned>(blocks)), dim3(gsh::FIR_THREADS), lds, s>>>(a);
                GSH_HIP(hipGetLastError());
            }
        if (n_in > 0 && f->n_taps > 1)
            {
                gsh::fir_history_kernel<<<dim3(4), dim3(256), 0, s>>>(a);
                GSH_HIP(hipGetLastError());
                // swap the halves: the new history becomes the current one (stream-ordered copy keeps the handle's pointer stable)
                GSH_HIP(hipMemcpyAsync(f->d_hist, f->d_hist + f->n_taps, sizeof(float2) * (f->n_taps - 1), hipMemcpyDeviceToDevice, s));
            }
        f->n_in_total = end_in;
        f->n_out_total += count;
        *n_out = count;
        if (!hip_stream) GSH_HIP(hipStreamSynchronize(s));
        return GSH_OK;
    }
}
