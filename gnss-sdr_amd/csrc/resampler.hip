// Direct (nearest-neighbour) resampler on the device: the arithmetic of gnss-sdr's direct_resampler_conditioner_cc
// (src/algorithms/resampler/gnuradio_blocks/direct_resampler_conditioner_cc.cc:39-129), the block the signal conditioner and the
// acquisition decimator (gnss_flowgraph.cc:1165-1209) use to change the sample rate.
//
// The reference walks the input with a 32-bit phase accumulator (phase_step = floor(2^32 * f_out / f_in), :52-59) and copies a sample
// whenever the accumulator wraps (:83-95); interpolation advances the input on a wrap instead (:99-110).  The accumulator starts at 0
// and only ever adds phase_step, so its value after i steps is (i * phase_step) mod 2^32 and the number of wraps is
// floor(i * phase_step / 2^32): the j-th output of the whole stream is
//     decimation     out[j] = in[ ceil(j * 2^32 / phase_step) ]          (sample 0 is always kept: 0 <= 0 at :85)
//     interpolation  out[j] = in[ floor((j + 1) * phase_step / 2^32) ]
// -- an exact integer gather, independent of how the stream is cut into work() calls.  HBM-bound: 8 bytes in (of the samples that are
// kept) + 8 bytes out per output sample.
#include "gsh_internal.h"
#include <cmath>

namespace gsh
{
namespace
{
constexpr int RS_THREADS = 256;

struct ResampleArgs
{
    const float2* src;           // src[0] is absolute input index in0
    float2* dst;                 // dst[0] is absolute output index out0
    unsigned long long in0;
    unsigned long long n_out;
    unsigned long long q0, r0;   // see resample_plan()
    unsigned step;
    int decimating;
};

__global__ __launch_bounds__(RS_THREADS) void resample_gather_kernel(ResampleArgs a)
{
    const unsigned long long stride = static_cast<unsigned long long>(gridDim.x) * RS_THREADS;
    for (unsigned long long d = static_cast<unsigned long long>(blockIdx.x) * RS_THREADS + threadIdx.x; d < a.n_out; d += stride)
        {
            unsigned long long i;
            if (a.decimating)
                i = a.q0 + (a.r0 + (d << 32)) / a.step;          // floor((A0 + step - 1 + d 2^32) / step), A0 = out0 2^32
            else
                i = a.q0 + ((a.r0 + d * a.step) >> 32);          // floor((B0 + d step) / 2^32), B0 = (out0 + 1) step
            a.dst[d] = a.src[i - a.in0];
        }
}

// phase step exactly as the reference computes it (:52-59); 0 stands for a ratio of one (2^32 does not fit the uint32 cast)
unsigned phase_step_of(double fs_in, double fs_out, int* decimating)
{
    const double two_32 = 4294967296.0;
    *decimating = fs_in >= fs_out ? 1 : 0;
    const double v = *decimating ? std::floor(two_32 * fs_out / fs_in) : std::floor(two_32 * fs_in / fs_out);
    if (v >= two_32) return 0u;
    return static_cast<unsigned>(v);
}

// absolute input index feeding absolute output j
unsigned long long input_index_of(unsigned long long j, unsigned step, int decimating)
{
    if (step == 0) return j;
    if (decimating) return static_cast<unsigned long long>(((static_cast<unsigned __int128>(j) << 32) + step - 1) / step);
    return static_cast<unsigned long long>((static_cast<unsigned __int128>(j + 1) * step) >> 32);
}
}  // namespace
}  // namespace gsh

extern "C"
{
    int gsh_direct_resample_device(int device, const void* device_src, uint64_t in0, uint64_t n_in, double fs_in, double fs_out, uint64_t out0,
        void* device_dst, uint64_t max_out, uint64_t* n_out, uint64_t* n_in_consumed, void* hip_stream)
    {
        GSH_REQUIRE(n_out != nullptr, "null n_out");
        *n_out = 0;
        if (n_in_consumed) *n_in_consumed = 0;
        GSH_REQUIRE(fs_in > 0.0 && fs_out > 0.0, "sample rates must be positive");
        GSH_REQUIRE(n_in == 0 || device_src != nullptr, "null source");
        GSH_REQUIRE(max_out == 0 || device_dst != nullptr, "null destination");
        int decimating = 1;
        const unsigned step = gsh::phase_step_of(fs_in, fs_out, &decimating);
        GSH_REQUIRE(step != 0u || fs_in == fs_out, "resampling ratio %g too extreme for the 32-bit phase accumulator", fs_out / fs_in);
        GSH_REQUIRE(gsh::input_index_of(out0, step, decimating) >= in0 || n_in == 0, "output %llu needs input %llu, before the block's first sample %llu",
            static_cast<unsigned long long>(out0), static_cast<unsigned long long>(gsh::input_index_of(out0, step, decimating)), static_cast<unsigned long long>(in0));
        // how many outputs does this input block feed?  largest count c with input_index_of(out0 + c - 1) < in0 + n_in  (monotone: bisection)
        const unsigned long long end_in = in0 + n_in;
        unsigned long long lo = 0, hi = max_out;
        while (lo < hi)
            {
                const unsigned long long mid = lo + (hi - lo + 1) / 2;
                if (gsh::input_index_of(out0 + mid - 1, step, decimating) < end_in)
                    lo = mid;
                else
                    hi = mid - 1;
            }
        const unsigned long long count = lo;
        *n_out = count;
        // inputs the reference would have consumed once output out0 + count - 1 ... is produced: everything before the NEXT output's sample
        // (decimation: the scan stops right after the sample it copies, :83-95; interpolation: the pointer rests on the last sample used)
        if (n_in_consumed)
            {
                if (count == 0)
                    *n_in_consumed = 0;
                else if (decimating)
                    *n_in_consumed = gsh::input_index_of(out0 + count - 1, step, decimating) + 1 - in0;
                else
                    *n_in_consumed = gsh::input_index_of(out0 + count - 1, step, decimating) - in0;
            }
        if (count == 0) return GSH_OK;
        int rc = gsh::use_device(device);
        if (rc != GSH_OK) return rc;
        hipStream_t s = static_cast<hipStream_t>(hip_stream);
        if (step == 0)
            {
                GSH_HIP(hipMemcpyAsync(device_dst, static_cast<const float2*>(device_src) + (out0 - in0), sizeof(float2) * count, hipMemcpyDeviceToDevice, s));
                return GSH_OK;
            }
        gsh::ResampleArgs a;
        a.src = static_cast<const float2*>(device_src);
        a.dst = static_cast<float2*>(device_dst);
        a.in0 = in0;
        a.n_out = count;
        a.step = step;
        a.decimating = decimating;
        if (decimating)
            {
                const unsigned __int128 A = (static_cast<unsigned __int128>(out0) << 32) + step - 1;
                a.q0 = static_cast<unsigned long long>(A / step);
                a.r0 = static_cast<unsigned long long>(A % step);
            }
        else
            {
                const unsigned __int128 B = static_cast<unsigned __int128>(out0 + 1) * step;
                a.q0 = static_cast<unsigned long long>(B >> 32);
                a.r0 = static_cast<unsigned long long>(B & 0xffffffffu);
            }
        // one launch handles at most 2^32 - 1 outputs (d << 32 and d * step must stay inside 64 bits): far above any ring
        GSH_REQUIRE(count < (1ull << 32), "more than 2^32 outputs in one call");
        unsigned long long blocks = (count + gsh::RS_THREADS - 1) / gsh::RS_THREADS;
        if (blocks > 256ull * 16ull) blocks = 256ull * 16ull;
        gsh::resample_gather_kernel<<<dim3(static_cast<unsigned>(blocks)), dim3(gsh::RS_THREADS), 0, s>>>(a);
        GSH_HIP(hipGetLastError());
        return GSH_OK;
    }
}
