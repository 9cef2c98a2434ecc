// HBM-bound format conversion kernels (see sample_convert.h).  Algorithmic bytes per sample: item size in + 8 out.
#include "sample_convert.h"

namespace gsh
{
namespace
{
constexpr int CV_THREADS = 256;
constexpr int CV_PER_THREAD = 4;  // samples per thread per iteration: 8 (byte) / 16 (short) bytes in, 32 bytes out

template <typename T>
struct Pair
{
    T i, q;
};

// SRC = int8_t / int16_t / float.  One thread converts CV_PER_THREAD consecutive samples when the block is whole and the
// source is aligned for a vector load, else it walks them one by one (head / tail / odd alignment).
template <typename SRC>
__global__ __launch_bounds__(CV_THREADS) void convert_kernel(const Pair<SRC>* __restrict__ src, float2* __restrict__ dst, size_t n, float qsign,
    int vec_ok)
{
    const size_t stride = static_cast<size_t>(gridDim.x) * CV_THREADS * CV_PER_THREAD;
    for (size_t base = (static_cast<size_t>(blockIdx.x) * CV_THREADS + threadIdx.x) * CV_PER_THREAD; base < n; base += stride)
        {
            if (vec_ok && base + CV_PER_THREAD <= n)
                {
                    struct alignas(sizeof(Pair<SRC>) * CV_PER_THREAD) Vec
                    {
                        Pair<SRC> v[CV_PER_THREAD];
                    };
                    const Vec in = *reinterpret_cast<const Vec*>(src + base);
                    float4 o0, o1;
                    o0.x = static_cast<float>(in.v[0].i);
                    o0.y = qsign * static_cast<float>(in.v[0].q);
                    o0.z = static_cast<float>(in.v[1].i);
                    o0.w = qsign * static_cast<float>(in.v[1].q);
                    o1.x = static_cast<float>(in.v[2].i);
                    o1.y = qsign * static_cast<float>(in.v[2].q);
                    o1.z = static_cast<float>(in.v[3].i);
                    o1.w = qsign * static_cast<float>(in.v[3].q);
                    if ((reinterpret_cast<uintptr_t>(dst + base) & 15u) == 0)
                        {
                            float4* d4 = reinterpret_cast<float4*>(dst + base);
                            d4[0] = o0;
                            d4[1] = o1;
                        }
                    else
                        {
                            dst[base] = make_float2(o0.x, o0.y);
                            dst[base + 1] = make_float2(o0.z, o0.w);
                            dst[base + 2] = make_float2(o1.x, o1.y);
                            dst[base + 3] = make_float2(o1.z, o1.w);
                        }
                }
            else
                {
                    for (size_t k = base; k < n && k < base + CV_PER_THREAD; k++)
                        dst[k] = make_float2(static_cast<float>(src[k].i), qsign * static_cast<float>(src[k].q));
                }
        }
}

template <typename SRC>
int launch_convert(const void* d_src, int conj, float2* d_dst, size_t n, hipStream_t s)
{
    const size_t per_block = static_cast<size_t>(CV_THREADS) * CV_PER_THREAD;
    size_t blocks = (n + per_block - 1) / per_block;
    if (blocks > 256u * 16u) blocks = 256u * 16u;  // grid-stride beyond 16 work-groups per CU
    const int vec_ok = (reinterpret_cast<uintptr_t>(d_src) % (sizeof(Pair<SRC>) * CV_PER_THREAD)) == 0 ? 1 : 0;
    convert_kernel<SRC><<<dim3(static_cast<unsigned>(blocks)), dim3(CV_THREADS), 0, s>>>(static_cast<const Pair<SRC>*>(d_src), d_dst, n,
        conj ? -1.0f : 1.0f, vec_ok);
    GSH_HIP(hipGetLastError());
    return GSH_OK;
}
}  // namespace

size_t item_bytes(int item_type)
{
    switch (item_type)
        {
        case GSH_ITEM_GR_COMPLEX:
            return 8;
        case GSH_ITEM_SHORT:
            return 4;
        case GSH_ITEM_BYTE:
            return 2;
        default:
            return 0;
        }
}

int convert_to_complex(const void* d_src, int item_type, int conj, float2* d_dst, size_t n, hipStream_t s)
{
    if (n == 0) return GSH_OK;
    switch (item_type)
        {
        case GSH_ITEM_GR_COMPLEX:
            if (!conj)
                {
                    GSH_HIP(hipMemcpyAsync(d_dst, d_src, n * sizeof(float2), hipMemcpyDeviceToDevice, s));
                    return GSH_OK;
                }
            return launch_convert<float>(d_src, conj, d_dst, n, s);
        case GSH_ITEM_SHORT:
            return launch_convert<int16_t>(d_src, conj, d_dst, n, s);
        case GSH_ITEM_BYTE:
            return launch_convert<int8_t>(d_src, conj, d_dst, n, s);
        default:
            return set_error(GSH_ERR_INVALID, "unknown item type %d", item_type);
        }
}
}  // namespace gsh
