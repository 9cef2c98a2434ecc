// Which way should a lane get a RUN of consecutive complex64 samples?  (design probe for the run-based correlator)
//   A  coalesced float4 per lane (today's access pattern), plain sum
//   B  each lane reads its own run of L consecutive samples straight from global memory (8-byte loads, lane stride L*8 B)
//   C  work-group stages a tile coalesced into LDS, each lane then reads its run from LDS (stride L*8 B, L odd)
// 32 "channels" (work-groups with the same blockIdx % n_windows) read the same window, as the tracking launch does.
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int L = 23;
constexpr int TILE = 256 * L;
__global__ __launch_bounds__(256) void kA(const float4* __restrict__ x, float2* out, int n_pairs_per_wg, int windows)
{
    const float4* p = x + (size_t)(blockIdx.x % windows) * n_pairs_per_wg;
    float2 acc = {0, 0};
    for (int i = threadIdx.x; i < n_pairs_per_wg; i += 256)
        {
            float4 v = p[i];
            acc.x += v.x + v.z;
            acc.y += v.y + v.w;
        }
    if (acc.x == 12345.f) out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void kB(const float2* __restrict__ x, float2* out, int n_per_wg, int windows)
{
    const float2* p = x + (size_t)(blockIdx.x % windows) * n_per_wg;
    float2 acc = {0, 0};
    for (int t0 = 0; t0 + TILE <= n_per_wg; t0 += TILE)
        {
            const float2* r = p + t0 + threadIdx.x * L;
#pragma unroll
            for (int i = 0; i < L; i++)
                {
                    float2 v = r[i];
                    acc.x += v.x;
                    acc.y += v.y;
                }
        }
    if (acc.x == 12345.f) out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void kC(const float2* __restrict__ x, float2* out, int n_per_wg, int windows)
{
    __shared__ float2 xs[TILE + 2];
    const float2* p = x + (size_t)(blockIdx.x % windows) * n_per_wg;
    float2 acc = {0, 0};
    for (int t0 = 0; t0 + TILE <= n_per_wg; t0 += TILE)
        {
            const float4* g = reinterpret_cast<const float4*>(p + t0);
            __syncthreads();
            for (int i = threadIdx.x; i < TILE / 2; i += 256) reinterpret_cast<float4*>(xs)[i] = g[i];
            __syncthreads();
            const float2* r = xs + threadIdx.x * L;
#pragma unroll
            for (int i = 0; i < L; i++)
                {
                    float2 v = r[i];
                    acc.x += v.x;
                    acc.y += v.y;
                }
        }
    if (acc.x == 12345.f) out[blockIdx.x * 256 + threadIdx.x] = acc;
}
int main()
{
    const int windows = 400, channels = 32;
    const int n_per_wg = TILE * 4 + 448;  // ~24 000 samples per window
    const int used = TILE * 4;
    float2* x;
    float2* out;
    hipMalloc(&x, sizeof(float2) * (size_t)windows * n_per_wg + 4096);
    hipMemset(x, 0, sizeof(float2) * (size_t)windows * n_per_wg + 4096);
    hipMalloc(&out, sizeof(float2) * 256 * windows * channels);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int blocks = windows * channels;
    for (int which = 0; which < 3; which++)
        {
            float best = 1e9;
            for (int rep = 0; rep < 5; rep++)
                {
                    hipEventRecord(e0);
                    if (which == 0) kA<<<blocks, 256>>>(reinterpret_cast<const float4*>(x), out, used / 2, windows);
                    if (which == 1) kB<<<blocks, 256>>>(x, out, n_per_wg, windows);
                    if (which == 2) kC<<<blocks, 256>>>(x, out, n_per_wg, windows);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    best = ms < best ? ms : best;
                }
            const double samples = (double)blocks * used;
            printf("%s: %.3f ms  %.2f T channel-samples/s  (%.1f samples/clk/CU at 2.4 GHz)\n", which == 0 ? "A coalesced float4" : which == 1 ? "B per-lane runs from global" : "C per-lane runs via LDS tile",
                best, samples / best / 1e9, samples / (best * 1e-3) / 2.4e9 / 256);
        }
    return 0;
}
