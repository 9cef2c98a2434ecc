// Library-level entry points of the C ABI (include/gnss_sdr_hip.h): version, device
// discovery, thread-local error text.
#include "gsh_internal.h"

namespace gsh
{
char* err_buf()
{
    static thread_local char buf[512] = "";
    return buf;
}

int set_error(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

int use_device(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return set_error(GSH_ERR_NO_DEVICE, "no HIP device visible (%s)", hipGetErrorString(e));
    if (device < 0 || device >= n) return set_error(GSH_ERR_NO_DEVICE, "device %d out of range (0..%d)", device, n - 1);
    e = hipSetDevice(device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice", __FILE__, __LINE__);
    return GSH_OK;
}
}  // namespace gsh

namespace
{
// streaming read of a buffer: every work-group walks its own contiguous slice with 16-byte loads (a wave reads 1 KiB per instruction), four loads in flight per lane;
// the sums only exist so that the loads cannot be dropped
__global__ __launch_bounds__(256) void read_probe_kernel(const float4* __restrict__ src, size_t n_vec, float* __restrict__ sink)
{
    const size_t per_group = (n_vec + gridDim.x - 1) / gridDim.x;
    const size_t lo = per_group * blockIdx.x;
    const size_t hi = lo + per_group < n_vec ? lo + per_group : n_vec;
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    size_t i = lo + threadIdx.x;
    for (; i + 3 * 256 < hi; i += 4 * 256)
        {
            const float4 a = src[i], b = src[i + 256], c = src[i + 512], d = src[i + 768];
            acc.x += a.x + b.x + c.x + d.x;
            acc.y += a.y + b.y + c.y + d.y;
            acc.z += a.z + b.z + c.z + d.z;
            acc.w += a.w + b.w + c.w + d.w;
        }
    for (; i < hi; i += 256)
        {
            const float4 a = src[i];
            acc.x += a.x;
            acc.y += a.y;
            acc.z += a.z;
            acc.w += a.w;
        }
    const float s = (acc.x + acc.y) + (acc.z + acc.w);
    if (s == 12345.678f) sink[blockIdx.x] = s;  // (never true for the zero-filled buffer; keeps the loads alive)
}
}  // namespace

extern "C"
{
    int gsh_abi_version(void) { return GSH_ABI_VERSION; }

    int gsh_probe_read_bandwidth(int device, uint64_t bytes, int reps, double* gb_per_s)
    {
        GSH_REQUIRE(gb_per_s != nullptr, "null argument");
        GSH_REQUIRE(bytes >= (1ull << 20) && reps >= 1, "probe of %llu bytes x %d", static_cast<unsigned long long>(bytes), reps);
        int rc = gsh::use_device(device);
        if (rc != GSH_OK) return rc;
        const size_t n_vec = static_cast<size_t>(bytes / sizeof(float4));
        float4* buf = nullptr;
        float* sink = nullptr;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        hipStream_t st = nullptr;
        const int groups = 256 * 16;  // sixteen work-groups per compute unit: the slices are long, the chip is full
        auto cleanup = [&]() {
            if (buf) (void)hipFree(buf);
            if (sink) (void)hipFree(sink);
            if (e0) (void)hipEventDestroy(e0);
            if (e1) (void)hipEventDestroy(e1);
            if (st) (void)hipStreamDestroy(st);
        };
#define GSH_PROBE(call)                                                  \
    do                                                                   \
        {                                                                \
            const hipError_t e__ = (call);                               \
            if (e__ != hipSuccess)                                       \
                {                                                        \
                    cleanup();                                           \
                    return gsh::hip_fail(e__, #call, __FILE__, __LINE__); \
                }                                                        \
        }                                                                \
    while (0)
        GSH_PROBE(hipMalloc(&buf, n_vec * sizeof(float4)));
        GSH_PROBE(hipMalloc(&sink, sizeof(float) * groups));
        GSH_PROBE(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        GSH_PROBE(hipMemsetAsync(buf, 0, n_vec * sizeof(float4), st));
        GSH_PROBE(hipEventCreate(&e0));
        GSH_PROBE(hipEventCreate(&e1));
        for (int i = 0; i < 3; i++) hipLaunchKernelGGL(read_probe_kernel, dim3(groups), dim3(256), 0, st, buf, n_vec, sink);  // clocks up
        GSH_PROBE(hipEventRecord(e0, st));
        for (int i = 0; i < reps; i++) hipLaunchKernelGGL(read_probe_kernel, dim3(groups), dim3(256), 0, st, buf, n_vec, sink);
        GSH_PROBE(hipEventRecord(e1, st));
        GSH_PROBE(hipEventSynchronize(e1));
        float ms = 0.0f;
        GSH_PROBE(hipEventElapsedTime(&ms, e0, e1));
#undef GSH_PROBE
        cleanup();
        *gb_per_s = static_cast<double>(n_vec * sizeof(float4)) * reps / (static_cast<double>(ms) * 1e-3) / 1e9;
        return GSH_OK;
    }

    int gsh_device_count(void)
    {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess)
            {
                (void)hipGetLastError();
                return 0;
            }
        return n;
    }

    const char* gsh_last_error(void) { return gsh::err_buf(); }

    int gsh_device_name(int device, char* buf, size_t buflen)
    {
        GSH_REQUIRE(buf != nullptr && buflen > 0, "null buffer");
        int rc = gsh::use_device(device);
        if (rc != GSH_OK) return rc;
        hipDeviceProp_t p;
        GSH_HIP(hipGetDeviceProperties(&p, device));
        snprintf(buf, buflen, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
        return GSH_OK;
    }
}
