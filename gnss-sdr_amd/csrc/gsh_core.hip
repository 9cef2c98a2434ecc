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

extern "C"
{
    int gsh_abi_version(void) { return GSH_ABI_VERSION; }

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
