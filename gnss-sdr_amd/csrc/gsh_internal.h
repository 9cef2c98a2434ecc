// Internal helpers shared by the HIP translation units of libgnss_sdr_hip.so.
// Not part of the ABI (include/gnss_sdr_hip.h is).
#ifndef GSH_INTERNAL_H
#define GSH_INTERNAL_H

#include "gnss_sdr_hip.h"
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>

namespace gsh
{
// thread-local last-error text behind gsh_last_error()
char* err_buf();
int set_error(int code, const char* fmt, ...);

inline int hip_fail(hipError_t e, const char* what, const char* file, int line)
{
    return set_error(GSH_ERR_HIP, "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
}

// select `device` for the calling thread; GSH_ERR_NO_DEVICE if it does not exist
int use_device(int device);

// inverse of the regularized lower incomplete gamma function P(a, x) in x (boost::math::gamma_p_inv); acquisition_api.hip
double gamma_p_inv(double a, double p);

// XCD-aware remap of a linear work-group id (MI355X: block b runs on XCD b % 8, each XCD has
// its own 4 MiB L2).  Consecutive *logical* ids land on the same XCD so that jobs which share
// input (same epoch, neighbouring channels) share an L2.  Bijective for any grid size.
__device__ __forceinline__ unsigned xcd_remap(unsigned b, unsigned grid)
{
    constexpr unsigned NXCD = 8;
    const unsigned xcd = b % NXCD;
    const unsigned slot = b / NXCD;
    const unsigned q = grid / NXCD;
    const unsigned r = grid % NXCD;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}
}  // namespace gsh

#define GSH_HIP(call)                                                          \
    do                                                                         \
        {                                                                      \
            hipError_t e__ = (call);                                           \
            if (e__ != hipSuccess) return gsh::hip_fail(e__, #call, __FILE__, __LINE__); \
        }                                                                      \
    while (0)

#define GSH_REQUIRE(cond, ...)                                       \
    do                                                               \
        {                                                            \
            if (!(cond)) return gsh::set_error(GSH_ERR_INVALID, __VA_ARGS__); \
        }                                                            \
    while (0)

#endif
