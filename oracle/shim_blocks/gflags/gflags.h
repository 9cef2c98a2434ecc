/* TEST INFRASTRUCTURE ONLY: gflags is absent.  The reference's gnss_sdr_flags.{h,cc} compile against these macros: a flag is a plain
 * global FLAGS_<name> initialised with the default the reference's file states; validators are accepted and ignored. */
#ifndef ORACLE_SHIM_GFLAGS_H
#define ORACLE_SHIM_GFLAGS_H
#include <cstdint>
#include <string>
namespace google
{
typedef int32_t int32;
typedef int64_t int64;
typedef uint32_t uint32;
typedef uint64_t uint64;
template <typename T, typename F>
inline bool RegisterFlagValidator(const T*, F) { return true; }
inline void ParseCommandLineFlags(int*, char***, bool) {}
inline void SetUsageMessage(const std::string&) {}
inline void SetVersionString(const std::string&) {}
inline void ShutDownCommandLineFlags() {}
}  // namespace google
namespace gflags = google;
#define DECLARE_bool(name) extern bool FLAGS_##name
#define DECLARE_int32(name) extern int32_t FLAGS_##name
#define DECLARE_int64(name) extern int64_t FLAGS_##name
#define DECLARE_uint32(name) extern uint32_t FLAGS_##name
#define DECLARE_uint64(name) extern uint64_t FLAGS_##name
#define DECLARE_double(name) extern double FLAGS_##name
#define DECLARE_string(name) extern std::string FLAGS_##name
#define DEFINE_bool(name, val, txt) bool FLAGS_##name = val
#define DEFINE_int32(name, val, txt) int32_t FLAGS_##name = val
#define DEFINE_int64(name, val, txt) int64_t FLAGS_##name = val
#define DEFINE_uint32(name, val, txt) uint32_t FLAGS_##name = val
#define DEFINE_uint64(name, val, txt) uint64_t FLAGS_##name = val
#define DEFINE_double(name, val, txt) double FLAGS_##name = val
#define DEFINE_string(name, val, txt) std::string FLAGS_##name = val
#define DEFINE_validator(name, fn) static const bool name##_validator_registered = true
#endif
