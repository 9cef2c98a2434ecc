/* TEST INFRASTRUCTURE: stand-in for <gnuradio/math.h>.  The reference calls gr::fast_atan2f (GNU Radio's table-based
 * approximation, un-vendored third-party code, version unpinned) in pll_four_quadrant_atan / fll_four_quadrant_atan only
 * (tracking_discriminators.cc:57, 86).  Here it is the exact atan2f: the two functions that use it are therefore
 * "parity unpinned" against GNU Radio's approximation and are compared at its published accuracy only. */
#ifndef SHIM_GR_MATH_H
#define SHIM_GR_MATH_H
#include <cmath>
namespace gr
{
inline float fast_atan2f(float y, float x) { return std::atan2(y, x); }
}  // namespace gr
#endif
