/* TEST INFRASTRUCTURE: stand-in for <gnuradio/gr_complex.h> (GNU Radio is not installed in this image) so that the
 * reference's tracking_discriminators.cc compiles from where it lies.  gr_complex IS std::complex<float> upstream. */
#ifndef SHIM_GR_COMPLEX_H
#define SHIM_GR_COMPLEX_H
#include <complex>
typedef std::complex<float> gr_complex;
#endif
