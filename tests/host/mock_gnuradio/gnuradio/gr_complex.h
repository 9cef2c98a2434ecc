#ifndef MOCK_GR_COMPLEX_H
#define MOCK_GR_COMPLEX_H
#include <complex>
typedef std::complex<float> gr_complex;
#endif
