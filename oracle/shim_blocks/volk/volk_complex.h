/* TEST INFRASTRUCTURE ONLY: the two typedefs of upstream VOLK's volk_complex.h the reference's blocks use. */
#ifndef ORACLE_SHIM_VOLK_COMPLEX_H
#define ORACLE_SHIM_VOLK_COMPLEX_H
#include <volk_gnsssdr/volk_gnsssdr_complex.h>
#endif
