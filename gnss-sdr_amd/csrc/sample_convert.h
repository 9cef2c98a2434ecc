// Sample-format conversion on the device: the arithmetic of gnss-sdr's data_type_adapter blocks
// (src/algorithms/data_type_adapter/adapters/ibyte_to_complex.cc:45-51, ishort_to_complex.cc:45-51: GNU Radio's
// interleaved_char_to_complex / interleaved_short_to_complex -- plain integer -> float casts, no scaling -- optionally
// followed by conjugate_cc for inverted_spectrum) and of volk_gnsssdr_16ic_convert_32fc (acq.cc:653-656).
#ifndef GSH_SAMPLE_CONVERT_H
#define GSH_SAMPLE_CONVERT_H
#include "gsh_internal.h"

namespace gsh
{
// bytes per complex sample of a GSH_ITEM_* type; 0 when the type is unknown
size_t item_bytes(int item_type);
// d_dst[i] = (float)I[i] + j (float)Q[i]  (conjugated when `conj`), i < n; d_src holds n items of `item_type`
// (GSH_ITEM_GR_COMPLEX copies).  Source alignment: natural alignment of one item; destination 8 bytes.
int convert_to_complex(const void* d_src, int item_type, int conj, float2* d_dst, size_t n, hipStream_t s);
}  // namespace gsh
#endif
