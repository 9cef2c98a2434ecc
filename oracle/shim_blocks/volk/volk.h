/*
 * TEST INFRASTRUCTURE ONLY (oracle/_ref build of the reference's acquisition blocks).
 *
 * Upstream VOLK is not vendored in /root/reference nor installed here.  These are the element-wise kernels the
 * acquisition blocks call, written as the plain loops VOLK's `_generic` implementations define (float32, one IEEE
 * operation per arithmetic step; the library is compiled with -ffp-contract=off).  volk_32f_accumulator_s32f sums
 * sequentially -- VOLK's SIMD flavours use lane-wise partial sums, so the last bits of that sum are machine
 * dependent in the reference too.
 */
#ifndef ORACLE_SHIM_VOLK_H
#define ORACLE_SHIM_VOLK_H
#include <volk/volk_complex.h>
#include <complex>
#include <cstring>  /* upstream volk.h pulls these in transitively; the blocks rely on it */
#include <numeric>

static inline void volk_32fc_conjugate_32fc(lv_32fc_t* out, const lv_32fc_t* in, unsigned int n)
{
    for (unsigned int i = 0; i < n; i++) out[i] = lv_cmake(lv_creal(in[i]), -lv_cimag(in[i]));
}
static inline void volk_32fc_x2_multiply_32fc(lv_32fc_t* out, const lv_32fc_t* a, const lv_32fc_t* b, unsigned int n)
{
    for (unsigned int i = 0; i < n; i++)
        {
            const float ar = lv_creal(a[i]), ai = lv_cimag(a[i]), br = lv_creal(b[i]), bi = lv_cimag(b[i]);
            out[i] = lv_cmake(ar * br - ai * bi, ar * bi + ai * br);
        }
}
static inline void volk_32fc_s32fc_multiply_32fc(lv_32fc_t* out, const lv_32fc_t* a, const lv_32fc_t s, unsigned int n)
{
    const float br = lv_creal(s), bi = lv_cimag(s);
    for (unsigned int i = 0; i < n; i++)
        {
            const float ar = lv_creal(a[i]), ai = lv_cimag(a[i]);
            out[i] = lv_cmake(ar * br - ai * bi, ar * bi + ai * br);
        }
}
static inline void volk_32fc_s32fc_multiply2_32fc(lv_32fc_t* out, const lv_32fc_t* a, const lv_32fc_t* s, unsigned int n)
{
    volk_32fc_s32fc_multiply_32fc(out, a, *s, n);
}
static inline void volk_32fc_magnitude_squared_32f(float* out, const lv_32fc_t* in, unsigned int n)
{
    for (unsigned int i = 0; i < n; i++)
        {
            const float r = lv_creal(in[i]), q = lv_cimag(in[i]);
            out[i] = r * r + q * q;
        }
}
static inline void volk_32f_x2_add_32f(float* out, const float* a, const float* b, unsigned int n)
{
    for (unsigned int i = 0; i < n; i++) out[i] = a[i] + b[i];
}
static inline void volk_32f_s32f_multiply_32f(float* out, const float* a, const float s, unsigned int n)
{
    for (unsigned int i = 0; i < n; i++) out[i] = a[i] * s;
}
static inline void volk_32f_accumulator_s32f(float* result, const float* in, unsigned int n)
{
    float acc = 0.0f;
    for (unsigned int i = 0; i < n; i++) acc += in[i];
    *result = acc;
}
/* sample-format converters referenced by src/algorithms/libs/item_type_helpers.cc (only item_type_valid / item_type_size of that
 * file are used by Acq_Conf; these exist so that the reference's file compiles unchanged) */
#include <cmath>
#include <cstdint>
static inline void volk_8i_convert_16i(int16_t* out, const int8_t* in, unsigned int n)
{
    for (unsigned int i = 0; i < n; i++) out[i] = static_cast<int16_t>(in[i]) * 256;
}
static inline void volk_8i_s32f_convert_32f(float* out, const int8_t* in, const float scalar, unsigned int n)
{
    for (unsigned int i = 0; i < n; i++) out[i] = static_cast<float>(in[i]) / scalar;
}
static inline void volk_16i_convert_8i(int8_t* out, const int16_t* in, unsigned int n)
{
    for (unsigned int i = 0; i < n; i++) out[i] = static_cast<int8_t>(in[i] >> 8);
}
static inline void volk_16i_s32f_convert_32f(float* out, const int16_t* in, const float scalar, unsigned int n)
{
    for (unsigned int i = 0; i < n; i++) out[i] = static_cast<float>(in[i]) / scalar;
}
static inline void volk_32f_s32f_convert_8i(int8_t* out, const float* in, const float scalar, unsigned int n)
{
    for (unsigned int i = 0; i < n; i++) out[i] = static_cast<int8_t>(std::fmax(-128.0f, std::fmin(127.0f, std::rint(in[i] * scalar))));
}
static inline void volk_32f_s32f_convert_16i(int16_t* out, const float* in, const float scalar, unsigned int n)
{
    for (unsigned int i = 0; i < n; i++) out[i] = static_cast<int16_t>(std::fmax(-32768.0f, std::fmin(32767.0f, std::rint(in[i] * scalar))));
}
/* ---- kernels of the input-filter blocks (pulse_blanking_cc.cc, notch_cc.cc, notch_lite_cc.cc); VOLK 3.x `_generic` definitions restated */
static inline size_t volk_get_alignment() { return 32; }
static inline void volk_32fc_x2_conjugate_dot_prod_32fc(lv_32fc_t* result, const lv_32fc_t* a, const lv_32fc_t* b, unsigned int n)
{
    /* sum a[i] * conj(b[i]), sequential */
    float sr = 0.0f, si = 0.0f;
    for (unsigned int i = 0; i < n; i++)
        {
            const float ar = lv_creal(a[i]), ai = lv_cimag(a[i]), br = lv_creal(b[i]), bi = lv_cimag(b[i]);
            sr += ar * br + ai * bi;
            si += ai * br - ar * bi;
        }
    *result = lv_cmake(sr, si);
}
static inline void volk_32fc_x2_multiply_conjugate_32fc(lv_32fc_t* out, const lv_32fc_t* a, const lv_32fc_t* b, unsigned int n)
{
    for (unsigned int i = 0; i < n; i++)
        {
            const float ar = lv_creal(a[i]), ai = lv_cimag(a[i]), br = lv_creal(b[i]), bi = lv_cimag(b[i]);
            out[i] = lv_cmake(ar * br + ai * bi, ai * br - ar * bi);
        }
}
static inline void volk_32fc_s32f_atan2_32f(float* out, const lv_32fc_t* in, const float normalize_factor, unsigned int n)
{
    const float inv = 1.0f / normalize_factor;
    for (unsigned int i = 0; i < n; i++) out[i] = std::atan2(lv_cimag(in[i]), lv_creal(in[i])) * inv;
}
static inline void volk_32fc_s32f_power_spectrum_32f(float* out, const lv_32fc_t* in, const float normalization_factor, unsigned int n)
{
    /* 10 log10(|x / normalization|^2)  (the library's generic kernel goes through a base-2 logarithm that returns -127 for 0) */
    const float inv = 1.0f / normalization_factor;
    for (unsigned int i = 0; i < n; i++)
        {
            const float re = lv_creal(in[i]) * inv, im = lv_cimag(in[i]) * inv;
            const float p = re * re + im * im;
            out[i] = p > 0.0f ? 3.01029995663981209120f * std::log2(p) : 3.01029995663981209120f * -127.0f;
        }
}
static inline void volk_32f_s32f_calc_spectral_noise_floor_32f(float* noise_floor, const float* in, const float spectral_exclusion_value, unsigned int n)
{
    /* mean of the points that do not exceed (mean of all points + exclusion value) */
    float sum = 0.0f;
    for (unsigned int i = 0; i < n; i++) sum += in[i];
    const float mean_amplitude = sum / static_cast<float>(n) + spectral_exclusion_value;
    sum = 0.0f;
    unsigned int kept = n;
    for (unsigned int i = 0; i < n; i++)
        {
            if (in[i] <= mean_amplitude)
                sum += in[i];
            else
                kept--;
        }
    *noise_floor = kept == 0 ? mean_amplitude : sum / static_cast<float>(kept);
}

#endif
