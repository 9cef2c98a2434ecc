/*
 * TEST INFRASTRUCTURE ONLY (oracle/_ref build) -- never linked into the product.
 *
 * Compiles the reference's volk_gnsssdr protokernels *from where they lie*
 * under /root/reference (the kernel headers are #included, not copied) and
 * gives them external linkage so that (a) the reference's own
 * cpu_multicorrelator_real_codes.cc links against them through the
 * dispatcher names and (b) tests can call single protokernels directly.
 *
 * Built twice by oracle/Makefile:
 *   -DREF_FLAVOUR_GENERIC  (gcc -O2 -ffp-contract=off, no -mavx/-mfma)
 *        -> the pinned parity oracle, SURVEY.md section 7 "Hard parts" (i)
 *   -DREF_FLAVOUR_SIMD     (adds -mavx -msse4.1)
 *        -> the protokernels volk_gnsssdr would dispatch to on an x86 host,
 *           used only as the CPU *timing* baseline.
 *
 * Kernel sources (all paths relative to
 * /root/reference/src/algorithms/libs/volk_gnsssdr_module/volk_gnsssdr/kernels/volk_gnsssdr/):
 *   volk_gnsssdr_32f_xn_resampler_32f_xn.h                      :63 generic, :440 u_avx
 *   volk_gnsssdr_32f_xn_high_dynamics_resampler_32f_xn.h        :67 generic, :519 u_avx
 *   volk_gnsssdr_32fc_32f_rotator_dot_prod_32fc_xn.h            :66 generic, :155 u_avx
 *   volk_gnsssdr_32fc_32f_high_dynamic_rotator_dot_prod_32fc_xn.h :68 generic
 *   volk_gnsssdr_s32f_sincos_32fc.h                             :390 generic
 *   volk_gnsssdr_32f_index_max_32u.h                            :446 generic
 *   volk_gnsssdr_16ic_xn_resampler_16ic_xn.h                    :60 generic, :438 u_avx
 *   volk_gnsssdr_16ic_x2_rotator_dot_prod_16ic_xn.h             :66 generic, :596 u_sse3
 */
#include <stdlib.h>
#include <string.h>

#if defined(REF_FLAVOUR_SIMD)
#define LV_HAVE_AVX 1
#define LV_HAVE_SSE3 1
#define LV_HAVE_SSE4_1 1
#define LV_HAVE_SSE2 1
#define LV_HAVE_SSE 1
#include <immintrin.h>
#define FLV(name) ref_simd_##name
#else
#define FLV(name) ref_generic_##name
#endif
#define LV_HAVE_GENERIC 1

#include <volk_gnsssdr/volk_gnsssdr.h>
#include "volk_gnsssdr_32f_xn_resampler_32f_xn.h"
#include "volk_gnsssdr_32f_xn_high_dynamics_resampler_32f_xn.h"
#include "volk_gnsssdr_32fc_32f_rotator_dot_prod_32fc_xn.h"
#include "volk_gnsssdr_32fc_32f_high_dynamic_rotator_dot_prod_32fc_xn.h"
#include "volk_gnsssdr_s32f_sincos_32fc.h"
#include "volk_gnsssdr_32f_index_max_32u.h"
#include "volk_gnsssdr_16ic_convert_32fc.h"
#include "volk_gnsssdr_16ic_xn_resampler_16ic_xn.h"
#include "volk_gnsssdr_16ic_x2_rotator_dot_prod_16ic_xn.h"

void FLV(resampler)(float** result, const float* local_code, float rem, float step,
    float* shifts, unsigned int code_len, int n_vec, unsigned int n)
{
#if defined(REF_FLAVOUR_SIMD)
    volk_gnsssdr_32f_xn_resampler_32f_xn_u_avx(result, local_code, rem, step, shifts, code_len, n_vec, n);
#else
    volk_gnsssdr_32f_xn_resampler_32f_xn_generic(result, local_code, rem, step, shifts, code_len, n_vec, n);
#endif
}

void FLV(hd_resampler)(float** result, const float* local_code, float rem, float step,
    float rate, float* shifts, unsigned int code_len, int n_vec, unsigned int n)
{
#if defined(REF_FLAVOUR_SIMD)
    volk_gnsssdr_32f_xn_high_dynamics_resampler_32f_xn_u_avx(result, local_code, rem, step, rate, shifts, code_len, n_vec, n);
#else
    volk_gnsssdr_32f_xn_high_dynamics_resampler_32f_xn_generic(result, local_code, rem, step, rate, shifts, code_len, n_vec, n);
#endif
}

void FLV(rotator_dot_prod)(lv_32fc_t* result, const lv_32fc_t* in, const lv_32fc_t phase_inc,
    lv_32fc_t* phase, const float** in_a, int n_vec, unsigned int n)
{
#if defined(REF_FLAVOUR_SIMD)
    volk_gnsssdr_32fc_32f_rotator_dot_prod_32fc_xn_u_avx(result, in, phase_inc, phase, in_a, n_vec, n);
#else
    volk_gnsssdr_32fc_32f_rotator_dot_prod_32fc_xn_generic(result, in, phase_inc, phase, in_a, n_vec, n);
#endif
}

void FLV(hd_rotator_dot_prod)(lv_32fc_t* result, const lv_32fc_t* in, const lv_32fc_t phase_inc,
    const lv_32fc_t phase_inc_rate, lv_32fc_t* phase, const float** in_a, int n_vec, unsigned int n)
{
    /* only generic protokernels exist for this kernel */
    volk_gnsssdr_32fc_32f_high_dynamic_rotator_dot_prod_32fc_xn_generic(result, in, phase_inc, phase_inc_rate, phase, in_a, n_vec, n);
}

/* the 16-bit family */
void FLV(resampler_16ic)(lv_16sc_t** result, const lv_16sc_t* local_code, float rem, float step,
    float* shifts, unsigned int code_len, int n_vec, unsigned int n)
{
#if defined(REF_FLAVOUR_SIMD)
    volk_gnsssdr_16ic_xn_resampler_16ic_xn_u_avx(result, local_code, rem, step, shifts, code_len, n_vec, n);
#else
    volk_gnsssdr_16ic_xn_resampler_16ic_xn_generic(result, local_code, rem, step, shifts, code_len, n_vec, n);
#endif
}

void FLV(rotator_dot_prod_16ic)(lv_16sc_t* result, const lv_16sc_t* in, const lv_32fc_t phase_inc,
    lv_32fc_t* phase, const lv_16sc_t** in_a, int n_vec, unsigned int n)
{
#if defined(REF_FLAVOUR_SIMD)
    volk_gnsssdr_16ic_x2_rotator_dot_prod_16ic_xn_u_sse3(result, in, phase_inc, phase, in_a, n_vec, n);
#else
    volk_gnsssdr_16ic_x2_rotator_dot_prod_16ic_xn_generic(result, in, phase_inc, phase, in_a, n_vec, n);
#endif
}

#if !defined(REF_FLAVOUR_SIMD)
void ref_generic_sincos(lv_32fc_t* out, float phase_inc, float* phase, unsigned int n)
{
    volk_gnsssdr_s32f_sincos_32fc_generic(out, phase_inc, phase, n);
}

void ref_generic_index_max(uint32_t* target, const float* src, uint32_t n)
{
    volk_gnsssdr_32f_index_max_32u_generic(target, src, n);
}

/* dispatcher names of the kernels the acquisition blocks call, bound to the `_generic` protokernels */
void volk_gnsssdr_s32f_sincos_32fc(lv_32fc_t* out, const float phase_inc, float* phase, unsigned int n)
{
    volk_gnsssdr_s32f_sincos_32fc_generic(out, phase_inc, phase, n);
}

void volk_gnsssdr_32f_index_max_32u(uint32_t* target, const float* src, uint32_t n)
{
    volk_gnsssdr_32f_index_max_32u_generic(target, src, n);
}

void volk_gnsssdr_16ic_convert_32fc(lv_32fc_t* out, const lv_16sc_t* in, unsigned int n)
{
    volk_gnsssdr_16ic_convert_32fc_generic(out, in, n);
}
#endif
