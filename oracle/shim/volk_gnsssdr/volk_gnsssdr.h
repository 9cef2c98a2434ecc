/*
 * TEST INFRASTRUCTURE ONLY (oracle/_ref build).
 *
 * Stand-in for the generated <volk_gnsssdr/volk_gnsssdr.h> of the reference's
 * volk_gnsssdr library, whose real header is produced by a CMake+Mako
 * generator that cannot run in this image (python-mako is absent).
 *
 * It only (a) pulls in the reference's own hand-written support headers from
 * /root/reference/.../volk_gnsssdr/include and (b) declares the dispatcher
 * entry points that the reference C++ code calls
 * (cpu_multicorrelator_real_codes.cc:81-98,117-124).  The definitions live in
 * oracle/ref_kernels.c, where each dispatcher is bound to ONE named
 * protokernel of the reference (`_generic` by default = the parity oracle;
 * `_u_avx` / `_u_sse4_1` for the "what volk would dispatch on x86" baseline).
 *
 * No reference source text is copied here.
 */
#ifndef ORACLE_SHIM_VOLK_GNSSSDR_H
#define ORACLE_SHIM_VOLK_GNSSSDR_H

#include <volk_gnsssdr/volk_gnsssdr_common.h>
#include <volk_gnsssdr/volk_gnsssdr_complex.h>
#include <volk_gnsssdr/volk_gnsssdr_malloc.h>
#include <stddef.h>
#include <stdint.h>

__VOLK_DECL_BEGIN

size_t volk_gnsssdr_get_alignment(void);

void volk_gnsssdr_32f_xn_resampler_32f_xn(float** result, const float* local_code,
    float rem_code_phase_chips, float code_phase_step_chips, float* shifts_chips,
    unsigned int code_length_chips, int num_out_vectors, unsigned int num_points);

void volk_gnsssdr_32f_xn_high_dynamics_resampler_32f_xn(float** result, const float* local_code,
    float rem_code_phase_chips, float code_phase_step_chips, float code_phase_rate_step_chips,
    float* shifts_chips, unsigned int code_length_chips, int num_out_vectors, unsigned int num_points);

void volk_gnsssdr_32fc_32f_rotator_dot_prod_32fc_xn(lv_32fc_t* result, const lv_32fc_t* in_common,
    const lv_32fc_t phase_inc, lv_32fc_t* phase, const float** in_a, int num_a_vectors,
    unsigned int num_points);

void volk_gnsssdr_32fc_32f_high_dynamic_rotator_dot_prod_32fc_xn(lv_32fc_t* result,
    const lv_32fc_t* in_common, const lv_32fc_t phase_inc, const lv_32fc_t phase_inc_rate,
    lv_32fc_t* phase, const float** in_a, int num_a_vectors, unsigned int num_points);

/* the 16-bit family's two dispatchers (cpu_multicorrelator_16sc.cc:68-75,93) */
void volk_gnsssdr_16ic_xn_resampler_16ic_xn(lv_16sc_t** result, const lv_16sc_t* local_code,
    float rem_code_phase_chips, float code_phase_step_chips, float* shifts_chips,
    unsigned int code_length_chips, int num_out_vectors, unsigned int num_points);

void volk_gnsssdr_16ic_x2_rotator_dot_prod_16ic_xn(lv_16sc_t* result, const lv_16sc_t* in_common,
    const lv_32fc_t phase_inc, lv_32fc_t* phase, const lv_16sc_t** in_a, int num_a_vectors,
    unsigned int num_points);

/* dispatchers the acquisition blocks call (pcps_acquisition.cc:280,419,466,512,655); bound to the `_generic` protokernels */
void volk_gnsssdr_s32f_sincos_32fc(lv_32fc_t* out, const float phase_inc, float* phase, unsigned int num_points);
void volk_gnsssdr_32f_index_max_32u(uint32_t* target, const float* src0, uint32_t num_points);
void volk_gnsssdr_16ic_convert_32fc(lv_32fc_t* outputVector, const lv_16sc_t* inputVector, unsigned int num_points);

__VOLK_DECL_END

#endif
