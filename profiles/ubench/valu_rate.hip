// VALU issue-rate microbenchmark (gfx950): wave-instructions per cycle per CU for the instruction kinds the
// correlator / FFT kernels are made of.  Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v2 __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    v2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const float c = 1.0000001f;
    const v2 cc = {c, c};
    for (int i = 0; i < iters; i++)
        {
            if (KIND == 0)  // v_fma_f32
                {
                    REP8(asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n"
                                      "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7\n"
                                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
                }
            else if (KIND == 1)  // v_pk_fma_f32
                {
                    REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %0\n v_pk_fma_f32 %1, %1, %8, %1\n v_pk_fma_f32 %2, %2, %8, %2\n v_pk_fma_f32 %3, %3, %8, %3\n"
                                      "v_pk_fma_f32 %4, %4, %8, %4\n v_pk_fma_f32 %5, %5, %8, %5\n v_pk_fma_f32 %6, %6, %8, %6\n v_pk_fma_f32 %7, %7, %8, %7\n"
                                      : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(cc));)
                }
            else if (KIND == 2)  // v_mov_b32 / v_xor_b32 mix
                {
                    REP8(asm volatile("v_xor_b32 %0, 0x80000000, %0\n v_mov_b32 %1, %0\n v_xor_b32 %2, 0x80000000, %2\n v_mov_b32 %3, %2\n"
                                      "v_xor_b32 %4, 0x80000000, %4\n v_mov_b32 %5, %4\n v_xor_b32 %6, 0x80000000, %6\n v_mov_b32 %7, %6\n"
                                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
                }
            else if (KIND == 3)  // v_pk_add_f32
                {
                    REP8(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                                      "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                                      : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(cc));)
                }
            else if (KIND == 4)  // v_add_f32
                {
                    REP8(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                                      "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
                }
            else if (KIND == 5)  // v_cvt_flr_i32_f32 + v_cndmask
                {
                    REP8(asm volatile("v_cvt_flr_i32_f32 %0, %0\n v_cvt_f32_i32 %0, %0\n v_cvt_flr_i32_f32 %1, %1\n v_cvt_f32_i32 %1, %1\n"
                                      "v_cvt_flr_i32_f32 %2, %2\n v_cvt_f32_i32 %2, %2\n v_cvt_flr_i32_f32 %3, %3\n v_cvt_f32_i32 %3, %3\n"
                                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
                }
            else if (KIND == 6)  // v_fract_f32
                {
                    REP8(asm volatile("v_fract_f32 %0, %0\n v_fract_f32 %1, %1\n v_fract_f32 %2, %2\n v_fract_f32 %3, %3\n"
                                      "v_fract_f32 %4, %4\n v_fract_f32 %5, %5\n v_fract_f32 %6, %6\n v_fract_f32 %7, %7\n"
                                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
                }
            else if (KIND == 7)  // v_cvt_flr_i32_f32 alone (result fed back as float bits)
                {
                    REP8(asm volatile("v_cvt_flr_i32_f32 %0, %0\n v_cvt_flr_i32_f32 %1, %1\n v_cvt_flr_i32_f32 %2, %2\n v_cvt_flr_i32_f32 %3, %3\n"
                                      "v_cvt_flr_i32_f32 %4, %4\n v_cvt_flr_i32_f32 %5, %5\n v_cvt_flr_i32_f32 %6, %6\n v_cvt_flr_i32_f32 %7, %7\n"
                                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
                }
            else if (KIND == 8)  // v_lshl_add_u32
                {
                    REP8(asm volatile("v_lshl_add_u32 %0, %0, 2, %8\n v_lshl_add_u32 %1, %1, 2, %8\n v_lshl_add_u32 %2, %2, 2, %8\n v_lshl_add_u32 %3, %3, 2, %8\n"
                                      "v_lshl_add_u32 %4, %4, 2, %8\n v_lshl_add_u32 %5, %5, 2, %8\n v_lshl_add_u32 %6, %6, 2, %8\n v_lshl_add_u32 %7, %7, 2, %8\n"
                                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
                }
            else if (KIND == 9)  // v_cmp_ge_f32_e64 (SGPR pair) + v_cndmask_b32_e64
                {
                    REP8(asm volatile("v_cmp_ge_f32_e64 s[20:21], %0, %4\n v_cndmask_b32_e64 %0, %0, %1, s[20:21]\n v_cmp_ge_f32_e64 s[22:23], %1, %4\n v_cndmask_b32_e64 %1, %1, %2, s[22:23]\n"
                                      "v_cmp_ge_f32_e64 s[20:21], %2, %4\n v_cndmask_b32_e64 %2, %2, %3, s[20:21]\n v_cmp_ge_f32_e64 s[22:23], %3, %4\n v_cndmask_b32_e64 %3, %3, %0, s[22:23]\n"
                                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c) : "s20", "s21", "s22", "s23");)
                }
            else if (KIND == 10)  // v_cmp_ge_f32_e32 (vcc) + v_addc_co_u32
                {
                    REP8(asm volatile("v_cmp_ge_f32_e32 vcc, %0, %4\n v_addc_co_u32_e32 %0, vcc, %0, %1, vcc\n v_cmp_ge_f32_e32 vcc, %1, %4\n v_addc_co_u32_e32 %1, vcc, %1, %2, vcc\n"
                                      "v_cmp_ge_f32_e32 vcc, %2, %4\n v_addc_co_u32_e32 %2, vcc, %2, %3, vcc\n v_cmp_ge_f32_e32 vcc, %3, %4\n v_addc_co_u32_e32 %3, vcc, %3, %0, vcc\n"
                                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c) : "vcc");)
                }
            else if (KIND == 11)  // v_add_u32 / v_and_b32 mix
                {
                    REP8(asm volatile("v_add_u32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_and_b32 %3, %3, %8\n"
                                      "v_add_u32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_and_b32 %7, %7, %8\n"
                                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
                }
            else if (KIND == 12)  // v_pk_fma_f32 with op_sel broadcast (the correlator's multiply-accumulate)
                {
                    REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n v_pk_fma_f32 %1, %1, %8, %1 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n"
                                      "v_pk_fma_f32 %2, %2, %8, %2 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n v_pk_fma_f32 %3, %3, %8, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n"
                                      "v_pk_fma_f32 %4, %4, %8, %4 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n v_pk_fma_f32 %5, %5, %8, %5 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n"
                                      "v_pk_fma_f32 %6, %6, %8, %6 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n v_pk_fma_f32 %7, %7, %8, %7 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n"
                                      : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(cc));)
                }
            else if (KIND == 13)  // v_pk_add_f32 with an SGPR pair operand and op_sel (the chip-index chain)
                {
                    REP8(asm volatile("v_pk_add_f32 %0, %0, s[20:21] op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_add_f32 %1, %1, s[20:21] op_sel:[0,0] op_sel_hi:[1,0]\n"
                                      "v_pk_add_f32 %2, %2, s[20:21] op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_add_f32 %3, %3, s[20:21] op_sel:[0,0] op_sel_hi:[1,0]\n"
                                      "v_pk_add_f32 %4, %4, s[20:21] op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_add_f32 %5, %5, s[20:21] op_sel:[0,0] op_sel_hi:[1,0]\n"
                                      "v_pk_add_f32 %6, %6, s[20:21] op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_add_f32 %7, %7, s[20:21] op_sel:[0,0] op_sel_hi:[1,0]\n"
                                      : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : : "s20", "s21");)
                }
            else if (KIND == 14)  // v_pk_fma_f32, three distinct 64-bit sources (accumulator, sample, code pair) as in the correlator
                {
                    REP8(asm volatile("v_pk_fma_f32 %0, %4, %6, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n v_pk_fma_f32 %1, %5, %7, %1 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n"
                                      "v_pk_fma_f32 %2, %4, %7, %2 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n v_pk_fma_f32 %3, %5, %6, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n"
                                      "v_pk_fma_f32 %0, %5, %7, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n v_pk_fma_f32 %1, %4, %6, %1 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n"
                                      "v_pk_fma_f32 %2, %5, %6, %2 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n v_pk_fma_f32 %3, %4, %7, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n"
                                      : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p4), "v"(p5), "v"(p6), "v"(p7));)
                }
            else if (KIND == 15)  // v_fma_f32, three distinct sources
                {
                    REP8(asm volatile("v_fma_f32 %0, %4, %6, %0\n v_fma_f32 %1, %5, %7, %1\n v_fma_f32 %2, %4, %7, %2\n v_fma_f32 %3, %5, %6, %3\n"
                                      "v_fma_f32 %0, %5, %7, %0\n v_fma_f32 %1, %4, %6, %1\n v_fma_f32 %2, %5, %6, %2\n v_fma_f32 %3, %4, %7, %3\n"
                                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5), "v"(a6), "v"(a7));)
                }
            else if (KIND == 16)  // the mix of a correlator trip: chain add (sgpr), cvt_flr, lshlrev, two pk_fma with distinct sources, twice
                {
                    REP8(asm volatile("v_add_f32 %4, s20, %5\n v_cvt_flr_i32_f32 %6, %4\n v_lshlrev_b32 %6, 2, %6\n"
                                      "v_pk_fma_f32 %0, %8, %9, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n v_pk_fma_f32 %1, %8, %9, %1 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n"
                                      "v_add_f32 %5, s21, %4\n v_cvt_flr_i32_f32 %7, %5\n v_lshlrev_b32 %7, 2, %7\n"
                                      "v_pk_fma_f32 %2, %9, %8, %2 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n v_pk_fma_f32 %3, %9, %8, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n"
                                      : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(p6), "v"(p7) : "s20", "s21");)
                }
            else if (KIND == 17)  // v_cvt_pkrtz_f16_f32 (round 6: two floor()s of non-negative scaled chip positions per instruction)
                {
                    REP8(asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1\n v_cvt_pkrtz_f16_f32 %1, %1, %2\n v_cvt_pkrtz_f16_f32 %2, %2, %3\n v_cvt_pkrtz_f16_f32 %3, %3, %4\n"
                                      "v_cvt_pkrtz_f16_f32 %4, %4, %5\n v_cvt_pkrtz_f16_f32 %5, %5, %6\n v_cvt_pkrtz_f16_f32 %6, %6, %7\n v_cvt_pkrtz_f16_f32 %7, %7, %0\n"
                                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
                }
            else if (KIND == 18)  // v_lshlrev_b16 / v_lshrrev_b32 (the two halves of a packed pair of indices to byte addresses)
                {
                    REP8(asm volatile("v_lshlrev_b16_e32 %0, 2, %0\n v_lshrrev_b32_e32 %1, 14, %1\n v_lshlrev_b16_e32 %2, 2, %2\n v_lshrrev_b32_e32 %3, 14, %3\n"
                                      "v_lshlrev_b16_e32 %4, 2, %4\n v_lshrrev_b32_e32 %5, 14, %5\n v_lshlrev_b16_e32 %6, 2, %6\n v_lshrrev_b32_e32 %7, 14, %7\n"
                                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
                }
            else if (KIND == 19)  // v_lshlrev_b32_sdwa, word selects
                {
                    REP8(asm volatile("v_lshlrev_b32_sdwa %0, 2, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n v_lshlrev_b32_sdwa %1, 2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                                      "v_lshlrev_b32_sdwa %2, 2, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n v_lshlrev_b32_sdwa %3, 2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                                      "v_lshlrev_b32_sdwa %4, 2, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n v_lshlrev_b32_sdwa %5, 2, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                                      "v_lshlrev_b32_sdwa %6, 2, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n v_lshlrev_b32_sdwa %7, 2, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
                }
            else if (KIND == 20)  // the rounding-mode window: s_setreg, four v_pk_fma_f32, s_setreg, two v_add_f32 (8 wave-instructions counted: the VALU ones + 2)
                {
                    REP8(asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 2\n v_pk_fma_f32 %0, %0, %8, %0\n v_pk_fma_f32 %1, %1, %8, %1\n v_pk_fma_f32 %2, %2, %8, %2\n v_pk_fma_f32 %3, %3, %8, %3\n"
                                      "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0\n v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n"
                                      : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(cc));)
                }
            else if (KIND == 21)  // the same eight instructions without the two s_setreg (6 VALU, counted as 8 like KIND 20: the difference is what the mode switches cost)
                {
                    REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %0\n v_pk_fma_f32 %1, %1, %8, %1\n v_pk_fma_f32 %2, %2, %8, %2\n v_pk_fma_f32 %3, %3, %8, %3\n"
                                      "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n"
                                      : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(cc));)
                }
            else if (KIND == 22)  // v_floor_f32 / v_cvt_i32_f32 / v_cvt_u32_f32 / v_cvt_f16_f32
                {
                    REP8(asm volatile("v_floor_f32 %0, %0\n v_cvt_i32_f32 %1, %1\n v_cvt_u32_f32 %2, %2\n v_cvt_f16_f32 %3, %3\n"
                                      "v_floor_f32 %4, %4\n v_cvt_i32_f32 %5, %5\n v_cvt_u32_f32 %6, %6\n v_cvt_f16_f32 %7, %7\n"
                                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
                }
            else if (KIND == 23)  // v_floor_f32 alone
                {
                    REP8(asm volatile("v_floor_f32 %0, %0\n v_floor_f32 %1, %1\n v_floor_f32 %2, %2\n v_floor_f32 %3, %3\n"
                                      "v_floor_f32 %4, %4\n v_floor_f32 %5, %5\n v_floor_f32 %6, %6\n v_floor_f32 %7, %7\n"
                                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
                }
            else if (KIND == 24)  // v_cvt_u32_f32 alone
                {
                    REP8(asm volatile("v_cvt_u32_f32 %0, %0\n v_cvt_u32_f32 %1, %1\n v_cvt_u32_f32 %2, %2\n v_cvt_u32_f32 %3, %3\n"
                                      "v_cvt_u32_f32 %4, %4\n v_cvt_u32_f32 %5, %5\n v_cvt_u32_f32 %6, %6\n v_cvt_u32_f32 %7, %7\n"
                                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
                }
        }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}
template <int KIND>
void run(const char* name, int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = one per SIMD per block
    float* d;
    hipMalloc(&d, sizeof(float) * blocks * 256);
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<KIND><<<blocks, 256>>>(d, 10);
    hipEventRecord(e0);
    k<KIND><<<blocks, 256>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr_per_simd = (double)iters * (KIND == 16 ? 80.0 : 64.0) * waves_per_simd;
    printf("%-28s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instruction per SIMD (= %.2f cycles at 2.4 GHz)\n", name, waves_per_simd, ms,
        ms * 1e6 / wave_instr_per_simd, ms * 1e6 / wave_instr_per_simd * 2.4);
    hipFree(d);
}
int main()
{
    for (int w : {2, 6})
        {
            run<0>("v_fma_f32", w);
            run<1>("v_pk_fma_f32", w);
            run<3>("v_pk_add_f32", w);
            run<4>("v_add_f32", w);
            run<2>("v_xor_b32/v_mov_b32", w);
            run<5>("v_cvt_flr_i32/v_cvt_f32_i32", w);
            run<7>("v_cvt_flr_i32_f32", w);
            run<6>("v_fract_f32", w);
            run<8>("v_lshl_add_u32", w);
            run<9>("v_cmp_e64 + v_cndmask_e64", w);
            run<10>("v_cmp_e32 + v_addc_co_u32", w);
            run<11>("v_add_u32/v_and_b32", w);
            run<12>("v_pk_fma_f32 op_sel", w);
            run<13>("v_pk_add_f32 sgpr op_sel", w);
            run<14>("v_pk_fma_f32 3 distinct srcs", w);
            run<15>("v_fma_f32 3 distinct srcs", w);
            run<16>("trip mix (per 10: 4 pk_fma, 2 add, 2 cvt, 2 shift)", w);
            run<17>("v_cvt_pkrtz_f16_f32", w);
            run<18>("v_lshlrev_b16/v_lshrrev_b32", w);
            run<19>("v_lshlrev_b32_sdwa WORD_0/1", w);
            run<20>("2 s_setreg + 4 pk_fma + 2 pk_add (as 8)", w);
            run<21>("4 pk_fma + 2 pk_add (as 8)", w);
            run<22>("floor/cvt_i32/cvt_u32/cvt_f16 mix", w);
            run<23>("v_floor_f32", w);
            run<24>("v_cvt_u32_f32", w);
        }
    return 0;
}
