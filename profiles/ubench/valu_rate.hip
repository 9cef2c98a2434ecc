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
    const double wave_instr_per_simd = (double)iters * 64.0 * waves_per_simd;
    printf("%-28s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instruction per SIMD (= %.2f cycles at 2.4 GHz)\n", name, waves_per_simd, ms,
        ms * 1e6 / wave_instr_per_simd, ms * 1e6 / wave_instr_per_simd * 2.4);
    hipFree(d);
}
int main()
{
    for (int w : {1, 2, 4})
        {
            run<0>("v_fma_f32", w);
            run<1>("v_pk_fma_f32", w);
            run<3>("v_pk_add_f32", w);
            run<4>("v_add_f32", w);
            run<2>("v_xor_b32/v_mov_b32", w);
            run<5>("v_cvt_flr_i32/v_cvt_f32_i32", w);
        }
    return 0;
}
