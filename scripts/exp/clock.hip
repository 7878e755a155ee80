// Sustained issue rate of the whole chip per instruction type: wall time of a launch that fills every SIMD with W
// waves of N independent instructions -> effective "SIMD clock" = 4 * N * W / t if one wave-instruction takes 4 cycles.
// hipcc --offload-arch=gfx950 -O3 clock.hip -o clock.bin && ./clock.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(X) X X X X X X X X X X X X X X X X
#define KERNEL(NAME, ASM, ...)                                                             \
    __global__ __launch_bounds__(256) void k_##NAME(float *out, double seed, int iters)     \
    {                                                                                      \
        double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3;                     \
        float f0 = (float)seed, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;                     \
        for (int i = 0; i < iters; ++i) { REP16(asm volatile(ASM : __VA_ARGS__);) }         \
        if (a0 + a1 + a2 + a3 + f0 + f1 + f2 + f3 == 12345.0) out[0] = 1;                  \
    }
KERNEL(add_f64, "v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed))
KERNEL(mul_f32, "v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4", "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(f0))
KERNEL(pk_mul_f32, "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed))
KERNEL(mov_dpp, "v_mov_b32_dpp %0, %4 wave_rol:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %1, %5 wave_rol:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %2, %6 wave_rol:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %3, %7 wave_rol:1 row_mask:0xf bank_mask:0xf bound_ctrl:1", "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(f0), "v"(f1), "v"(f2), "v"(f3))
KERNEL(cvt_f64_f32, "v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(f0), "v"(f1), "v"(f2), "v"(f3))
KERNEL(mix, "v_add_f64 %0, %0, %4\n v_mov_b32_dpp %2, %3 wave_rol:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f64 %1, %1, %4\n v_mul_f32 %3, %3, %3", "+v"(a0), "+v"(a1), "+v"(f0), "+v"(f1) : "v"(seed))

// the 8-bit cost / re-quantisation path (integer registers)
#define KERNEL_U(NAME, ASM)                                                                \
    __global__ __launch_bounds__(256) void k_##NAME(float *out, double seed, int iters)     \
    {                                                                                      \
        unsigned u0 = (unsigned)seed + threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3; \
        float f0 = (float)seed, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;                     \
        for (int i = 0; i < iters; ++i) { REP16(asm volatile(ASM : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));) } \
        if (u0 + u1 + u2 + u3 + f0 + f1 + f2 + f3 == 12345.0f) out[0] = 1;                 \
    }
KERNEL_U(sad_u8, "v_sad_u8 %0, %0, %1, 0\n v_sad_u8 %1, %1, %2, 0\n v_sad_u8 %2, %2, %3, 0\n v_sad_u8 %3, %3, %0, 0")
KERNEL_U(mul_u24, "v_mul_u32_u24 %0, %0, %1\n v_mul_u32_u24 %1, %1, %2\n v_mul_u32_u24 %2, %2, %3\n v_mul_u32_u24 %3, %3, %0")
KERNEL_U(cvt_f32_u32, "v_cvt_f32_u32 %4, %0\n v_cvt_f32_u32 %5, %1\n v_cvt_f32_u32 %6, %2\n v_cvt_f32_u32 %7, %3")
KERNEL_U(trunc_f32, "v_trunc_f32 %4, %4\n v_trunc_f32 %5, %5\n v_trunc_f32 %6, %6\n v_trunc_f32 %7, %7")
KERNEL_U(rndne_f32, "v_rndne_f32 %4, %4\n v_rndne_f32 %5, %5\n v_rndne_f32 %6, %6\n v_rndne_f32 %7, %7")
KERNEL_U(med3_f32, "v_med3_f32 %4, %4, %5, %6\n v_med3_f32 %5, %5, %6, %7\n v_med3_f32 %6, %6, %7, %4\n v_med3_f32 %7, %7, %4, %5")
KERNEL_U(and_b32, "v_and_b32 %0, %0, %1\n v_and_b32 %1, %1, %2\n v_and_b32 %2, %2, %3\n v_and_b32 %3, %3, %0")
KERNEL_U(lshrrev_b32, "v_lshrrev_b32 %0, 16, %0\n v_lshrrev_b32 %1, 16, %1\n v_lshrrev_b32 %2, 16, %2\n v_lshrrev_b32 %3, 16, %3")
KERNEL_U(sub_u32, "v_sub_u32 %0, %0, %1\n v_sub_u32 %1, %1, %2\n v_sub_u32 %2, %2, %3\n v_sub_u32 %3, %3, %0")
KERNEL_U(fma_f32, "v_fma_f32 %4, %4, %5, %6\n v_fma_f32 %5, %5, %6, %7\n v_fma_f32 %6, %6, %7, %4\n v_fma_f32 %7, %7, %4, %5")
KERNEL_U(add_f32, "v_add_f32 %4, %4, %5\n v_add_f32 %5, %5, %6\n v_add_f32 %6, %6, %7\n v_add_f32 %7, %7, %4")
KERNEL_U(cmp_cnd, "v_cmp_lt_f32 vcc, %4, %5\n v_cndmask_b32 %6, %6, %7, vcc\n v_cmp_lt_f32 vcc, %5, %4\n v_cndmask_b32 %7, %7, %6, vcc")

int main()
{
    float *d;
    hipMalloc(&d, 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096;   // x 64 instructions
#define RUN(NAME, WPS)                                                                                  \
    {                                                                                                  \
        const int blocks = 256 * (WPS);   /* 256 CUs x WPS blocks of 4 waves = WPS waves per SIMD */    \
        hipLaunchKernelGGL(k_##NAME, dim3(blocks), dim3(256), 0, 0, d, 1.0, 64);                        \
        hipEventRecord(e0);                                                                            \
        hipLaunchKernelGGL(k_##NAME, dim3(blocks), dim3(256), 0, 0, d, 1.0, iters);                     \
        hipEventRecord(e1); hipEventSynchronize(e1);                                                    \
        float ms; hipEventElapsedTime(&ms, e0, e1);                                                     \
        const double instr_per_simd = (double)iters * 64 * (WPS);                                       \
        printf("%-12s %d waves/SIMD: %.3f ms, %.3f G wave-instr/s per SIMD -> %.2f GHz x 4 clk\n", #NAME, WPS, ms, instr_per_simd / ms * 1e-6, instr_per_simd * 4 / ms * 1e-6); \
    }
    RUN(add_f64, 1) RUN(add_f64, 2) RUN(add_f64, 4) RUN(mul_f32, 1) RUN(mul_f32, 2) RUN(mul_f32, 4) RUN(pk_mul_f32, 2) RUN(mov_dpp, 2) RUN(cvt_f64_f32, 2)
    RUN(mix, 2) RUN(mix, 3)
    RUN(sad_u8, 4) RUN(mul_u24, 4) RUN(cvt_f32_u32, 4) RUN(trunc_f32, 4) RUN(rndne_f32, 4) RUN(med3_f32, 4) RUN(and_b32, 4) RUN(lshrrev_b32, 4)
    RUN(sub_u32, 4) RUN(fma_f32, 4) RUN(add_f32, 4) RUN(cmp_cnd, 4)
    return 0;
}
