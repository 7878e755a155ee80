// Instruction issue-rate microbenchmark (one wave, independent instructions): cycles per wave-instruction.
// hipcc --offload-arch=gfx950 -O3 rate.hip -o rate.bin && ./rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X X X X X X X X
#define BENCH(NAME, ASM, ...)                                                              \
    __global__ void k_##NAME(long long *out, double seed)                                  \
    {                                                                                      \
        double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3;                     \
        float f0 = (float)seed, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;                     \
        long long t0 = clock64();                                                          \
        for (int i = 0; i < 256; ++i) { REP8(asm volatile(ASM : __VA_ARGS__);) }           \
        long long t1 = clock64();                                                          \
        if (threadIdx.x == 0) out[0] = t1 - t0;                                            \
        if (a0 + a1 + a2 + a3 + f0 + f1 + f2 + f3 == 12345.0) out[1] = 1;                  \
    }
BENCH(add_f64, "v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed))
BENCH(mul_f64, "v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed))
BENCH(ldexp_f64, "v_ldexp_f64 %0, %0, -6\n v_ldexp_f64 %1, %1, -6\n v_ldexp_f64 %2, %2, -6\n v_ldexp_f64 %3, %3, -6", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed))
BENCH(cvt_f64_f32, "v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(f0), "v"(f1), "v"(f2), "v"(f3))
BENCH(cvt_f32_f64, "v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7", "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3))
BENCH(mul_f32, "v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4", "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(f0))
BENCH(pk_mul_f32, "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed))
BENCH(mov_dpp, "v_mov_b32_dpp %0, %4 wave_rol:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %1, %5 wave_rol:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %2, %6 wave_rol:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %3, %7 wave_rol:1 row_mask:0xf bank_mask:0xf bound_ctrl:1", "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(f0), "v"(f1), "v"(f2), "v"(f3))
BENCH(mov_dpp_row, "v_mov_b32_dpp %0, %4 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %1, %5 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %2, %6 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %3, %7 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1", "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(f0), "v"(f1), "v"(f2), "v"(f3))
BENCH(bpermute, "ds_bpermute_b32 %0, %4, %0\n ds_bpermute_b32 %1, %4, %1\n ds_bpermute_b32 %2, %4, %2\n ds_bpermute_b32 %3, %4, %3\n s_waitcnt lgkmcnt(0)", "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(threadIdx.x * 4))
BENCH(add_f64_dep, "v_add_f64 %0, %0, %4\n v_add_f64 %0, %0, %4\n v_add_f64 %0, %0, %4\n v_add_f64 %0, %0, %4", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed))
BENCH(cndmask, "v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc", "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(f0) : "vcc")

// the 8-bit cost / re-quantisation path
#define BENCH_U(NAME, ASM)                                                                 \
    __global__ void k_##NAME(long long *out, double seed)                                  \
    {                                                                                      \
        unsigned u0 = (unsigned)seed + threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3; \
        float f0 = (float)seed, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;                     \
        long long t0 = clock64();                                                          \
        for (int i = 0; i < 256; ++i) { REP8(asm volatile(ASM : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));) } \
        long long t1 = clock64();                                                          \
        if (threadIdx.x == 0) out[0] = t1 - t0;                                            \
        if (u0 + u1 + u2 + u3 + f0 + f1 + f2 + f3 == 12345.0f) out[1] = 1;                 \
    }
BENCH_U(sad_u8, "v_sad_u8 %0, %0, %1, 0\n v_sad_u8 %1, %1, %2, 0\n v_sad_u8 %2, %2, %3, 0\n v_sad_u8 %3, %3, %0, 0")
BENCH_U(mul_u24, "v_mul_u32_u24 %0, %0, %1\n v_mul_u32_u24 %1, %1, %2\n v_mul_u32_u24 %2, %2, %3\n v_mul_u32_u24 %3, %3, %0")
BENCH_U(cvt_f32_u32, "v_cvt_f32_u32 %4, %0\n v_cvt_f32_u32 %5, %1\n v_cvt_f32_u32 %6, %2\n v_cvt_f32_u32 %7, %3")
BENCH_U(cvt_f32_ubyte0, "v_cvt_f32_ubyte0 %4, %0\n v_cvt_f32_ubyte0 %5, %1\n v_cvt_f32_ubyte0 %6, %2\n v_cvt_f32_ubyte0 %7, %3")
BENCH_U(trunc_f32, "v_trunc_f32 %4, %4\n v_trunc_f32 %5, %5\n v_trunc_f32 %6, %6\n v_trunc_f32 %7, %7")
BENCH_U(rndne_f32, "v_rndne_f32 %4, %4\n v_rndne_f32 %5, %5\n v_rndne_f32 %6, %6\n v_rndne_f32 %7, %7")
BENCH_U(med3_f32, "v_med3_f32 %4, %4, %5, %6\n v_med3_f32 %5, %5, %6, %7\n v_med3_f32 %6, %6, %7, %4\n v_med3_f32 %7, %7, %4, %5")
BENCH_U(and_b32, "v_and_b32 %0, %0, %1\n v_and_b32 %1, %1, %2\n v_and_b32 %2, %2, %3\n v_and_b32 %3, %3, %0")
BENCH_U(lshrrev_b32, "v_lshrrev_b32 %0, 16, %0\n v_lshrrev_b32 %1, 16, %1\n v_lshrrev_b32 %2, 16, %2\n v_lshrrev_b32 %3, 16, %3")
BENCH_U(sub_u32, "v_sub_u32 %0, %0, %1\n v_sub_u32 %1, %1, %2\n v_sub_u32 %2, %2, %3\n v_sub_u32 %3, %3, %0")
BENCH_U(fma_f32, "v_fma_f32 %4, %4, %5, %6\n v_fma_f32 %5, %5, %6, %7\n v_fma_f32 %6, %6, %7, %4\n v_fma_f32 %7, %7, %4, %5")

int main()
{
    long long *d, h[2];
    hipMalloc(&d, 16);
#define RUN(NAME, WAVES)                                                                        \
    for (int w = 1; w <= WAVES; w *= 2) {                                                        \
        hipLaunchKernelGGL(k_##NAME, dim3(1), dim3(64 * w), 0, 0, d, 1.0);                       \
        hipLaunchKernelGGL(k_##NAME, dim3(1), dim3(64 * w), 0, 0, d, 1.0);                       \
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);                                              \
        printf("%-14s waves/CU %2d: %.2f clk per wave-instruction (clock64 ticks)\n", #NAME, w, (double)h[0] / (256.0 * 32)); \
    }
    RUN(add_f64, 8) RUN(mul_f64, 4) RUN(ldexp_f64, 4) RUN(cvt_f64_f32, 4) RUN(cvt_f32_f64, 4) RUN(mul_f32, 4) RUN(pk_mul_f32, 4)
    RUN(mov_dpp, 4) RUN(mov_dpp_row, 4) RUN(bpermute, 4) RUN(add_f64_dep, 4) RUN(cndmask, 4)
    RUN(sad_u8, 4) RUN(mul_u24, 4) RUN(cvt_f32_u32, 4) RUN(cvt_f32_ubyte0, 4) RUN(trunc_f32, 4) RUN(rndne_f32, 4) RUN(med3_f32, 4)
    RUN(and_b32, 4) RUN(lshrrev_b32, 4) RUN(sub_u32, 4) RUN(fma_f32, 4)
    return 0;
}
