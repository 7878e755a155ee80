// Debug aid: the weighted median's weights of ONE window on the device, bit patterns to stdout (scripts/dbg_wmf_pixel.py).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -I primestereomatch_amd/csrc scripts/exp/wmw.hip -o scripts/exp/wmw.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
#define PSM_EXP_FN __device__ __forceinline__
#define PSM_EXP_FMA(a, b, c) __fma_rn(a, b, c)
#include "psm_exp.h"
__device__ const unsigned long long wm_exp_tab[256] = PSM_EXP_TAB_INIT;
template <bool RIGHT>
__device__ float wm_weight(float4 p, float4 q, int wx, int wy, double *argout)
{
    float disWgt = (float)(wx * wx + wy * wy);
    const float d0 = __fsub_rn(p.x, q.x), d1 = __fsub_rn(p.y, q.y), d2 = __fsub_rn(p.z, q.z);
    float clrWgt = __fadd_rn(__fadd_rn(__fmul_rn(d0, d0), __fmul_rn(d1, d1)), __fmul_rn(d2, d2));
    if (RIGHT) { disWgt = __fsqrt_rn(disWgt); clrWgt = __fsqrt_rn(clrWgt); }
    const double arg = __dsub_rn((double)__fdiv_rn(-disWgt, 81.0f), __ddiv_rn((double)clrWgt, 0.1 * 0.1));
    *argout = arg;
    return (float)psm_exp_nonpos(arg, wm_exp_tab);
}
__global__ void k(const float *pq, float *w, double *arg, int right)
{
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    if (t >= 361) return;
    const float4 p = make_float4(pq[0], pq[1], pq[2], 0), q = make_float4(pq[3 + 3 * t], pq[4 + 3 * t], pq[5 + 3 * t], 0);
    const int wy = t / 19 - 9, wx = t % 19 - 9;
    w[t] = right ? wm_weight<true>(p, q, wx, wy, &arg[t]) : wm_weight<false>(p, q, wx, wy, &arg[t]);
}
int main(int argc, char **argv)
{
    std::vector<float> pq(3 + 3 * 361);
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(pq.data(), 4, pq.size(), f) != pq.size()) return 1;
    fclose(f);
    const int right = argc > 2 ? atoi(argv[2]) : 1;
    float *dpq, *dw; double *da;
    hipMalloc(&dpq, pq.size() * 4); hipMalloc(&dw, 361 * 4); hipMalloc(&da, 361 * 8);
    hipMemcpy(dpq, pq.data(), pq.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(6), dim3(64), 0, 0, dpq, dw, da, right);
    std::vector<float> w(361); std::vector<double> a(361);
    hipMemcpy(w.data(), dw, 361 * 4, hipMemcpyDeviceToHost); hipMemcpy(a.data(), da, 361 * 8, hipMemcpyDeviceToHost);
    for (int t = 0; t < 361; ++t) { unsigned u; unsigned long long v; memcpy(&u, &w[t], 4); memcpy(&v, &a[t], 8); printf("%08x %016llx\n", u, v); }
    return 0;
}
