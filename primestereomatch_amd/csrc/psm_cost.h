// psm_cost.h - the matching cost of one pixel pair (device code shared by psm_kernels.hip and psm_fgf.hip)
#pragma once
#include <hip/hip_runtime.h>

namespace psm {

__device__ __forceinline__ float cost_pair(float4 a, float4 b)
{  // myCostGrd(lC, rC, lG, rG), src/CVC.cpp:18-27
    float clr = __fadd_rn(__fadd_rn(fabsf(__fsub_rn(a.x, b.x)), fabsf(__fsub_rn(a.y, b.y))), fabsf(__fsub_rn(a.z, b.z)));
    float grd = fabsf(__fsub_rn(a.w, b.w));
    return __fadd_rn(__fmul_rn(0.9f, clr), __fmul_rn(__fsub_rn(1.0f, 0.9f), grd));
}
__device__ __forceinline__ float cost_border(float4 a)
{  // myCostGrd(lC, lG), src/CVC.cpp:30-39: BC_32F is the double 1.0 -> double differences/sum
    double s = __dadd_rn(__dadd_rn(fabs(__dsub_rn((double)a.x, 1.0)), fabs(__dsub_rn((double)a.y, 1.0))),
                         fabs(__dsub_rn((double)a.z, 1.0)));
    float clr = (float)s;
    float grd = (float)fabs(__dsub_rn((double)a.w, 1.0));
    return __fadd_rn(__fmul_rn(0.9f, clr), __fmul_rn(__fsub_rn(1.0f, 0.9f), grd));
}

}  // namespace psm
