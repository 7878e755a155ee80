// psm_dev.h - device helpers shared by the marching kernels (psm_kernels.hip, psm_pc.hip): REFLECT_101 indexing,
// cross-lane exchanges, the sliding fp64 trees of the 8-tap box filter, the per-voxel model solve.
#pragma once
#include <hip/hip_runtime.h>

namespace psm {

// ------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int r101(int k, int n)
{
    k = k < 0 ? -k : k;
    k = k >= n ? 2 * (n - 1) - k : k;
    return k;
}
__device__ __forceinline__ int r101c(int k, int n)
{  // reflect, then clamp (only matters for the unused overshoot rows/lanes)
    k = r101(k, n);
    return k < 0 ? 0 : (k > n - 1 ? n - 1 : k);
}

__device__ __forceinline__ double t8(double t0, double t1, double t2, double t3, double t4, double t5,
                                     double t6, double t7)
{
    return __dadd_rn(__dadd_rn(__dadd_rn(t0, t1), __dadd_rn(t2, t3)),
                     __dadd_rn(__dadd_rn(t4, t5), __dadd_rn(t6, t7)));
}
__device__ __forceinline__ float box_out(double s) { return (float)(s * 0.015625); }

// gray of CVC::preprocess: cvtColor(CV_RGB2GRAY) on B,G,R data -> 0.299 multiplies c0
__device__ __forceinline__ float gray_of(float c0, float c1, float c2)
{
    return __fadd_rn(__fadd_rn(__fmul_rn(c0, 0.299f), __fmul_rn(c1, 0.587f)), __fmul_rn(c2, 0.114f));
}

// cross-lane gather: lane l receives the value of lane (byte_idx/4)
__device__ __forceinline__ float lane_get(float v, int byte_idx)
{
    return __int_as_float(__builtin_amdgcn_ds_bpermute(byte_idx, __float_as_int(v)));
}
__device__ __forceinline__ double lane_get(double v, int byte_idx)
{
    int lo = __builtin_amdgcn_ds_bpermute(byte_idx, __double2loint(v));
    int hi = __builtin_amdgcn_ds_bpermute(byte_idx, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

// One-lane rotation of the whole wave in the VALU (DPP wave_rol:1: lane l <- lane l+1, lane 63 <-
// lane 0; verified on gfx950).  No LDS round trip, unlike ds_bpermute.
// bound_ctrl=1: every lane has a source under wave_rol, and it spares the compiler a v_mov to seed `old`
__device__ __forceinline__ int rol1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x134, 0xf, 0xf, true); }
__device__ __forceinline__ float rol1(float v) { return __int_as_float(rol1(__float_as_int(v))); }
__device__ __forceinline__ double rol2(double v)
{
    int lo = rol1(rol1(__double2loint(v))), hi = rol1(rol1(__double2hiint(v)));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rol4(double v)
{
    int lo = rol1(rol1(rol1(rol1(__double2loint(v))))), hi = rol1(rol1(rol1(rol1(__double2hiint(v)))));
    return __hiloint2double(hi, lo);
}

// Horizontal 8-tap window sum over lanes l..l+7 (sliding balanced tree).
// PSM_XLANE_MODE: which exchange levels use DPP rotations instead of ds_bpermute - 0: none, 1: distance 1, 2: distances 1
// and 2, 3: all (1, 2, 4).  Measured on the fused filter at 1080p x 256: 5.29 / 4.76 / 4.69 / 5.98 ms per volume; 2 it is.
constexpr int PSM_XLANE_MODE = 2;
// L1F32 (the tolerance form, PSM_FLAG_F32_TOL): level 1 of the tree on the fp32 inputs - one DPP move, one fp32 add and one
// conversion instead of two conversions, a DPP move and an fp64 add.  One more fp32 rounding per pair of taps: not the oracle's
// bits, max |dq| 1e-5 on Cones / Teddy and 4e-5 on the synthetic pairs, no WTA pixel changed (oracle model PSMO_VAR_F32_L1).
template <bool L1F32 = false>
__device__ __forceinline__ double hsum8(float v, int i1, int i2, int i4)
{
    double s2 = L1F32 ? (double)__fadd_rn(v, rol1(v)) : __dadd_rn((double)v, (double)(PSM_XLANE_MODE >= 1 ? rol1(v) : lane_get(v, i1)));
    double s4 = __dadd_rn(s2, PSM_XLANE_MODE >= 2 ? rol2(s2) : lane_get(s2, i2));
    return __dadd_rn(s4, PSM_XLANE_MODE >= 3 ? rol4(s4) : lane_get(s4, i4));
}

// Vertical 8-tap sliding tree.  After feeding row yy, returns the window sum of rows yy-7..yy.
struct VTree {
    double hp;
    double s2[2];
    double s4[4];
};
template <int K>
__device__ __forceinline__ double vstep(VTree &t, double hs)
{
    double n2 = __dadd_rn(t.hp, hs);         // hs[yy-1] + hs[yy]
    double n4 = __dadd_rn(t.s2[K & 1], n2);  // s2[yy-3] + s2[yy-1]
    double n8 = __dadd_rn(t.s4[K & 3], n4);  // s4[yy-7] + s4[yy-3]
    t.s2[K & 1] = n2;
    t.s4[K & 3] = n4;
    t.hp = hs;
    return n8;
}

// The per-voxel linear-model solve of GuidedFilter_cv (src/CVF.cpp:91-155) with the d-invariant
// adjugate entries and 1/DET taken from the guidance planes.
// FMA (PSM_FLAG_FMA_SOLVE): the reading of src/CVF.cpp:129-147 a compiler with -ffp-contract=fast gives on an FMA target (GCC's
// default; the ARM boards the reference ran on) - of the three products of an accumulation GCC rounds the MIDDLE one and fuses
// the other two: fma(c2, A2, fma(c0, A0, RN(c1*A1))) (oracle: PSMO_VAR_FMA_SOLVE, pinned against a live gcc -O2 -mfma compile by
// tests/test_oracle.py; the minors and 1/DET in the guidance planes are then the fused forms too, k_guide_march).  The covariance
// and the b line are cv::Mat passes in the reference (src/CVF.cpp:91-95,152-155) - separate multiply and subtract either way.
template <bool FMA = false>
__device__ __forceinline__ float4 solve_ab(float mp, float mIp0, float mIp1, float mIp2, float4 g2,
                                           float4 g3, float2 g4)
{
    const float mI0 = g2.x, mI1 = g2.y, mI2 = g2.z, inv = g2.w;
    const float A00 = g3.x, A01 = g3.y, A02 = g3.z, A11 = g3.w, A12 = g4.x, A22 = g4.y;
    float c0 = __fsub_rn(mIp0, __fmul_rn(mI0, mp));
    float c1 = __fsub_rn(mIp1, __fmul_rn(mI1, mp));
    float c2 = __fsub_rn(mIp2, __fmul_rn(mI2, mp));
    float a0, a1, a2;
    if (FMA) {
        a0 = __fmul_rn(inv, __fmaf_rn(c2, A02, __fmaf_rn(c0, A00, __fmul_rn(c1, A01))));
        a1 = __fmul_rn(inv, __fmaf_rn(c2, A12, __fmaf_rn(c0, A01, __fmul_rn(c1, A11))));
        a2 = __fmul_rn(inv, __fmaf_rn(c2, A22, __fmaf_rn(c0, A02, __fmul_rn(c1, A12))));
    } else {
        a0 = __fmul_rn(inv, __fadd_rn(__fadd_rn(__fmul_rn(c0, A00), __fmul_rn(c1, A01)), __fmul_rn(c2, A02)));
        a1 = __fmul_rn(inv, __fadd_rn(__fadd_rn(__fmul_rn(c0, A01), __fmul_rn(c1, A11)), __fmul_rn(c2, A12)));
        a2 = __fmul_rn(inv, __fadd_rn(__fadd_rn(__fmul_rn(c0, A02), __fmul_rn(c1, A12)), __fmul_rn(c2, A22)));
    }
    float b = __fsub_rn(__fsub_rn(__fsub_rn(mp, __fmul_rn(a0, mI0)), __fmul_rn(a1, mI1)), __fmul_rn(a2, mI2));
    return make_float4(a0, a1, a2, b);
}
// q = ((box(b) + box(a0)*I0) + box(a1)*I1) + box(a2)*I2   (src/CVF.cpp:157-163)
__device__ __forceinline__ float recombine(float ma0, float ma1, float ma2, float mb, float4 g1)
{
    return __fadd_rn(__fadd_rn(__fadd_rn(mb, __fmul_rn(ma0, g1.x)), __fmul_rn(ma1, g1.y)), __fmul_rn(ma2, g1.z));
}

// WTA exchange key: (monotone-uint32(cost) << 32 | d), signed-comparable; the minimum over candidates reproduces
// strict-< / lowest-d-wins (src/DispSel.cpp:96-104) across slices, shards and ranks
__device__ __forceinline__ long long pack_key_f32(float cost, int d)
{
    cost = __fadd_rn(cost, 0.0f);  // -0 -> +0 so that equal costs compare equal
    // monotone map float -> signed int: non-negative floats keep their bits, negative ones get their magnitude bits flipped
    // (the same 64 bits as ((b < 0 ? ~b : b | 0x80000000) << 32 | d) ^ (1 << 63), in three integer ops instead of five)
    const int b = __float_as_int(cost);
    const int hi = b ^ ((b >> 31) & 0x7fffffff);
    return (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)d);
}

}  // namespace psm
