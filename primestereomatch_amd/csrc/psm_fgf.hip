// psm_fgf.hip - Fast Guided Filter variant of the cost aggregation ("next" row of SURVEY.md 8f):
// DispEst::CostFilter_FGF (src/DispEst.cpp:281-296) -> FastGuidedFilterColor (src/fastguidedfilter.cpp:
// 124-209) on the device.  The guided filter runs on images subsampled by s (nearest neighbour), its
// box filter shrinks to k = 2*(8/s)+1 taps, and the two smoothed model planes are upsampled bilinearly
// before the linear model is applied at full resolution.
//
// Canonical OpenCV semantics (oracle/psm_oracle.h "CVF, Fast Guided Filter variant"):
//   INTER_NN:      src index = min(floor(x * ifx), size-1), ifx = 1 / ((size/s) / (double)size)
//   cv::blur k x k: taps -k/2..+k/2, REFLECT_101, fp64 sums (x taps left to right, then y taps top to
//                  bottom), (float)(sum * (1.0/(k*k)))
//   INTER_LINEAR:  fx = (float)((dx+0.5)*scale-0.5); sx = floor(fx); fx -= sx; clamped at both ends;
//                  row pass S[sx]*(1-fx) + S[sx+1]*fx, then column pass, fp32, no FMA
// Work split: the subsampled planes are 1/s^2 of the pixels, so the three small-image kernels below are
// plain per-pixel (direct) kernels; the only full-resolution kernel is the upsample + linear model, which
// reads each filtered slice nowhere and writes it once (4 B/voxel).
#include "psm_kernels.h"

namespace psm {

namespace {

__device__ __forceinline__ int r101s(int k, int n)
{
    k = k < 0 ? -k : k;
    k = k >= n ? 2 * (n - 1) - k : k;
    return k < 0 ? 0 : (k > n - 1 ? n - 1 : k);
}
__device__ __forceinline__ int nn_src(int x, int dsize, int ssize)
{   // cv::resize INTER_NN source index
    const double ifx = 1. / ((double)dsize / ssize);
    int sx = (int)floor(x * ifx);
    return sx < ssize - 1 ? sx : ssize - 1;
}
__device__ __forceinline__ void lin_src(int d, int ssize, int dsize, int &sx, float &f)
{   // cv::resize INTER_LINEAR source index and weight of the second tap
    const double scale = (double)ssize / dsize;
    f = (float)((d + 0.5) * scale - 0.5);
    sx = (int)floorf(f);
    f = __fsub_rn(f, (float)sx);
    if (sx < 0) { f = 0.f; sx = 0; }
    if (sx >= ssize - 1) { f = 0.f; sx = ssize - 1; }
}

// ---- setup: subsampled guidance, its k x k means, the inverse of (Sigma + eps I) -------------------
// ism[y][x] = {I0,I1,I2,0} (subsampled), msm = {mean0,mean1,mean2,0}, v1 = {irr,irg,irb,igg}, v2 = {igb,ibb}
__global__ __launch_bounds__(256) void k_fgf_small_img(const float4 *__restrict__ g1, int W, int H, int ws, int hs,
                                                      float4 *__restrict__ ism)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= ws) return;
    float4 g = g1[(size_t)nn_src(y, hs, H) * W + nn_src(x, ws, W)];
    ism[(size_t)y * ws + x] = make_float4(g.x, g.y, g.z, 0.f);
}

template <int K>
__global__ __launch_bounds__(256) void k_fgf_setup(const float4 *__restrict__ ism, int ws, int hs, float4 *__restrict__ msm,
                                                  float4 *__restrict__ v1, float2 *__restrict__ v2)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= ws) return;
    constexpr int R = K / 2;
    const double scale = 1.0 / (K * K);
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = -R; j <= R; ++j) {
        const float4 *row = ism + (size_t)r101s(y + j, hs) * ws;
        double h[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int i = -R; i <= R; ++i) {
            float4 g = row[r101s(x + i, ws)];
            h[0] = __dadd_rn(h[0], (double)g.x);
            h[1] = __dadd_rn(h[1], (double)g.y);
            h[2] = __dadd_rn(h[2], (double)g.z);
            h[3] = __dadd_rn(h[3], (double)__fmul_rn(g.x, g.x));
            h[4] = __dadd_rn(h[4], (double)__fmul_rn(g.x, g.y));
            h[5] = __dadd_rn(h[5], (double)__fmul_rn(g.x, g.z));
            h[6] = __dadd_rn(h[6], (double)__fmul_rn(g.y, g.y));
            h[7] = __dadd_rn(h[7], (double)__fmul_rn(g.y, g.z));
            h[8] = __dadd_rn(h[8], (double)__fmul_rn(g.z, g.z));
        }
#pragma unroll
        for (int c = 0; c < 9; ++c) acc[c] = __dadd_rn(acc[c], h[c]);
    }
    float m[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) m[c] = (float)(acc[c] * scale);
    const float eps = 0.0001f;  // GIF_EPS; the double eps is added to CV_32F planes in float
    // src/fastguidedfilter.cpp:150-155
    float rr = __fadd_rn(__fsub_rn(m[3], __fmul_rn(m[0], m[0])), eps);
    float rg = __fsub_rn(m[4], __fmul_rn(m[0], m[1]));
    float rb = __fsub_rn(m[5], __fmul_rn(m[0], m[2]));
    float gg = __fadd_rn(__fsub_rn(m[6], __fmul_rn(m[1], m[1])), eps);
    float gb = __fsub_rn(m[7], __fmul_rn(m[1], m[2]));
    float bb = __fadd_rn(__fsub_rn(m[8], __fmul_rn(m[2], m[2])), eps);
    // src/fastguidedfilter.cpp:158-172
    float irr = __fsub_rn(__fmul_rn(gg, bb), __fmul_rn(gb, gb));
    float irg = __fsub_rn(__fmul_rn(gb, rb), __fmul_rn(rg, bb));
    float irb = __fsub_rn(__fmul_rn(rg, gb), __fmul_rn(gg, rb));
    float igg = __fsub_rn(__fmul_rn(rr, bb), __fmul_rn(rb, rb));
    float igb = __fsub_rn(__fmul_rn(rb, rg), __fmul_rn(rr, gb));
    float ibb = __fsub_rn(__fmul_rn(rr, gg), __fmul_rn(rg, rg));
    float det = __fadd_rn(__fadd_rn(__fmul_rn(irr, rr), __fmul_rn(irg, rg)), __fmul_rn(irb, rb));
    size_t o = (size_t)y * ws + x;
    msm[o] = make_float4(m[0], m[1], m[2], 0.f);
    v1[o] = make_float4(__fdiv_rn(irr, det), __fdiv_rn(irg, det), __fdiv_rn(irb, det), __fdiv_rn(igg, det));
    v2[o] = make_float2(__fdiv_rn(igb, det), __fdiv_rn(ibb, det));
}

// ---- per slice, small grid: means of p and I*p -> linear model (a_r,a_g,a_b,b) ---------------------
template <int K>
__global__ __launch_bounds__(256) void k_fgf_model(const float *__restrict__ vol, int W, int H, int ws, int hs,
                                                  const float4 *__restrict__ ism, const float4 *__restrict__ msm,
                                                  const float4 *__restrict__ v1, const float2 *__restrict__ v2,
                                                  float4 *__restrict__ ab)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, d = blockIdx.z;
    if (x >= ws) return;
    constexpr int R = K / 2;
    const double scale = 1.0 / (K * K);
    const float *vd = vol + (size_t)d * H * W;
    int sxs[K];
#pragma unroll
    for (int i = 0; i < K; ++i) sxs[i] = r101s(x - R + i, ws);
    double acc[4] = {0, 0, 0, 0};
    for (int j = -R; j <= R; ++j) {
        const int ys = r101s(y + j, hs);
        const float *prow = vd + (size_t)nn_src(ys, hs, H) * W;
        const float4 *irow = ism + (size_t)ys * ws;
        double h[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const float p = prow[nn_src(sxs[i], ws, W)];
            const float4 g = irow[sxs[i]];
            h[0] = __dadd_rn(h[0], (double)p);
            h[1] = __dadd_rn(h[1], (double)__fmul_rn(g.x, p));
            h[2] = __dadd_rn(h[2], (double)__fmul_rn(g.y, p));
            h[3] = __dadd_rn(h[3], (double)__fmul_rn(g.z, p));
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = __dadd_rn(acc[c], h[c]);
    }
    const float mp = (float)(acc[0] * scale), mr = (float)(acc[1] * scale), mg = (float)(acc[2] * scale),
                mb = (float)(acc[3] * scale);
    const size_t o = (size_t)y * ws + x;
    const float4 m = msm[o], a = v1[o];
    const float2 b2 = v2[o];
    // src/fastguidedfilter.cpp:184-194
    float cr = __fsub_rn(mr, __fmul_rn(m.x, mp));
    float cg = __fsub_rn(mg, __fmul_rn(m.y, mp));
    float cb = __fsub_rn(mb, __fmul_rn(m.z, mp));
    float ar = __fadd_rn(__fadd_rn(__fmul_rn(a.x, cr), __fmul_rn(a.y, cg)), __fmul_rn(a.z, cb));
    float ag = __fadd_rn(__fadd_rn(__fmul_rn(a.y, cr), __fmul_rn(a.w, cg)), __fmul_rn(b2.x, cb));
    float abl = __fadd_rn(__fadd_rn(__fmul_rn(a.z, cr), __fmul_rn(b2.x, cg)), __fmul_rn(b2.y, cb));
    float bq = __fsub_rn(__fsub_rn(__fsub_rn(mp, __fmul_rn(ar, m.x)), __fmul_rn(ag, m.y)), __fmul_rn(abl, m.z));
    ab[(size_t)d * hs * ws + o] = make_float4(ar, ag, abl, bq);
}

// ---- per slice, small grid: k x k means of the model planes ----------------------------------------
template <int K>
__global__ __launch_bounds__(256) void k_fgf_smooth(const float4 *__restrict__ ab, int ws, int hs, float4 *__restrict__ mab)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, d = blockIdx.z;
    if (x >= ws) return;
    constexpr int R = K / 2;
    const double scale = 1.0 / (K * K);
    const float4 *ad = ab + (size_t)d * hs * ws;
    double acc[4] = {0, 0, 0, 0};
    for (int j = -R; j <= R; ++j) {
        const float4 *row = ad + (size_t)r101s(y + j, hs) * ws;
        double h[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = -R; i <= R; ++i) {
            const float4 v = row[r101s(x + i, ws)];
            h[0] = __dadd_rn(h[0], (double)v.x);
            h[1] = __dadd_rn(h[1], (double)v.y);
            h[2] = __dadd_rn(h[2], (double)v.z);
            h[3] = __dadd_rn(h[3], (double)v.w);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = __dadd_rn(acc[c], h[c]);
    }
    mab[(size_t)d * hs * ws + (size_t)y * ws + x] =
        make_float4((float)(acc[0] * scale), (float)(acc[1] * scale), (float)(acc[2] * scale), (float)(acc[3] * scale));
}

// ---- full resolution: bilinear upsampling of the four smoothed planes + the linear model -----------
__global__ __launch_bounds__(256) void k_fgf_apply(const float4 *__restrict__ mab, int ws, int hs, const float4 *__restrict__ g1,
                                                  int W, int H, float *__restrict__ vol)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, d = blockIdx.z;
    if (x >= W) return;
    int sx, sy;
    float fx, fy;
    lin_src(x, ws, W, sx, fx);
    lin_src(y, hs, H, sy, fy);
    const int sx1 = sx + 1 < ws ? sx + 1 : ws - 1, sy1 = sy + 1 < hs ? sy + 1 : hs - 1;
    const float a0 = __fsub_rn(1.f, fx), a1 = fx, b0 = __fsub_rn(1.f, fy), b1 = fy;
    const float4 *md = mab + (size_t)d * hs * ws;
    const float4 p00 = md[(size_t)sy * ws + sx], p01 = md[(size_t)sy * ws + sx1];
    const float4 p10 = md[(size_t)sy1 * ws + sx], p11 = md[(size_t)sy1 * ws + sx1];
#define PSM_BILIN(C) __fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p00.C, a0), __fmul_rn(p01.C, a1)), b0), \
                               __fmul_rn(__fadd_rn(__fmul_rn(p10.C, a0), __fmul_rn(p11.C, a1)), b1))
    const float ur = PSM_BILIN(x), ug = PSM_BILIN(y), ub = PSM_BILIN(z), uq = PSM_BILIN(w);
#undef PSM_BILIN
    const float4 g = g1[(size_t)y * W + x];
    // src/fastguidedfilter.cpp:204: mean_a_r.mul(I_r) + mean_a_g.mul(I_g) + mean_a_b.mul(I_b) + mean_b
    vol[(size_t)d * H * W + (size_t)y * W + x] =
        __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(ur, g.x), __fmul_rn(ug, g.y)), __fmul_rn(ub, g.z)), uq);
}

}  // namespace

// FgfScratch: small planes of one side (allocated by the API layer)
void launch_fgf_setup(hipStream_t s, const float4 *g1, int W, int H, int sub, float4 *ism, float4 *msm, float4 *v1, float2 *v2)
{
    const int ws = W / sub, hs = H / sub, k = 2 * (8 / sub) + 1;
    dim3 grid((ws + 255) / 256, hs);
    hipLaunchKernelGGL(k_fgf_small_img, grid, dim3(256), 0, s, g1, W, H, ws, hs, ism);
    if (k == 3) hipLaunchKernelGGL(k_fgf_setup<3>, grid, dim3(256), 0, s, (const float4 *)ism, ws, hs, msm, v1, v2);
    else if (k == 5) hipLaunchKernelGGL(k_fgf_setup<5>, grid, dim3(256), 0, s, (const float4 *)ism, ws, hs, msm, v1, v2);
    else hipLaunchKernelGGL(k_fgf_setup<9>, grid, dim3(256), 0, s, (const float4 *)ism, ws, hs, msm, v1, v2);
}

void launch_fgf_filter(hipStream_t s, float *vol, const float4 *g1, int W, int H, int Dloc, int sub, const float4 *ism,
                       const float4 *msm, const float4 *v1, const float2 *v2, float4 *ab, float4 *mab)
{
    const int ws = W / sub, hs = H / sub, k = 2 * (8 / sub) + 1;
    dim3 gs((ws + 255) / 256, hs, Dloc);
    if (k == 3) {
        hipLaunchKernelGGL(k_fgf_model<3>, gs, dim3(256), 0, s, (const float *)vol, W, H, ws, hs, ism, msm, v1, v2, ab);
        hipLaunchKernelGGL(k_fgf_smooth<3>, gs, dim3(256), 0, s, (const float4 *)ab, ws, hs, mab);
    } else if (k == 5) {
        hipLaunchKernelGGL(k_fgf_model<5>, gs, dim3(256), 0, s, (const float *)vol, W, H, ws, hs, ism, msm, v1, v2, ab);
        hipLaunchKernelGGL(k_fgf_smooth<5>, gs, dim3(256), 0, s, (const float4 *)ab, ws, hs, mab);
    } else {
        hipLaunchKernelGGL(k_fgf_model<9>, gs, dim3(256), 0, s, (const float *)vol, W, H, ws, hs, ism, msm, v1, v2, ab);
        hipLaunchKernelGGL(k_fgf_smooth<9>, gs, dim3(256), 0, s, (const float4 *)ab, ws, hs, mab);
    }
    dim3 gf((W + 255) / 256, H, Dloc);
    hipLaunchKernelGGL(k_fgf_apply, gf, dim3(256), 0, s, (const float4 *)mab, ws, hs, g1, W, H, vol);
}

}  // namespace psm
