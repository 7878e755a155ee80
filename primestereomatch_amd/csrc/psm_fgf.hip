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
// Kernels (DESIGN.md "Fast Guided Filter row"):
//   k_fgf_small_img / k_fgf_setup   per image, small grid, direct k x k windows (d-invariant, ~10 us)
//   k_fgf_model<K, MODE>            per slice, small grid: one wave marches down 64 subsampled columns; the row of
//                                   (p, I0 p, I1 p, I2 p) goes through LDS for the k horizontal taps, the k row sums
//                                   live in a register ring for the vertical taps.  MODE 1/2 builds the matching
//                                   cost of the sampled pixel from the g1 planes, so the FGF path never needs the
//                                   full-resolution cost volume in memory (it samples 1/s^2 of it).
//   k_fgf_smooth<K>                 same marching scheme on the (a_r,a_g,a_b,b) planes
//   k_fgf_apply_wta                 full resolution, default consumer: upsample + linear model + winner-takes-all in one
//                                   pass (4 pixels x 2 rows per thread, guidance in registers across a chunk of slices,
//                                   one 64-bit atomic minimum per pixel and chunk) - the filtered volume is never written
//   k_fgf_apply4                    the same upsample + model writing the volume (4 pixels x 4 rows per thread, one
//                                   16-byte store per lane and row); runs only when something else reads the volume
#include "psm_kernels.h"
#include "psm_cost.h"
#include "psm_dev.h"

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

// ---- separable k x k mean of four planes (one float4 per pixel), marching down the rows -------------
// One 64-lane workgroup owns OUTW = 64 - 2R output columns (+R halo lanes each side, reflected) of one slice and
// the rows [ybeg, yend).  Summation order = cv::blur's: x taps left to right (fp64), then y taps top to bottom.
template <int K, class Load, class Finish>
__device__ __forceinline__ void blur4_march(int ws, int hs, int strip, int ybeg, int yend, float4 *lds, Load load, Finish finish)
{
    constexpr int R = K / 2, OUTW = 64 - 2 * R;
    const int lane = threadIdx.x;
    const int x = strip * OUTW - R + lane;
    const int xc = r101s(x, ws);
    const bool out_lane = lane >= R && lane < 64 - R && x < ws;
    const double scale = 1.0 / (K * K);
    int tap[K];
#pragma unroll
    for (int i = 0; i < K; ++i) {
        int l = lane - R + i;
        tap[i] = l < 0 ? 0 : (l > 63 ? 63 : l);
    }
    double ring[K][4];
#pragma unroll
    for (int u = 0; u < K; ++u) ring[u][0] = ring[u][1] = ring[u][2] = ring[u][3] = 0.0;
    int par = 0;
    for (int yy0 = ybeg - R; yy0 < yend + R; yy0 += K) {
#pragma unroll
        for (int u = 0; u < K; ++u) {
            const int yy = yy0 + u;
            if (yy < yend + R) {  // uniform over the workgroup
                lds[par * 64 + lane] = load(r101s(yy, hs), xc);
                __syncthreads();
                double h0 = 0.0, h1 = 0.0, h2 = 0.0, h3 = 0.0;
#pragma unroll
                for (int i = 0; i < K; ++i) {
                    const float4 t = lds[par * 64 + tap[i]];
                    h0 = __dadd_rn(h0, (double)t.x);
                    h1 = __dadd_rn(h1, (double)t.y);
                    h2 = __dadd_rn(h2, (double)t.z);
                    h3 = __dadd_rn(h3, (double)t.w);
                }
                ring[u][0] = h0; ring[u][1] = h1; ring[u][2] = h2; ring[u][3] = h3;
                par ^= 1;
                if (yy >= ybeg + R) {  // rows yy-2R .. yy are in the ring, oldest in slot u+1
                    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
                    for (int j = 1; j <= K; ++j) {
                        const int sl = (u + j) % K;
                        a0 = __dadd_rn(a0, ring[sl][0]);
                        a1 = __dadd_rn(a1, ring[sl][1]);
                        a2 = __dadd_rn(a2, ring[sl][2]);
                        a3 = __dadd_rn(a3, ring[sl][3]);
                    }
                    if (out_lane)
                        finish(yy - R, x, make_float4((float)(a0 * scale), (float)(a1 * scale), (float)(a2 * scale), (float)(a3 * scale)));
                }
            }
        }
    }
}

// ---- per slice, small grid: means of p and I*p -> linear model (a_r,a_g,a_b,b) ---------------------
// MODE 0: p from the cost volume in memory; 1 / 2: p = left / right matching cost of the sampled pixel, built
// from the g1 planes (CVC::buildCV_left/right arithmetic, src/CVC.cpp:122-179)
template <int K, int MODE>
__global__ __launch_bounds__(64) void k_fgf_model(const float *__restrict__ vol, int W, int H, int ws, int hs, int nstrips,
                                                 int seg, const float4 *__restrict__ g1, const float4 *__restrict__ g1_other,
                                                 int d_begin, const float4 *__restrict__ msm, const float4 *__restrict__ v1,
                                                 const float2 *__restrict__ v2, float4 *__restrict__ ab)
{
    __shared__ float4 lds[2 * 64];
    const int strip = blockIdx.x % nstrips, sgi = blockIdx.x / nstrips, d = blockIdx.y;
    const int ybeg = sgi * seg, yend = ybeg + seg < hs ? ybeg + seg : hs;
    const float *vd = vol + (size_t)d * H * W;
    const int D = d_begin + d;
    int Xc = -1;  // full-resolution column of this lane's subsampled column (fixed for the whole march)
    auto load = [&](int ys, int xs) -> float4 {
        if (Xc < 0) Xc = nn_src(xs, ws, W);
        const size_t o = (size_t)nn_src(ys, hs, H) * W;
        const float4 g = g1[o + Xc];
        float p;
        if (MODE == 0) p = vd[o + Xc];
        else if (MODE == 1) p = Xc >= D ? cost_pair(g, g1_other[o + Xc - D]) : cost_border(g);
        else p = Xc < W - D ? cost_pair(g, g1_other[o + Xc + D]) : cost_border(g);
        return make_float4(p, __fmul_rn(g.x, p), __fmul_rn(g.y, p), __fmul_rn(g.z, p));
    };
    auto finish = [&](int y, int x, float4 mean) {
        const float mp = mean.x, mr = mean.y, mg = mean.z, mb = mean.w;
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
    };
    blur4_march<K>(ws, hs, strip, ybeg, yend, lds, load, finish);
}

// ---- per slice, small grid: k x k means of the model planes ----------------------------------------
template <int K>
__global__ __launch_bounds__(64) void k_fgf_smooth(const float4 *__restrict__ ab, int ws, int hs, int nstrips, int seg,
                                                  float4 *__restrict__ mab)
{
    __shared__ float4 lds[2 * 64];
    const int strip = blockIdx.x % nstrips, sgi = blockIdx.x / nstrips, d = blockIdx.y;
    const int ybeg = sgi * seg, yend = ybeg + seg < hs ? ybeg + seg : hs;
    const float4 *ad = ab + (size_t)d * hs * ws;
    float4 *md = mab + (size_t)d * hs * ws;
    auto load = [&](int ys, int xs) -> float4 { return ad[(size_t)ys * ws + xs]; };
    auto finish = [&](int y, int x, float4 mean) { md[(size_t)y * ws + x] = mean; };
    blur4_march<K>(ws, hs, strip, ybeg, yend, lds, load, finish);
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


// 4 pixels x RB rows per thread, looping over a chunk of slices with the guidance in registers (W % 4 == 0).
// Plain fp32 arithmetic: on this part a packed fp32 op issues in four cycles, a scalar one in two, and the vector
// forms cost registers (even-aligned pairs) and therefore occupancy.
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
// Software pipeline: the model rows of slice d+1 are requested before the outputs of slice d are stored.  A wave's
// vector memory operations retire through one in-order counter, so a load issued after a store cannot be waited
// for before that store has reached memory; issued in this order the loads only ever wait for stores that are a
// whole iteration old.
// Row blocks start at y = RB*by - yshift: with yshift = (s/2) % RB the RB rows of a block share one pair of model
// rows whenever H is a multiple of s, so a block needs exactly two interpolated model rows per slice (the generic
// "next pair" step below stays for the other sizes).
template <int RB>
__global__ __launch_bounds__(256) void k_fgf_apply4(const f4v *__restrict__ mab, int ws, int hs, const f4v *__restrict__ g1,
                                                   int W, int H, float *__restrict__ vol, int Dloc, int dchunk, int yshift)
{
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4, y0 = (int)blockIdx.y * RB - yshift;
    if (x0 >= W) return;
    const int dbeg = blockIdx.z * dchunk, dend = dbeg + dchunk < Dloc ? dbeg + dchunk : Dloc;
    unsigned oa[4], ob[4];   // byte offsets of the two model columns of each pixel within a model row
    float a0[4], a1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int ca;
        lin_src(x0 + j, ws, W, ca, a1[j]);
        oa[j] = (unsigned)ca * 16u;
        ob[j] = (unsigned)(ca + 1 < ws ? ca + 1 : ws - 1) * 16u;
        a0[j] = __fsub_rn(1.f, a1[j]);
    }
    float gx[RB][4], gy[RB][4], gz[RB][4];
    int sy[RB], sy1[RB];
    float b0[RB], b1[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        int y = y0 + r;
        y = y < 0 ? 0 : (y < H ? y : H - 1);   // rows outside the image are skipped below; clamped here to stay in range
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f4v g = g1[(size_t)y * W + x0 + j];
            gx[r][j] = g.x; gy[r][j] = g.y; gz[r][j] = g.z;
        }
        lin_src(y, hs, H, sy[r], b1[r]);
        sy1[r] = sy[r] + 1 < hs ? sy[r] + 1 : hs - 1;
        b0[r] = __fsub_rn(1.f, b1[r]);
    }
    const size_t HW = (size_t)H * W;
    f4v Ra[8], Rb[8];   // raw model columns {S[sx], S[sx+1]} of the 4 pixels, rows sy[0] and sy1[0]
    auto request = [&](int d, int row, f4v *R) {
        size_t roff = (((size_t)d * hs + row) * ws) * sizeof(f4v);   // uniform base + 32-bit lane offsets
        asm volatile("" : "+s"(roff));   // keep it one scalar base: no per-load 64-bit induction pointers
        const char *mr = (const char *)mab + roff;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            R[2 * j] = *(const f4v *)(mr + oa[j]);
            R[2 * j + 1] = *(const f4v *)(mr + ob[j]);
        }
    };
    auto lerp4 = [](f4v s0, float w0, f4v s1, float w1) -> f4v {   // S0*w0 + S1*w1 per component, fp32, no FMA
        f4v r;
        r.x = __fadd_rn(__fmul_rn(s0.x, w0), __fmul_rn(s1.x, w1));
        r.y = __fadd_rn(__fmul_rn(s0.y, w0), __fmul_rn(s1.y, w1));
        r.z = __fadd_rn(__fmul_rn(s0.z, w0), __fmul_rn(s1.z, w1));
        r.w = __fadd_rn(__fmul_rn(s0.w, w0), __fmul_rn(s1.w, w1));
        return r;
    };
    auto hrow = [&](const f4v *R, f4v *U) {  // row pass of cv::resize INTER_LINEAR: S[sx]*(1-fx) + S[sx+1]*fx
#pragma unroll
        for (int j = 0; j < 4; ++j) U[j] = lerp4(R[2 * j], a0[j], R[2 * j + 1], a1[j]);
    };
    request(dbeg, sy[0], Ra);
    request(dbeg, sy1[0], Rb);
    for (int d = dbeg; d < dend; ++d) {
        f4v Ua[4], Ub[4];
        hrow(Ra, Ua);
        hrow(Rb, Ub);
        __builtin_amdgcn_sched_barrier(0);
        if (d + 1 < dend) {
            request(d + 1, sy[0], Ra);
            request(d + 1, sy1[0], Rb);
        }
        __builtin_amdgcn_sched_barrier(0);
        f4v o[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            // consecutive image rows use the same pair of model rows or the next one (uniform over the block):
            // sy advances by 0 or 1 per row and the new sy is the old sy1
            if (r > 0 && sy[r] != sy[r - 1]) {
                f4v Rc[8];
                request(d, sy1[r], Rc);
#pragma unroll
                for (int j = 0; j < 4; ++j) Ua[j] = Ub[j];
                hrow(Rc, Ub);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f4v u = lerp4(Ua[j], b0[r], Ub[j], b1[r]);      // column pass
                // src/fastguidedfilter.cpp:204: mean_a_r.mul(I_r) + mean_a_g.mul(I_g) + mean_a_b.mul(I_b) + mean_b
                o[r][j] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(u.x, gx[r][j]), __fmul_rn(u.y, gy[r][j])), __fmul_rn(u.z, gz[r][j])), u.w);
            }
        }
#pragma unroll
        for (int r = 0; r < RB; ++r)
            if (y0 + r >= 0 && y0 + r < H) *(f4v *)(vol + (size_t)d * HW + (size_t)(y0 + r) * W + x0) = o[r];
    }
}


// ---- upsample + linear model + WTA in one pass: the filtered volume is never written ---------------
// Same arithmetic per voxel as k_fgf_apply4; instead of storing q the thread keeps the running strict-< minimum
// over its chunk of slices (DispSel::CVSelect, src/DispSel.cpp:83-109: d = 0 is never a candidate, NaN never
// wins) and merges it into the per-pixel key plane with one 64-bit atomic minimum per pixel and chunk.
__global__ __launch_bounds__(256) void k_fgf_key_init(long long *__restrict__ keys, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) keys[i] = pack_key_f32(__builtin_inff(), 0);
}

template <int RB>
__global__ __launch_bounds__(256) void k_fgf_apply_wta(const f4v *__restrict__ mab, int ws, int hs, const f4v *__restrict__ g1,
                                                      int W, int H, int Dloc, int dchunk, int yshift, int d_begin,
                                                      long long *__restrict__ keys)
{
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4, y0 = (int)blockIdx.y * RB - yshift;
    if (x0 >= W) return;
    const int dbeg = blockIdx.z * dchunk, dend = dbeg + dchunk < Dloc ? dbeg + dchunk : Dloc;
    unsigned oa[4], ob[4];
    float a0[4], a1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int ca;
        lin_src(x0 + j, ws, W, ca, a1[j]);
        oa[j] = (unsigned)ca * 16u;
        ob[j] = (unsigned)(ca + 1 < ws ? ca + 1 : ws - 1) * 16u;
        a0[j] = __fsub_rn(1.f, a1[j]);
    }
    float gx[RB][4], gy[RB][4], gz[RB][4];
    int sy[RB], sy1[RB];
    float b0[RB], b1[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        int y = y0 + r;
        y = y < 0 ? 0 : (y < H ? y : H - 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f4v g = g1[(size_t)y * W + x0 + j];
            gx[r][j] = g.x; gy[r][j] = g.y; gz[r][j] = g.z;
        }
        lin_src(y, hs, H, sy[r], b1[r]);
        sy1[r] = sy[r] + 1 < hs ? sy[r] + 1 : hs - 1;
        b0[r] = __fsub_rn(1.f, b1[r]);
    }
    float mc[RB][4];
    int md[RB][4];
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) { mc[r][j] = __builtin_inff(); md[r][j] = 0; }
    auto lerp4 = [](f4v s0, float w0, f4v s1, float w1) -> f4v {   // S0*w0 + S1*w1 per component, fp32, no FMA
        f4v r;
        r.x = __fadd_rn(__fmul_rn(s0.x, w0), __fmul_rn(s1.x, w1));
        r.y = __fadd_rn(__fmul_rn(s0.y, w0), __fmul_rn(s1.y, w1));
        r.z = __fadd_rn(__fmul_rn(s0.z, w0), __fmul_rn(s1.z, w1));
        r.w = __fadd_rn(__fmul_rn(s0.w, w0), __fmul_rn(s1.w, w1));
        return r;
    };
    auto hrow = [&](int d, int row, f4v *U) {
        size_t roff = (((size_t)d * hs + row) * ws) * sizeof(f4v);
        asm volatile("" : "+s"(roff));
        const char *mr = (const char *)mab + roff;
#pragma unroll
        for (int j = 0; j < 4; ++j) U[j] = lerp4(*(const f4v *)(mr + oa[j]), a0[j], *(const f4v *)(mr + ob[j]), a1[j]);
    };
    for (int d = dbeg; d < dend; ++d) {
        const int dg = d_begin + d;
        if (dg == 0) continue;            // d = 0 is never a candidate (src/DispSel.cpp:96)
        f4v Ua[4], Ub[4];
        hrow(d, sy[0], Ua);
        hrow(d, sy1[0], Ub);
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            if (r > 0 && sy[r] != sy[r - 1]) {
#pragma unroll
                for (int j = 0; j < 4; ++j) Ua[j] = Ub[j];
                hrow(d, sy1[r], Ub);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f4v u = lerp4(Ua[j], b0[r], Ub[j], b1[r]);
                const float q = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(u.x, gx[r][j]), __fmul_rn(u.y, gy[r][j])), __fmul_rn(u.z, gz[r][j])), u.w);
                if (q < mc[r][j]) { mc[r][j] = q; md[r][j] = dg; }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RB; ++r)
        if (y0 + r >= 0 && y0 + r < H) {
#pragma unroll
            for (int j = 0; j < 4; ++j) atomicMin(keys + (size_t)(y0 + r) * W + x0 + j, pack_key_f32(mc[r][j], md[r][j]));
        }
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

void launch_fgf_model(hipStream_t s, const float *vol, const float4 *g1, const float4 *g1_other, int W, int H, int Dloc, int d_begin,
                      int sub, int cvc_mode, const float4 *msm, const float4 *v1, const float2 *v2, float4 *ab, float4 *mab)
{
    const int ws = W / sub, hs = H / sub, k = 2 * (8 / sub) + 1;
    const int nstrips = (ws + (64 - 2 * (k / 2)) - 1) / (64 - 2 * (k / 2));
    // rows per marching segment: enough workgroups to fill the chip, at least 4 k rows each
    int nsegs = (8192 + nstrips * Dloc - 1) / (nstrips * Dloc);
    int seg = (hs + nsegs - 1) / nsegs;
    if (seg < 4 * k) seg = 4 * k;
    if (seg > hs) seg = hs;
    nsegs = (hs + seg - 1) / seg;
    dim3 gs(nstrips * nsegs, Dloc);
#define PSM_FGF_MODEL(KK, MM) hipLaunchKernelGGL((k_fgf_model<KK, MM>), gs, dim3(64), 0, s, (const float *)vol, W, H, ws, hs, nstrips, seg, \
                                                 g1, g1_other, d_begin, msm, v1, v2, ab)
#define PSM_FGF_K(KK)                                                                                           \
    do {                                                                                                        \
        if (cvc_mode == 0) PSM_FGF_MODEL(KK, 0); else if (cvc_mode == 1) PSM_FGF_MODEL(KK, 1); else PSM_FGF_MODEL(KK, 2); \
        hipLaunchKernelGGL(k_fgf_smooth<KK>, gs, dim3(64), 0, s, (const float4 *)ab, ws, hs, nstrips, seg, mab);  \
    } while (0)
    if (k == 3) PSM_FGF_K(3); else if (k == 5) PSM_FGF_K(5); else PSM_FGF_K(9);
#undef PSM_FGF_K
#undef PSM_FGF_MODEL
}

void launch_fgf_apply(hipStream_t s, float *vol, const float4 *g1, int W, int H, int Dloc, int sub, const float4 *mab)
{
    const int ws = W / sub, hs = H / sub;
    if (W % 4 == 0) {
        const int dchunk = Dloc < 32 ? Dloc : 32;
        const int rb = sub == 2 ? 2 : 4, yshift = (sub / 2) % rb;
        dim3 gf((W / 4 + 255) / 256, (H + yshift + rb - 1) / rb, (Dloc + dchunk - 1) / dchunk);
        if (rb == 2)
            hipLaunchKernelGGL(k_fgf_apply4<2>, gf, dim3(256), 0, s, (const f4v *)mab, ws, hs, (const f4v *)g1, W, H, vol, Dloc, dchunk, yshift);
        else
            hipLaunchKernelGGL(k_fgf_apply4<4>, gf, dim3(256), 0, s, (const f4v *)mab, ws, hs, (const f4v *)g1, W, H, vol, Dloc, dchunk, yshift);
    } else {
        dim3 gf((W + 255) / 256, H, Dloc);
        hipLaunchKernelGGL(k_fgf_apply, gf, dim3(256), 0, s, (const float4 *)mab, ws, hs, g1, W, H, vol);
    }
}

bool fgf_can_fuse_wta(int W) { return (W & 3) == 0; }

void launch_fgf_apply_wta(hipStream_t s, const float4 *g1, int W, int H, int Dloc, int d_begin, int sub, const float4 *mab,
                          long long *keys)
{
    const int ws = W / sub, hs = H / sub, n = W * H;
    hipLaunchKernelGGL(k_fgf_key_init, dim3((n + 255) / 256), dim3(256), 0, s, keys, n);
    const int dchunk = Dloc < 32 ? Dloc : 32;
    // two rows per thread, plain fp32 arithmetic: 134 VGPRs, three waves per SIMD -> 0.75 ms per 1080p x 256 volume.
    // Measured alternatives: four rows 1.00 ms, one row 1.35 ms; packed-fp32 vector arithmetic (186 VGPRs) 0.98 ms;
    // loading every distinct model column once per row instead of once per pixel and tap 0.87 ms (the duplicate
    // addresses coalesce in the L1, the extra control flow does not pay).
    constexpr int rb = 2;
    const int yshift = (sub / 2) % rb;
    dim3 gf((W / 4 + 255) / 256, (H + yshift + rb - 1) / rb, (Dloc + dchunk - 1) / dchunk);
    hipLaunchKernelGGL(k_fgf_apply_wta<rb>, gf, dim3(256), 0, s, (const f4v *)mab, ws, hs, (const f4v *)g1, W, H, Dloc, dchunk, yshift,
                       d_begin, keys);
}

}  // namespace psm
