// psm_kernels.hip - hand-written gfx950 (CDNA4, wave64) kernels of the DispEst hot path.
//
// Arithmetic contract (order fixed in oracle/psm_oracle.h):
//   * every fp32 expression is evaluated op-for-op (no FMA contraction: the file is built with
//     -ffp-contract=off AND the expressions use __fmul_rn/__fadd_rn/__fsub_rn), IEEE division;
//   * cv::boxFilter(Size(8,8)) = fp64 window sums, horizontal then vertical, each 8-tap sum
//     evaluated as the balanced tree ((t0+t1)+(t2+t3))+((t4+t5)+(t6+t7)), x(1/64), round to fp32;
//     taps at offsets -4..+3, BORDER_REFLECT_101.
// Reference arithmetic followed: src/CVC.cpp:18-46,122-179, src/CVF.cpp:44-165,
// src/DispSel.cpp:83-109, src/PP.cpp:17-50 (paths in the reference repository).
//
// Kernel design: the guided filter's two box-filter rounds are "marching" kernels: one wave
// owns 64 adjacent columns of one disparity slice (up to 57 outputs + 7 halo) and walks down the rows.
// Horizontal 8-tap sums are built with three cross-lane exchanges (distance 1, 2, 4: a sliding balanced
// tree, 3 fp64 adds per output), vertical sums with a register-resident sliding tree (7 doubles of state
// per channel, 3 fp64 adds per output).  The product path is k_cvf_pc: cost build + both filter rounds in
// one kernel, producer and consumer waves coupled through an LDS ring, so a voxel costs one 4-byte store
// and no volume read.  The two-stage kernels (k_cvc_t, k_cvf_a, k_cvf_b), the plain box filter (k_box8) and
// the direct per-voxel kernels are the fallback / diagnostic / cross-check forms of the same arithmetic.
// The Fast Guided Filter row lives in psm_fgf.hip, the fused product kernel in psm_pc.hip, post-processing in psm_pp.hip.
// MFMA is not used: nothing here is a dense contraction.
#include "psm_kernels.h"
#include "psm_cost.h"
#include "psm_dev.h"

namespace psm {

// ------------------------------------------------------------------------------------------
// image preparation: planarise + scale + gray + x-gradient  -> g1 = {I0,I1,I2,GrdX}
// (src/StereoMatch.cpp:195-196 convertTo; src/CVC.cpp:41-46 CVC::preprocess)
// ------------------------------------------------------------------------------------------
template <bool F32>
__device__ __forceinline__ void load_px(const void *src, size_t pitch, int y, int x, float &c0, float &c1, float &c2)
{
    if (F32) {
        const float *p = (const float *)((const char *)src + (size_t)y * pitch) + 3 * x;
        c0 = p[0]; c1 = p[1]; c2 = p[2];
    } else {
        const uint8_t *p = (const uint8_t *)src + (size_t)y * pitch + 3 * x;
        const float alpha = 1 / 255.0f;
        c0 = __fmul_rn((float)p[0], alpha);
        c1 = __fmul_rn((float)p[1], alpha);
        c2 = __fmul_rn((float)p[2], alpha);
    }
}

template <bool F32>
__global__ __launch_bounds__(256) void k_prep(const void *src, size_t pitch, int W, int H, float4 *g1, const void *src1, float4 *g11,
                                             const PcPair *__restrict__ tab)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (tab) { src = tab[blockIdx.z >> 1].raw[blockIdx.z & 1]; g1 = tab[blockIdx.z >> 1].g[blockIdx.z & 1].g1; }   // batch: image z & 1 of pair z >> 1
    else if (blockIdx.z == 1) { src = src1; g1 = g11; }    // second image of a two-image launch
    if (x >= W) return;
    float c0, c1, c2, l0, l1, l2, r0, r1, r2;
    load_px<F32>(src, pitch, y, x, c0, c1, c2);
    load_px<F32>(src, pitch, y, r101(x - 1, W), l0, l1, l2);
    load_px<F32>(src, pitch, y, r101(x + 1, W), r0, r1, r2);
    float grd = __fsub_rn(gray_of(r0, r1, r2), gray_of(l0, l1, l2));
    g1[(size_t)y * W + x] = make_float4(c0, c1, c2, grd);
}

void launch_prep(hipStream_t s, const void *src, size_t pitch, int depth_f32, int W, int H, float4 *g1, const void *src1, float4 *g11)
{   // src1 != NULL: both images in one launch
    dim3 grid((W + 255) / 256, H, src1 ? 2 : 1);
    if (depth_f32)
        hipLaunchKernelGGL(k_prep<true>, grid, dim3(256), 0, s, src, pitch, W, H, g1, src1, g11, (const PcPair *)nullptr);
    else
        hipLaunchKernelGGL(k_prep<false>, grid, dim3(256), 0, s, src, pitch, W, H, g1, src1, g11, (const PcPair *)nullptr);
}

// ------------------------------------------------------------------------------------------
// Range of a float buffer as biased exponents: out[0] = max exponent, out[1] = min exponent over the non-zero values
// (caller initialises {0, 255}; NaN / inf count as 255, subnormals as 0).  Guards the domain of the scaled window sums of the
// select forms (psm_pc.hip): float images and uploaded cost volumes outside it run the storing form.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_range_f32(const float *__restrict__ p, size_t n, unsigned *out)
{
    __shared__ unsigned smax[4], smin[4];
    unsigned emax = 0, emin = 255;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned b = __float_as_uint(p[i]) & 0x7fffffffu;
        if (b) { const unsigned e = b >> 23; emax = max(emax, e); emin = min(emin, e); }
    }
    for (int o = 32; o > 0; o >>= 1) { emax = max(emax, (unsigned)__shfl_xor((int)emax, o)); emin = min(emin, (unsigned)__shfl_xor((int)emin, o)); }
    if ((threadIdx.x & 63) == 0) { smax[threadIdx.x >> 6] = emax; smin[threadIdx.x >> 6] = emin; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMax(out, max(max(smax[0], smax[1]), max(smax[2], smax[3])));
        atomicMin(out + 1, min(min(smin[0], smin[1]), min(smin[2], smin[3])));
    }
}
// Copy n bytes (n % 16 == 0, 16-byte aligned) between device memory and page-locked host memory with a kernel instead of the
// copy engines: the asynchronous PCIe legs of a frame loop (psm_upload_pair_async, psm_download_maps_async).  A kernel behind a
// hipStreamWaitEvent is ordered by the command processor; the runtime's asynchronous copies behind such a wait cost 0.2 - 0.4 ms
// of HOST time each on this stack (measured with 8 contexts per frame), which tied the host to the device's pace.
__global__ __launch_bounds__(256) void k_copy16(uint4 *__restrict__ dst, const uint4 *__restrict__ src, size_t n16, int tail)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
    if (blockIdx.x == 0 && (int)threadIdx.x < tail)     // the last bytes % 16, one by one
        reinterpret_cast<uint8_t *>(dst + n16)[threadIdx.x] = reinterpret_cast<const uint8_t *>(src + n16)[threadIdx.x];
}
void launch_copy_bytes(hipStream_t s, void *dst, const void *src, size_t bytes)
{   // dst and src 16-byte aligned
    const size_t n16 = bytes / 16;
    const unsigned blocks = (unsigned)((n16 + 255) / 256 < 64 ? (n16 + 255) / 256 : 64);     // a few workgroups saturate the link
    hipLaunchKernelGGL(k_copy16, dim3(blocks ? blocks : 1), dim3(256), 0, s, (uint4 *)dst, (const uint4 *)src, n16, (int)(bytes % 16));
}

void launch_range_f32(hipStream_t s, const float *p, size_t n, unsigned *out)
{
    const unsigned blocks = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(k_range_f32, dim3(blocks ? blocks : 1), dim3(256), 0, s, p, n, out);
}

// ------------------------------------------------------------------------------------------
// guidance precompute (CVF::preprocess, src/CVF.cpp:44-70, + the d-invariant part of the solve,
// src/CVF.cpp:120-132): 9 channels I0,I1,I2,I0I0,I0I1,I0I2,I1I1,I1I2,I2I2.  One wave marches a 56-column strip down a
// segment of rows with nine sliding trees - the same machinery as the volume kernels.  (Round 1's two-pass form
// wrote and re-read 149 MB of fp64 row sums per side; this work does not shrink when the job is sharded over GPUs.)
// ------------------------------------------------------------------------------------------
struct GuideM { float m[9]; };
__device__ __forceinline__ void guide_finish(const float *m, float4 &g2, float4 &g3, float2 &g4, bool fma)
{   // var_k = box(I_c*I_c') - mean_c*mean_c' (src/CVF.cpp:58-68); Sigma + eps I, DET (src/CVF.cpp:120-132); adjugate entries as
    // written at src/CVF.cpp:133-147 (the matrix is symmetric, so the nine expressions take six distinct values bit for bit)
    const float eps = 0.0001f;  // GIF_EPS, include/ComFunc.h:50
    float v0 = __fsub_rn(m[3], __fmul_rn(m[0], m[0]));
    float v1 = __fsub_rn(m[4], __fmul_rn(m[0], m[1]));
    float v2 = __fsub_rn(m[5], __fmul_rn(m[0], m[2]));
    float v3 = __fsub_rn(m[6], __fmul_rn(m[1], m[1]));
    float v4 = __fsub_rn(m[7], __fmul_rn(m[1], m[2]));
    float v5 = __fsub_rn(m[8], __fmul_rn(m[2], m[2]));
    float a11 = __fadd_rn(v0, eps), a12 = v1, a13 = v2;
    float a21 = v1, a22 = __fadd_rn(v3, eps), a23 = v4;
    float a31 = v2, a32 = v4, a33 = __fadd_rn(v5, eps);
    float X = __fsub_rn(__fmul_rn(a33, a22), __fmul_rn(a32, a23));
    float Y = __fsub_rn(__fmul_rn(a33, a12), __fmul_rn(a32, a13));
    float Z = __fsub_rn(__fmul_rn(a23, a12), __fmul_rn(a22, a13));
    float det = __fadd_rn(__fsub_rn(__fmul_rn(a11, X), __fmul_rn(a21, Y)), __fmul_rn(a31, Z));
    float inv = __fdiv_rn(1.0f, det);
    float A00 = __fsub_rn(__fmul_rn(a33, a22), __fmul_rn(a32, a23));
    float A01 = __fsub_rn(__fmul_rn(a31, a23), __fmul_rn(a33, a21));
    float A02 = __fsub_rn(__fmul_rn(a32, a21), __fmul_rn(a31, a22));
    float A11 = __fsub_rn(__fmul_rn(a33, a11), __fmul_rn(a31, a13));
    float A12 = __fsub_rn(__fmul_rn(a31, a12), __fmul_rn(a32, a11));
    float A22 = __fsub_rn(__fmul_rn(a22, a11), __fmul_rn(a21, a12));
    if (fma) {
        // PSM_FLAG_FMA_SOLVE: src/CVF.cpp:129-147 as GCC's -ffp-contract=fast contracts it on an FMA target (oracle:
        // PSMO_VAR_FMA_SOLVE, pinned against a live gcc -O2 -mfma compile of the same expressions by tests/test_oracle.py):
        // x*y - z*w -> fma(x, y, -RN(z*w)); p0 - p1 + p2 -> fma(x2, y2, fma(x0, y0, -RN(x1*y1))); a31*a23 and a32*a13 are ONE
        // rounded product, so (a31*a23 - a33*a21) = (a32*a13 - a33*a12) = fma(-a33, a21, RN(a31*a23)).  Still six distinct values.
#define PSM_M2(x, y, z, w) __fmaf_rn((x), (y), -__fmul_rn((z), (w)))
        const float m00 = PSM_M2(a33, a22, a32, a23), m01 = PSM_M2(a33, a12, a32, a13), m02 = PSM_M2(a23, a12, a22, a13);
        det = __fmaf_rn(a31, m02, __fmaf_rn(a11, m00, -__fmul_rn(a21, m01)));
        inv = __fdiv_rn(1.0f, det);
        A00 = m00;
        A01 = __fmaf_rn(-a33, a21, __fmul_rn(a31, a23));
        A02 = PSM_M2(a32, a21, a31, a22);
        A11 = PSM_M2(a33, a11, a31, a13);
        A12 = PSM_M2(a31, a12, a32, a11);
        A22 = PSM_M2(a22, a11, a21, a12);
#undef PSM_M2
    }
    g2 = make_float4(m[0], m[1], m[2], inv);
    g3 = make_float4(A00, A01, A02, A11);
    g4 = make_float2(A12, A22);
}

// Three waves per (strip, segment): wave w marches three of the nine channels (its own sliding trees: 42 tree registers
// instead of 126, so six and more waves per SIMD instead of two) and parks the three means of every output pixel in LDS;
// after each batch of four rows all 192 threads share the per-pixel finish (variances, adjugate, 1/DET) of the batch's
// 4 x 56 pixels.  Round 2 ran one wave per (strip, segment) with all nine trees: one resident round of two waves per SIMD,
// i.e. the kernel took as long as ONE wave's serial march (54 us at 1080p, 19 us at 450 x 375).  Same arithmetic, same bits.
// SRC 0: the image planes come from g1 (k_prep ran).  SRC 1 / 2 (round 6): CVC::preprocess (src/CVC.cpp:41-46, and the convertTo of
// src/StereoMatch.cpp:195-196 for 8-bit images) is done HERE - every wave converts the staged interleaved pixel of its column itself
// (u8: (float)b * (1/255.0f) as load_px; float: as it is), and wave 0 also forms gray and the x-gradient from its lane neighbours
// (the lanes of a wave hold consecutive columns, reflected at the image border exactly as r101 asks) and writes the g1 rows the
// workgroup owns: one launch instead of k_prep + k_guide_march, and g1 is written once instead of written and read back.
template <int SRC>
__global__ __launch_bounds__(192) void k_guide_march(const float4 *g1, int W, int H, int nstrips, int seg_rows,
                                                    float4 *g2, float4 *g3, float2 *g4, Guidance second, int ybeg, int yend,
                                                    const PcPair *__restrict__ tab, int fma, const void *raw0, const void *raw1, size_t pitch)
{
    __shared__ float ms[2][4][9][64];            // [batch parity][row of the batch][channel][lane]
    const void *raw = raw0;
    float4 *g1w = const_cast<float4 *>(g1);      // SRC != 0: g1 is an OUTPUT
    if (tab) {   // batch: image y & 1 of pair y >> 1
        const Guidance gg = tab[blockIdx.y >> 1].g[blockIdx.y & 1];
        g1 = gg.g1; g1w = gg.g1; g2 = gg.g2; g3 = gg.g3; g4 = gg.g4;
        raw = tab[blockIdx.y >> 1].raw[blockIdx.y & 1];
    } else if (blockIdx.y == 1) { g1 = second.g1; g1w = second.g1; g2 = second.g2; g3 = second.g3; g4 = second.g4; raw = raw1; }   // second image of a two-image launch
    const int strip = blockIdx.x % nstrips, seg = blockIdx.x / nstrips;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int x0 = strip * 56;
    const int cs = r101c(x0 - 4 + lane, W);
    const int y0 = ybeg + seg * seg_rows, y1 = min(yend, y0 + seg_rows);   // output rows [y0, y1) of the rows [ybeg, yend) asked for
    const int n = (y1 - y0) + 7, ybase = y0 - 4;
    const int i1 = ((lane + 1) & 63) << 2, i2 = ((lane + 2) & 63) << 2, i4 = ((lane + 4) & 63) << 2;
    const int im1 = ((lane + 63) & 63) << 2;
    (void)i1; (void)im1;
    VTree t[3] = {};
    // the image rows of the next batch of four steps are in flight while this batch computes
    auto load_row = [&](int yy) -> float4 {
        if constexpr (SRC == 0) return g1[(size_t)yy * W + cs];
        else if constexpr (SRC == 1) {
            // three bytes of an interleaved B,G,R pixel as one (unaligned) dword: the staged image buffers are 12 bytes per pixel
            // long whatever the depth (psm_create), so the byte past the last pixel exists
            const uint8_t *p = (const uint8_t *)raw + (size_t)yy * pitch + 3 * cs;
            unsigned v;
            __builtin_memcpy(&v, p, 4);
            const float alpha = 1 / 255.0f;
            return make_float4(__fmul_rn((float)(v & 0xffu), alpha), __fmul_rn((float)((v >> 8) & 0xffu), alpha), __fmul_rn((float)((v >> 16) & 0xffu), alpha), 0.0f);
        } else {
            const float *p = (const float *)((const char *)raw + (size_t)yy * pitch) + 3 * cs;
            return make_float4(p[0], p[1], p[2], 0.0f);
        }
    };
    float4 gq[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) gq[k] = load_row(r101c(ybase + k, H));
    for (int i = 0, b = 0; i < n; i += 4, ++b) {
        float4 gc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) gc[k] = gq[k];
#pragma unroll
        for (int k = 0; k < 4; ++k) gq[k] = load_row(r101c(ybase + i + 4 + k, H));
#define PSM_STEP_G(K)                                                                              \
    {                                                                                              \
        const float4 g = gc[K];                                                                    \
        if (SRC != 0 && wave == 0) {   /* CVC::preprocess of the rows this workgroup owns (step i + K is image row ybase + i + K) */ \
            const float gr_ = gray_of(g.x, g.y, g.z);                                              \
            const float grd_ = __fsub_rn(rol1(gr_), lane_get(gr_, im1));   /* gray(x + 1) - gray(x - 1): lanes l + 1 / l - 1 */ \
            const int st_ = i + K, xo_ = x0 - 4 + lane;                                            \
            if (st_ >= 4 && st_ < n - 3 && lane >= 4 && lane < 60 && xo_ < W)                      \
                g1w[(size_t)(ybase + st_) * W + xo_] = make_float4(g.x, g.y, g.z, grd_);            \
        }                                                                                          \
        /* channels I0,I1,I2 | I0I0,I0I1,I0I2 | I1I1,I1I2,I2I2: wave w takes the w-th triple */      \
        const float v0 = wave == 0 ? g.x : (wave == 1 ? __fmul_rn(g.x, g.x) : __fmul_rn(g.y, g.y)); \
        const float v1 = wave == 0 ? g.y : (wave == 1 ? __fmul_rn(g.x, g.y) : __fmul_rn(g.y, g.z)); \
        const float v2 = wave == 0 ? g.z : (wave == 1 ? __fmul_rn(g.x, g.z) : __fmul_rn(g.z, g.z)); \
        ms[b & 1][K][3 * wave + 0][lane] = box_out(vstep<K>(t[0], hsum8(v0, i1, i2, i4)));        \
        ms[b & 1][K][3 * wave + 1][lane] = box_out(vstep<K>(t[1], hsum8(v1, i1, i2, i4)));        \
        ms[b & 1][K][3 * wave + 2][lane] = box_out(vstep<K>(t[2], hsum8(v2, i1, i2, i4)));        \
    }
        PSM_STEP_G(0) PSM_STEP_G(1) PSM_STEP_G(2) PSM_STEP_G(3)
#undef PSM_STEP_G
        __syncthreads();     // (one barrier per batch: the other parity's buffer is rewritten only after the NEXT barrier)
        // finish the batch's pixels: 4 rows x 56 columns over 192 threads
        for (int q = threadIdx.x; q < 4 * 56; q += 192) {
            const int k = q / 56, col = q - k * 56;
            const int step = i + k, xo = x0 + col;
            if (step >= 7 && step < n && xo < W) {
                float m[9];
#pragma unroll
                for (int c = 0; c < 9; ++c) m[c] = ms[b & 1][k][c][col];
                float4 r2, r3; float2 r4;
                guide_finish(m, r2, r3, r4, fma != 0);
                const size_t o = (size_t)(ybase + step - 3) * W + xo;
                g2[o] = r2; g3[o] = r3; g4[o] = r4;
            }
        }
    }
}

void launch_guidance(hipStream_t s, Guidance g, int W, int H, const Guidance *second, int ybeg, int yend, bool fma,
                     const void *raw0, const void *raw1, size_t pitch, int raw_f32)
{   // second != NULL: the guidance of both images in one launch; [ybeg, yend) (yend <= ybeg: all rows): the rows of g2..g4
    // to produce - a row stripe of the filter needs its own rows + 4 either side.  raw0 / raw1 != NULL: CVC::preprocess in the
    // same launch - the images are read from the staged interleaved copies, and the g1 rows [ybeg, yend) are WRITTEN
    if (yend <= ybeg) { ybeg = 0; yend = H; }
    const int rows = yend - ybeg;
    // one workgroup of three waves per (strip, segment); segments as short as still fill ~2 workgroups per SIMD-quad of the
    // chip in one resident round (8..64 rows each: 7 halo rows per segment)
    const int nstrips = (W + 55) / 56;
    const PcDev dev = pc_dev();
    const int wgs = 6 * dev.nxcd * dev.cus_per_xcd / (second ? 2 : 1);   // per image (87 VGPRs, 18 KB of LDS: ~6 resident workgroups per CU)
    int seg_rows = 8;
    while (seg_rows < 64 && nstrips * ((rows + seg_rows - 1) / seg_rows) > wgs) ++seg_rows;
    const int nsegs = (rows + seg_rows - 1) / seg_rows;
#define PSM_LAUNCH_G(SRC)                                                                                                            \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_guide_march<SRC>), dim3(nstrips * nsegs, second ? 2 : 1), dim3(192), 0, s, (const float4 *)g.g1, W, H, nstrips, \
                       seg_rows, g.g2, g.g3, g.g4, second ? *second : Guidance{}, ybeg, yend, (const PcPair *)nullptr, fma ? 1 : 0, raw0, raw1, pitch)
    if (!raw0) PSM_LAUNCH_G(0); else if (raw_f32) PSM_LAUNCH_G(2); else PSM_LAUNCH_G(1);
#undef PSM_LAUNCH_G
}

// ---- the same two kernels for every pair of a batch (psm_compute_batch): one launch each, images indexed through the table ----
void launch_guidance_batch(hipStream_t s, const PcPair *tab, int npairs, int W, int H, size_t pitch, int src)
{   // src 0: g1 of every image exists (launch_prep_batch); 1 / 2: image preparation in the same launch, from the table's staged 8-bit /
    // float images (row pitch `pitch`), as launch_guidance does for one pair
    const int nstrips = (W + 55) / 56;
    const PcDev dev = pc_dev();
    const int wgs = 6 * dev.nxcd * dev.cus_per_xcd / (2 * npairs) > 0 ? 6 * dev.nxcd * dev.cus_per_xcd / (2 * npairs) : 1;
    int seg_rows = 8;
    while (seg_rows < 64 && nstrips * ((H + seg_rows - 1) / seg_rows) > wgs) ++seg_rows;
    const int nsegs = (H + seg_rows - 1) / seg_rows;
#define PSM_LAUNCH_GB(SRC)                                                                                                          \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_guide_march<SRC>), dim3(nstrips * nsegs, 2 * npairs), dim3(192), 0, s, (const float4 *)nullptr, W, H, nstrips, seg_rows, \
                       (float4 *)nullptr, (float4 *)nullptr, (float2 *)nullptr, Guidance{}, 0, H, tab, 0, (const void *)nullptr, (const void *)nullptr, pitch)
    if (src == 1) PSM_LAUNCH_GB(1); else if (src == 2) PSM_LAUNCH_GB(2); else PSM_LAUNCH_GB(0);
#undef PSM_LAUNCH_GB
}

// ------------------------------------------------------------------------------------------
// CVC: cost volume construction (src/CVC.cpp:18-39,122-179).  One thread = one pixel, DC
// consecutive disparities; lanes run along x so every store is a coalesced row segment and the
// partner-image reads of neighbouring lanes / iterations overlap in L1.
// ------------------------------------------------------------------------------------------
constexpr int CVC_DC = 8;
template <bool RIGHT>
__global__ __launch_bounds__(256) void k_cvc(const float4 *__restrict__ base, const float4 *__restrict__ other,
                                            float *__restrict__ vol, int W, int H, int d_begin, int Dloc, int y0)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y + y0;
    int dl0 = blockIdx.z * CVC_DC;
    if (x >= W) return;
    const size_t HW = (size_t)H * W;
    const size_t row = (size_t)y * W;
    float4 a = base[row + x];
    float cb = cost_border(a);
#pragma unroll
    for (int k = 0; k < CVC_DC; ++k) {
        int dl = dl0 + k;
        if (dl >= Dloc) break;
        int d = d_begin + dl;
        float c;
        if (RIGHT) {  // buildCV_right: partner x+d while x < W-d
            if (x < W - d) c = cost_pair(a, other[row + x + d]); else c = cb;
        } else {      // buildCV_left: partner x-d while x >= d
            if (x >= d) c = cost_pair(a, other[row + x - d]); else c = cb;
        }
        vol[(size_t)dl * HW + row + x] = c;
    }
}

// One pixel per lane for the (coalesced) loads and the arithmetic, but the eight cost rows of a wave
// are parked in LDS and written back with 16 bytes per lane: one store instruction then covers four
// 256-byte row pieces, each two complete cache lines, instead of four 64-byte partial writes per
// dword store (which made the L2 fill every line from HBM first).  Needs W % 4 == 0.
template <bool RIGHT>
__global__ __launch_bounds__(256) void k_cvc_t(const float4 *__restrict__ base, const float4 *__restrict__ other,
                                              float *__restrict__ vol, int W, int H, int d_begin, int Dloc, int y0)
{
    __shared__ __attribute__((aligned(16))) float lds[4][CVC_DC][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y + y0;
    const int dl0 = blockIdx.z * CVC_DC;
    const int xc = min(x, W - 1);
    const size_t HW = (size_t)H * W;
    const size_t row = (size_t)y * W;
    const float4 a = base[row + xc];
    const float cb = cost_border(a);
#pragma unroll
    for (int k = 0; k < CVC_DC; ++k) {
        const int d = d_begin + dl0 + k;
        float c;
        if (RIGHT) {
            if (xc < W - d) c = cost_pair(a, other[row + xc + d]); else c = cb;
        } else {
            if (xc >= d) c = cost_pair(a, other[row + xc - d]); else c = cb;
        }
        lds[wave][k][lane] = c;
    }
    __syncthreads();
    const int q4 = (lane & 15) * 4;
    const int xq = blockIdx.x * 256 + wave * 64 + q4;
#pragma unroll
    for (int j = 0; j < CVC_DC / 4; ++j) {
        const int k = (lane >> 4) + 4 * j;
        const int dl = dl0 + k;
        if (dl < Dloc && xq < W)
            *reinterpret_cast<float4 *>(vol + (size_t)dl * HW + row + xq) = *reinterpret_cast<const float4 *>(&lds[wave][k][q4]);
    }
}

void launch_cvc(hipStream_t s, const float4 *g1_base, const float4 *g1_other, float *vol, int W, int H,
                int d_begin, int Dloc, int right, int ybeg, int yend)
{
    if (yend <= ybeg) return;
    const int rows = yend - ybeg;
    const int nz = (Dloc + CVC_DC - 1) / CVC_DC;
#define PSM_LAUNCH_CVC(KERNEL, GX)                                                                              \
    {                                                                                                           \
        dim3 grid(GX, rows, nz);                                                                                \
        if (right) hipLaunchKernelGGL(KERNEL<true>, grid, dim3(256), 0, s, g1_base, g1_other, vol, W, H, d_begin, Dloc, ybeg); \
        else hipLaunchKernelGGL(KERNEL<false>, grid, dim3(256), 0, s, g1_base, g1_other, vol, W, H, d_begin, Dloc, ybeg);      \
    }
    if ((W & 3) == 0) PSM_LAUNCH_CVC(k_cvc_t, (W + 255) / 256)   // (k_cvc: widths that are not a multiple of 4)
    else PSM_LAUNCH_CVC(k_cvc, (W + 255) / 256)
#undef PSM_LAUNCH_CVC
}

// ------------------------------------------------------------------------------------------
// marching kernels
// ------------------------------------------------------------------------------------------
constexpr int OUT_PER_WAVE = 56;  // 64 lanes - 7 halo columns, rounded down to a multiple of 8 so that a
                                  // wave-row of float4 outputs (896 B) starts and ends on 128-byte lines:
                                  // measured 5.4 TB/s for aligned full-line stores vs 3.2 TB/s with 57

struct MarchPos {
    int d, lane, cs, xo, y0, y1;
    bool ok, ovalid;
};

// Block -> (column strip, y segment, slice group).  Blocks are observed to be dispatched round-robin over the XCDs
// (block b -> XCD b % nxcd); every XCD gets a contiguous range of (strip, segment) pairs and walks the slice groups of
// one pair back to back, so the pair's guidance stays in that XCD's L2 while all D slices stream past it.  Speed only -
// nothing depends on the placement.
template <int NW>
__device__ __forceinline__ MarchPos march_pos(int W, int ybeg, int yend, int Dloc, int nstrips, int nsegs, int seg_rows, int nzg, int nxcd)
{
    MarchPos p;
    const int id = blockIdx.x;
    const int xcd = id % nxcd, j = id / nxcd;
    const int npairs = nstrips * nsegs;
    const int ppx = (npairs + nxcd - 1) / nxcd;
    const int zg = j % nzg, pl = j / nzg;
    const int pair = xcd * ppx + pl;
    const bool ok = pl < ppx && pair < npairs;
    const int strip = pair % nstrips, seg = pair / nstrips;
    int wave = threadIdx.x >> 6;
    p.lane = threadIdx.x & 63;
    p.d = zg * NW + wave;
    p.ok = ok && p.d < Dloc;
    int x0 = strip * OUT_PER_WAVE;
    p.cs = r101c(x0 - 4 + p.lane, W);
    p.xo = x0 + p.lane;
    p.ovalid = p.lane < OUT_PER_WAVE && p.xo < W;
    p.y0 = ybeg + seg * seg_rows;
    p.y1 = min(yend, p.y0 + seg_rows);
    return p;
}

#define PSM_LANE_IDX()                             \
    const int i1 = ((pos.lane + 1) & 63) << 2;     \
    const int i2 = ((pos.lane + 2) & 63) << 2;     \
    const int i4 = ((pos.lane + 4) & 63) << 2

// The loops below are software pipelined by hand: at step s a wave first ISSUES the loads of step
// s+3 (input row and output-side guidance, clamped addresses, no branches), then consumes the
// registers of step s.  vmcnt retires in order, so the only wait per step is for data issued
// three steps earlier; with 3-4 waves per SIMD that covers the HBM/L2 latency.  The 4-slot
// register rings are indexed with compile-time constants (unroll by 4).
typedef float f4v __attribute__((ext_vector_type(4)));

// ---- stage A: p -> (a0,a1,a2,b) -------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(NW * 64) void k_cvf_a(const float *__restrict__ vol, float4 *__restrict__ ab,
                                                  const float4 *__restrict__ G1, const float4 *__restrict__ G2,
                                                  const float4 *__restrict__ G3, const float2 *__restrict__ G4,
                                                  int W, int H, int Dloc, int nstrips, int nsegs, int seg_rows, int nzg, int nxcd,
                                                  int ybeg, int yend)
{
    const MarchPos pos = march_pos<NW>(W, ybeg, yend, Dloc, nstrips, nsegs, seg_rows, nzg, nxcd);
    if (!pos.ok) return;
    PSM_LANE_IDX();
    const size_t HW = (size_t)H * W;
    const float *vd = vol + (size_t)pos.d * HW;
    float4 *abd = ab + (size_t)pos.d * HW;
    VTree t0 = {}, t1 = {}, t2 = {}, t3 = {};
    const int n = (pos.y1 - pos.y0) + 7;
    const int ybase = pos.y0 - 4;
    const int xoc = min(pos.xo, W - 1);

    float pin[4];      // p        at (input row, input column)
    float4 gin[4];     // g1
    float4 o2[4], o3[4];  // g2, g3 at (output row, output column)
    float2 o4[4];
#define PSM_ISSUE_A(SLOT, STEP)                                                         \
    {                                                                                   \
        const size_t off_ = (size_t)r101c(ybase + (STEP), H) * W + pos.cs;              \
        pin[SLOT] = vd[off_];                                                           \
        gin[SLOT] = G1[off_];                                                           \
        int yo_ = ybase + (STEP) - 3;                                                   \
        yo_ = yo_ < 0 ? 0 : (yo_ > H - 1 ? H - 1 : yo_);                                \
        const size_t oo_ = (size_t)yo_ * W + xoc;                                       \
        o2[SLOT] = G2[oo_];                                                             \
        o3[SLOT] = G3[oo_];                                                             \
        o4[SLOT] = G4[oo_];                                                             \
    }
    PSM_ISSUE_A(0, 0) __builtin_amdgcn_sched_barrier(0);
    PSM_ISSUE_A(1, 1) __builtin_amdgcn_sched_barrier(0);
    PSM_ISSUE_A(2, 2) __builtin_amdgcn_sched_barrier(0);
    for (int i = 0; i < n; i += 4) {
#define PSM_STEP_A(K)                                                                               \
    {                                                                                               \
        const int step = i + K;                                                                     \
        PSM_ISSUE_A((K + 3) & 3, step + 3)                                                          \
        const float p = pin[K];                                                                     \
        double h0 = hsum8(p, i1, i2, i4);                                                           \
        double h1 = hsum8(__fmul_rn(gin[K].x, p), i1, i2, i4);                                      \
        double h2 = hsum8(__fmul_rn(gin[K].y, p), i1, i2, i4);                                      \
        double h3 = hsum8(__fmul_rn(gin[K].z, p), i1, i2, i4);                                      \
        double n0 = vstep<K>(t0, h0), n1 = vstep<K>(t1, h1), n2 = vstep<K>(t2, h2), n3 = vstep<K>(t3, h3); \
        float4 r = solve_ab(box_out(n0), box_out(n1), box_out(n2), box_out(n3), o2[K], o3[K], o4[K]); \
        if (step >= 7 && step < n && pos.ovalid) {                                                  \
            abd[(size_t)(ybase + step - 3) * W + pos.xo] = r;                                       \
        }                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                          \
    }
        PSM_STEP_A(0) PSM_STEP_A(1) PSM_STEP_A(2) PSM_STEP_A(3)
#undef PSM_STEP_A
    }
#undef PSM_ISSUE_A
}

// ---- plain box filter: 4-byte outputs -----------------------------------------------------------
// A wave-row of 56 floats is 224 bytes: 1.75 cache lines at an odd offset.  Written straight from
// the lanes it costs partial-line writes plus a fill read per line (measured ~3 TB/s).  So the four
// waves of a workgroup take four ADJACENT strips of one slice (224 columns = 7 full 128-byte lines),
// park four rows of results in LDS and each wave then writes one merged row with 16-byte lane
// stores: every global store covers whole, aligned lines.
constexpr int X4_COLS = 4 * OUT_PER_WAVE;  // 224 output columns per workgroup

struct MarchPosX4 {
    int d, lane, wave, cs, xo, xg, y0, y1;
    bool ok;      // workgroup has work (uniform over the workgroup)
    bool ovalid;  // this lane produces an output column
};
__device__ __forceinline__ MarchPosX4 march_pos_x4(int W, int ybeg, int yend, int Dloc, int ngroups, int nsegs, int seg_rows)
{
    MarchPosX4 p;
    int id = blockIdx.x;
    int g = id % ngroups, rest = id / ngroups;   // strip groups fastest: neighbours in x run together
    p.d = rest % Dloc;
    int seg = rest / Dloc;
    p.ok = seg < nsegs;
    p.wave = threadIdx.x >> 6;
    p.lane = threadIdx.x & 63;
    p.xg = g * X4_COLS;
    int x0 = p.xg + p.wave * OUT_PER_WAVE;
    p.cs = r101c(x0 - 4 + p.lane, W);
    p.xo = x0 + p.lane;
    p.ovalid = p.lane < OUT_PER_WAVE && p.xo < W;
    p.y0 = ybeg + seg * seg_rows;
    p.y1 = min(yend, p.y0 + seg_rows);
    return p;
}

// rows[4][224] of the current 4-step batch -> global.  Called by all 256 threads.
template <bool VEC4>
__device__ __forceinline__ void flush_rows_x4(float *lds_buf, const float (&qb)[4], const MarchPosX4 &pos, float *vd,
                                              int W, int ybase, int i, int n)
{
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (pos.lane < OUT_PER_WAVE) lds_buf[k * X4_COLS + pos.wave * OUT_PER_WAVE + pos.lane] = qb[k];
    __syncthreads();
    const int step = i + pos.wave;  // wave w writes the row produced at step i+w
    if (step >= 7 && step < n) {
        float *row = vd + (size_t)(ybase + step - 3) * W + pos.xg;
        if (VEC4) {
            int c = pos.lane * 4;
            if (pos.lane < X4_COLS / 4 && pos.xg + c < W)
                *reinterpret_cast<float4 *>(row + c) = *reinterpret_cast<const float4 *>(lds_buf + pos.wave * X4_COLS + c);
        } else {
#pragma unroll
            for (int c = pos.lane; c < X4_COLS; c += 64)
                if (pos.xg + c < W) row[c] = lds_buf[pos.wave * X4_COLS + c];
        }
    }
}

// ---- plain box filter of every slice ----------------------------------------------------------
template <bool VEC4>
__global__ __launch_bounds__(256) void k_box8(const float *__restrict__ vol, float *__restrict__ out, int W,
                                             int H, int Dloc, int ngroups, int nsegs, int seg_rows)
{
    __shared__ __attribute__((aligned(16))) float lds[2][4 * X4_COLS];
    const MarchPosX4 pos = march_pos_x4(W, 0, H, Dloc, ngroups, nsegs, seg_rows);
    if (!pos.ok) return;
    PSM_LANE_IDX();
    const size_t HW = (size_t)H * W;
    const float *vd = vol + (size_t)pos.d * HW;
    float *od = out + (size_t)pos.d * HW;
    VTree t0 = {};
    const int n = (pos.y1 - pos.y0) + 7;
    const int ybase = pos.y0 - 4;
    float pv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) pv[k] = vd[(size_t)r101c(ybase + k, H) * W + pos.cs];
    for (int i = 0; i < n; i += 8) {
        float pc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) pc[k] = pv[k];
#pragma unroll
        for (int k = 0; k < 8; ++k) pv[k] = vd[(size_t)r101c(ybase + i + 8 + k, H) * W + pos.cs];
        float qa[4], qc[4];
#define PSM_STEP_X(K, Q) Q[K & 3] = box_out(vstep<K>(t0, hsum8(pc[K], i1, i2, i4)));
        PSM_STEP_X(0, qa) PSM_STEP_X(1, qa) PSM_STEP_X(2, qa) PSM_STEP_X(3, qa)
        flush_rows_x4<VEC4>(lds[0], qa, pos, od, W, ybase, i, n);
        PSM_STEP_X(4, qc) PSM_STEP_X(5, qc) PSM_STEP_X(6, qc) PSM_STEP_X(7, qc)
        flush_rows_x4<VEC4>(lds[1], qc, pos, od, W, ybase, i + 4, n);
#undef PSM_STEP_X
    }
}

// ---- direct (per-voxel) variants: an independent formulation of the same arithmetic, used to
// cross-check the marching kernels (PSM_OPT_KERNEL_VARIANT=1).  64 taps per channel per voxel.
__global__ __launch_bounds__(256) void k_cvf_a_direct(const float *__restrict__ vol, float4 *__restrict__ ab,
                                                     const float4 *__restrict__ G1, const float4 *__restrict__ G2,
                                                     const float4 *__restrict__ G3, const float2 *__restrict__ G4,
                                                     int W, int H)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, d = blockIdx.z;
    if (x >= W) return;
    const size_t HW = (size_t)H * W;
    const float *vd = vol + (size_t)d * HW;
    int rx[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) rx[i] = r101(x - 4 + i, W);
    double hs[4][8];
    for (int j = 0; j < 8; ++j) {
        size_t row = (size_t)r101(y - 4 + j, H) * W;
        double t[4][8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float p = vd[row + rx[i]];
            float4 g = G1[row + rx[i]];
            t[0][i] = p;
            t[1][i] = __fmul_rn(g.x, p);
            t[2][i] = __fmul_rn(g.y, p);
            t[3][i] = __fmul_rn(g.z, p);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
            hs[c][j] = t8(t[c][0], t[c][1], t[c][2], t[c][3], t[c][4], t[c][5], t[c][6], t[c][7]);
    }
    float m[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
        m[c] = box_out(t8(hs[c][0], hs[c][1], hs[c][2], hs[c][3], hs[c][4], hs[c][5], hs[c][6], hs[c][7]));
    size_t oo = (size_t)y * W + x;
    ab[(size_t)d * HW + oo] = solve_ab(m[0], m[1], m[2], m[3], G2[oo], G3[oo], G4[oo]);
}

__global__ __launch_bounds__(256) void k_cvf_b_direct(const float4 *__restrict__ ab, float *__restrict__ vol,
                                                     const float4 *__restrict__ G1, int W, int H)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, d = blockIdx.z;
    if (x >= W) return;
    const size_t HW = (size_t)H * W;
    const float4 *abd = ab + (size_t)d * HW;
    int rx[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) rx[i] = r101(x - 4 + i, W);
    double hs[4][8];
    for (int j = 0; j < 8; ++j) {
        size_t row = (size_t)r101(y - 4 + j, H) * W;
        double t[4][8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float4 a = abd[row + rx[i]];
            t[0][i] = a.x; t[1][i] = a.y; t[2][i] = a.z; t[3][i] = a.w;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
            hs[c][j] = t8(t[c][0], t[c][1], t[c][2], t[c][3], t[c][4], t[c][5], t[c][6], t[c][7]);
    }
    float m[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
        m[c] = box_out(t8(hs[c][0], hs[c][1], hs[c][2], hs[c][3], hs[c][4], hs[c][5], hs[c][6], hs[c][7]));
    size_t oo = (size_t)y * W + x;
    vol[(size_t)d * HW + oo] = recombine(m[0], m[1], m[2], m[3], G1[oo]);
}

__global__ __launch_bounds__(256) void k_box8_direct(const float *__restrict__ vol, float *__restrict__ out, int W, int H)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, d = blockIdx.z;
    if (x >= W) return;
    const size_t HW = (size_t)H * W;
    const float *vd = vol + (size_t)d * HW;
    double hs[8];
    for (int j = 0; j < 8; ++j) {
        size_t row = (size_t)r101(y - 4 + j, H) * W;
        double t[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = vd[row + r101(x - 4 + i, W)];
        hs[j] = t8(t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7]);
    }
    out[(size_t)d * HW + (size_t)y * W + x] = box_out(t8(hs[0], hs[1], hs[2], hs[3], hs[4], hs[5], hs[6], hs[7]));
}

struct MarchGrid {
    int nstrips, nsegs, seg_rows, nzg, nblocks, nxcd;
};
static MarchGrid march_grid(March m, int W, int H, int Dloc)
{   // H = number of output rows of this launch
    MarchGrid g;
    g.nstrips = (W + OUT_PER_WAVE - 1) / OUT_PER_WAVE;
    // auto: split H evenly into segments of about 128 rows (7 halo rows each: ~5 % extra row
    // reads, but 8x more blocks than whole columns -> no tail at 1080p x 256; measured best)
    if (m.seg_rows > 0) g.seg_rows = m.seg_rows;
    else { int k = (H + 127) / 128; g.seg_rows = (H + k - 1) / k; }
    if (g.seg_rows > H) g.seg_rows = H;
    g.nsegs = (H + g.seg_rows - 1) / g.seg_rows;
    g.nzg = (Dloc + m.waves - 1) / m.waves;
    g.nxcd = pc_dev().nxcd;
    g.nblocks = g.nxcd * ((g.nstrips * g.nsegs + g.nxcd - 1) / g.nxcd) * g.nzg;
    return g;
}

#define PSM_DISPATCH_NW(NWV, KERNEL, ...)                                                               \
    switch (NWV) {                                                                                      \
    case 1: hipLaunchKernelGGL(KERNEL<1>, dim3(g.nblocks), dim3(64), 0, s, __VA_ARGS__); break;         \
    case 2: hipLaunchKernelGGL(KERNEL<2>, dim3(g.nblocks), dim3(128), 0, s, __VA_ARGS__); break;        \
    case 8: hipLaunchKernelGGL(KERNEL<8>, dim3(g.nblocks), dim3(512), 0, s, __VA_ARGS__); break;        \
    default: hipLaunchKernelGGL(KERNEL<4>, dim3(g.nblocks), dim3(256), 0, s, __VA_ARGS__); break;       \
    }

static int norm_waves(int w) { return (w == 1 || w == 2 || w == 8) ? w : 4; }

void launch_cvf_a(hipStream_t s, int variant, March m, const float *vol, float4 *ab, Guidance gd, int W, int H, int Dloc,
                  int ybeg, int yend)
{
    if (yend <= ybeg) return;
    if (variant == 1) {
        dim3 grid((W + 255) / 256, H, Dloc);
        hipLaunchKernelGGL(k_cvf_a_direct, grid, dim3(256), 0, s, vol, ab, (const float4 *)gd.g1, (const float4 *)gd.g2,
                           (const float4 *)gd.g3, (const float2 *)gd.g4, W, H);
        return;
    }
    m.waves = norm_waves(m.waves);
    MarchGrid g = march_grid(m, W, yend - ybeg, Dloc);
    PSM_DISPATCH_NW(m.waves, k_cvf_a, vol, ab, (const float4 *)gd.g1, (const float4 *)gd.g2, (const float4 *)gd.g3,
                    (const float2 *)gd.g4, W, H, Dloc, g.nstrips, g.nsegs, g.seg_rows, g.nzg, g.nxcd, ybeg, yend)
}

void launch_cvf_b_direct(hipStream_t s, const float4 *ab, float *vol, Guidance gd, int W, int H, int Dloc)
{
    dim3 grid((W + 255) / 256, H, Dloc);
    hipLaunchKernelGGL(k_cvf_b_direct, grid, dim3(256), 0, s, ab, vol, (const float4 *)gd.g1, W, H);
}

void launch_box8(hipStream_t s, int variant, March m, const float *vol, float *out, int W, int H, int Dloc)
{
    if (variant == 1) {
        dim3 grid((W + 255) / 256, H, Dloc);
        hipLaunchKernelGGL(k_box8_direct, grid, dim3(256), 0, s, vol, out, W, H);
        return;
    }
    MarchGrid g = march_grid(m, W, H, Dloc);
    const int ngroups = (W + X4_COLS - 1) / X4_COLS;
    const int nblocks = ngroups * Dloc * g.nsegs;
    if ((W & 3) == 0)
        hipLaunchKernelGGL(k_box8<true>, dim3(nblocks), dim3(256), 0, s, vol, out, W, H, Dloc, ngroups, g.nsegs, g.seg_rows);
    else
        hipLaunchKernelGGL(k_box8<false>, dim3(nblocks), dim3(256), 0, s, vol, out, W, H, Dloc, ngroups, g.nsegs, g.seg_rows);
}

// ------------------------------------------------------------------------------------------
// DispSel: WTA (src/DispSel.cpp:83-109) over the local slices, with global semantics
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_wta(const float *__restrict__ vol, int HW, int d_begin, int Dloc,
                                            long long *__restrict__ keys, uint8_t *__restrict__ map)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= HW) return;
    float minCost = __builtin_inff();
    int minDis = 0;
    int dl = (d_begin == 0) ? 1 : 0;  // d = 0 is never a candidate (src/DispSel.cpp:96)
    const float *p = vol + (size_t)dl * HW + i;
#pragma unroll 8
    for (; dl < Dloc; ++dl, p += HW) {
        float c = __builtin_nontemporal_load(p);   // streamed once: 7.1 vs 6.4 TB/s in a read microbenchmark
        if (c < minCost) {
            minCost = c;
            minDis = d_begin + dl;
        }
    }
    if (keys) keys[i] = pack_key_f32(minCost, minDis);
    if (map) map[i] = (uint8_t)minDis;
}

void launch_wta(hipStream_t s, const float *vol, int W, int H, int d_begin, int Dloc, long long *keys, uint8_t *map)
{
    int HW = W * H;
    hipLaunchKernelGGL(k_wta, dim3((HW + 255) / 256), dim3(256), 0, s, vol, HW, d_begin, Dloc, keys, map);
}

__global__ __launch_bounds__(256) void k_merge(const long long *__restrict__ keys_all, size_t rank_stride, int nranks,
                                              int n, uint8_t *__restrict__ map)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    long long best = keys_all[i];
    for (int r = 1; r < nranks; ++r) {
        long long k = keys_all[(size_t)r * rank_stride + i];
        best = k < best ? k : best;
    }
    map[i] = (uint8_t)((unsigned long long)best & 0xffffffffull);
}

void launch_merge(hipStream_t s, const long long *keys_all, size_t rank_stride, int nranks, int n, uint8_t *map)
{
    hipLaunchKernelGGL(k_merge, dim3((n + 255) / 256), dim3(256), 0, s, keys_all, rank_stride, nranks, n, map);
}

// keys -> maps of every pair of a batch (the two-phase selection leaves keys; blockIdx.y = pair)
__global__ __launch_bounds__(256) void k_merge_batch(const PcPair *__restrict__ tab, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    tab[blockIdx.y].maps[i] = (uint8_t)((unsigned long long)tab[blockIdx.y].keys[i] & 0xffull);
}
void launch_merge_batch(hipStream_t s, const PcPair *tab, int npairs, int W, int H)
{
    const int n = 2 * W * H;
    hipLaunchKernelGGL(k_merge_batch, dim3((n + 255) / 256, npairs), dim3(256), 0, s, tab, n);
}

// ------------------------------------------------------------------------------------------
// PP lrCheck (src/PP.cpp:17-50)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lr_check(const uint8_t *__restrict__ l, const uint8_t *__restrict__ r, int W, int H,
                                                 uint8_t *__restrict__ lv, uint8_t *__restrict__ rv)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const uint8_t *lr = l + (size_t)y * W, *rr = r + (size_t)y * W;
    int lDep = lr[x];
    int rLoc = (x - lDep + W) % W;
    int rDep = rr[rLoc];
    lv[(size_t)y * W + x] = (lDep == rDep && lDep >= 2) ? 1 : 0;
    rDep = rr[x];
    int lLoc = (x + rDep + W) % W;
    lDep = lr[lLoc];
    rv[(size_t)y * W + x] = (rDep == lDep && rDep >= 2) ? 1 : 0;
}

// ------------------------------------------------------------------------------------------
// PP fillInv (src/PP.cpp:52-143): every invalid pixel takes the smaller disparity of its nearest
// valid neighbours to the left and to the right in the same row (only valid pixels are read, so the
// result does not depend on the order).
// ------------------------------------------------------------------------------------------
// Round 6: one WAVE per row and map, both maps per launch.  The nearest valid pixel on either side comes from the wave's ballot
// of the validity bytes (highest set bit below / lowest above the lane, a carry across the 64-pixel chunks), its value through a
// lane read: no LDS scan, no barrier (the block scan of rounds 2 - 5 went through 2 x 8 x 16 barriers per row: 39 us a map at
// 1080p).  The row is read once, 16 chunks' loads in flight at a time; the pass from the right works from LDS.
__global__ __launch_bounds__(64) void k_fill_inv(uint8_t *__restrict__ dis0, const uint8_t *__restrict__ valid0, size_t side, int W)
{
    extern __shared__ short row[];        // [W] the pixel's value, -1: invalid | [W] value of the nearest valid pixel to the left, -1: none
    short *sv = row, *lval = row + W;
    const int y = blockIdx.x, lane = threadIdx.x;
    uint8_t *d = dis0 + blockIdx.y * side + (size_t)y * W;
    const uint8_t *v = valid0 + blockIdx.y * side + (size_t)y * W;
    const unsigned long long below = (1ull << lane) - 1ull, above = lane == 63 ? 0ull : ~((2ull << lane) - 1ull);
    constexpr int NC = 16;
    int carry = -1;
    for (int xb = 0; xb < W; xb += 64 * NC) {
        int vb[NC], db[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const int x = xb + 64 * k + lane;
            vb[k] = x < W ? v[x] : 0;
            db[k] = x < W ? d[x] : 0;
        }
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const int x = xb + 64 * k + lane;
            const bool ok = vb[k] != 0;             // (chunks beyond the row: nothing valid, nothing stored)
            const unsigned long long m = __builtin_amdgcn_ballot_w64(ok);
            const unsigned long long lo = m & below;
            const int got = __shfl(db[k], lo ? 63 - __builtin_clzll(lo) : 0);
            if (x < W) {
                sv[x] = (short)(ok ? db[k] : -1);
                lval[x] = (short)(lo ? got : carry);
            }
            if (m) carry = __builtin_amdgcn_readlane(db[k], 63 - __builtin_clzll(m));
        }
    }
    carry = -1;                             // from here on: the nearest valid value to the right
    for (int x0 = ((W - 1) / 64) * 64; x0 >= 0; x0 -= 64) {
        const int x = x0 + lane;
        const int val = x < W ? sv[x] : -1;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(val >= 0);
        const unsigned long long hi = m & above;
        const int got = __shfl(val, hi ? __builtin_ctzll(hi) : 0);
        const int r = hi ? got : carry;
        if (x < W && val < 0) {
            const int l = lval[x];
            if (l >= 0 && r >= 0) d[x] = (uint8_t)(l <= r ? l : r);
            else if (l >= 0) d[x] = (uint8_t)l;
            else if (r >= 0) d[x] = (uint8_t)r;
        }
        if (m) carry = __builtin_amdgcn_readlane(val, __builtin_ctzll(m));
    }
}

// both maps of a pair: dis / valid of the right map lie `side` bytes behind the left map's
void launch_fill_inv(hipStream_t s, uint8_t *dis, const uint8_t *valid, size_t side, int W, int H)
{
    hipLaunchKernelGGL(k_fill_inv, dim3(H, 2), dim3(64), 2 * (size_t)W * sizeof(short), s, dis, valid, side, W);
}

void launch_lr_check(hipStream_t s, const uint8_t *l, const uint8_t *r, int W, int H, uint8_t *lv, uint8_t *rv)
{
    hipLaunchKernelGGL(k_lr_check, dim3((W + 255) / 256, H), dim3(256), 0, s, l, r, W, H, lv, rv);
}

// ------------------------------------------------------------------------------------------
// 8-bit char mode (build-defined contract, DESIGN.md "8-bit mode"; oracle: psmo_*_u8)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int gray_u8(const uint8_t *p)
{
    return (p[0] * 4899 + p[1] * 9617 + p[2] * 1868 + (1 << 13)) >> 14;
}

// planes4[y][x] = {c0, c1, c2, grad} as one 32-bit word per pixel
__global__ __launch_bounds__(256) void k_prep_u8(const uint8_t *src, size_t pitch, int W, int H, uchar4 *out, const PcPair *__restrict__ tab, float4 *g1)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (tab) {
        src = (const uint8_t *)tab[blockIdx.z >> 1].raw[blockIdx.z & 1]; out = (uchar4 *)tab[blockIdx.z >> 1].p4[blockIdx.z & 1];
        g1 = tab[blockIdx.z >> 1].g[blockIdx.z & 1].g1;
    }
    if (x >= W) return;
    const uint8_t *row = src + (size_t)y * pitch;
    const uint8_t *p = row + 3 * x;
    int g = gray_u8(row + 3 * r101(x + 1, W)) - gray_u8(row + 3 * r101(x - 1, W));
    g = g < 0 ? 0 : (g > 255 ? 255 : g);
    const uchar4 w = make_uchar4(p[0], p[1], p[2], (uint8_t)g);
    out[(size_t)y * W + x] = w;
    // 8-bit char mode: the fused kernel's producer waves load g1 of their own pixel anyway (the I.p products) and never use its
    // float gradient - the pixel's four bytes ride in that slot (bit pattern), which saves them a load instruction per step
    reinterpret_cast<uchar4 *>(&g1[(size_t)y * W + x])[3] = w;
}
void launch_prep_u8(hipStream_t s, const uint8_t *src, size_t pitch, int W, int H, uint8_t *planes4, float4 *g1)
{
    hipLaunchKernelGGL(k_prep_u8, dim3((W + 255) / 256, H), dim3(256), 0, s, src, pitch, W, H, (uchar4 *)planes4, (const PcPair *)nullptr, g1);
}
void launch_prep_batch(hipStream_t s, const PcPair *tab, int npairs, size_t pitch, int depth_f32, int W, int H, bool u8_planes)
{
    const dim3 grid((W + 255) / 256, H, 2 * npairs);
    if (depth_f32)
        hipLaunchKernelGGL(k_prep<true>, grid, dim3(256), 0, s, (const void *)nullptr, pitch, W, H, (float4 *)nullptr, (const void *)nullptr, (float4 *)nullptr, tab);
    else
        hipLaunchKernelGGL(k_prep<false>, grid, dim3(256), 0, s, (const void *)nullptr, pitch, W, H, (float4 *)nullptr, (const void *)nullptr, (float4 *)nullptr, tab);
    if (u8_planes)
        hipLaunchKernelGGL(k_prep_u8, grid, dim3(256), 0, s, (const uint8_t *)nullptr, pitch, W, H, (uchar4 *)nullptr, tab, (float4 *)nullptr);
}

__device__ __forceinline__ uint8_t cost_u8(int clr3, int grd)
{  // assets/cvc.cl:279-301
    float f = __fadd_rn(__fmul_rn(0.9f, (float)(clr3 / 3)), __fmul_rn(__fsub_rn(1.0f, 0.9f), (float)grd));
    return (uint8_t)f;
}
template <bool RIGHT>
__global__ __launch_bounds__(256) void k_cvc_u8(const uchar4 *__restrict__ base, const uchar4 *__restrict__ other,
                                               uint8_t *__restrict__ vol, int W, int H, int d_begin, int Dloc)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    int dl0 = blockIdx.z * CVC_DC;
    if (x >= W) return;
    const size_t HW = (size_t)H * W, row = (size_t)y * W;
    uchar4 a = base[row + x];
    uint8_t cb = cost_u8(abs(a.x - 255) + abs(a.y - 255) + abs(a.z - 255), abs(a.w - 255));
#pragma unroll
    for (int k = 0; k < CVC_DC; ++k) {
        int dl = dl0 + k;
        if (dl >= Dloc) break;
        int d = d_begin + dl;
        bool in = RIGHT ? (x < W - d) : (x >= d);
        uint8_t c = cb;
        if (in) {
            uchar4 b = other[row + (RIGHT ? x + d : x - d)];
            c = cost_u8(abs(a.x - b.x) + abs(a.y - b.y) + abs(a.z - b.z), abs(a.w - b.w));
        }
        vol[(size_t)dl * HW + row + x] = c;
    }
}
void launch_cvc_u8(hipStream_t s, const uint8_t *base4, const uint8_t *other4, uint8_t *vol, int W, int H, int d_begin,
                   int Dloc, int right)
{
    dim3 grid((W + 255) / 256, H, (Dloc + CVC_DC - 1) / CVC_DC);
    if (right)
        hipLaunchKernelGGL(k_cvc_u8<true>, grid, dim3(256), 0, s, (const uchar4 *)base4, (const uchar4 *)other4, vol, W, H, d_begin, Dloc);
    else
        hipLaunchKernelGGL(k_cvc_u8<false>, grid, dim3(256), 0, s, (const uchar4 *)base4, (const uchar4 *)other4, vol, W, H, d_begin, Dloc);
}

__global__ __launch_bounds__(256) void k_u8_to_f32(const uint8_t *__restrict__ src, float *__restrict__ dst, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const float alpha = 1 / 255.0f;
    for (; i < n; i += stride) dst[i] = __fmul_rn((float)src[i], alpha);
}
__global__ __launch_bounds__(256) void k_f32_to_u8(const float *__restrict__ src, uint8_t *__restrict__ dst, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float r = rintf(__fmul_rn(src[i], 255.0f));  // round-half-even; NaN -> 0
        dst[i] = !(r > 0.0f) ? 0 : (r > 255.0f ? 255 : (uint8_t)r);
    }
}
static int grid_for(size_t n)
{
    size_t b = (n + 255) / 256;
    return (int)(b > 8192 ? 8192 : (b ? b : 1));
}
void launch_u8_to_f32(hipStream_t s, const uint8_t *src, float *dst, size_t n)
{
    hipLaunchKernelGGL(k_u8_to_f32, dim3(grid_for(n)), dim3(256), 0, s, src, dst, n);
}
void launch_f32_to_u8(hipStream_t s, const float *src, uint8_t *dst, size_t n)
{
    hipLaunchKernelGGL(k_f32_to_u8, dim3(grid_for(n)), dim3(256), 0, s, src, dst, n);
}

__global__ __launch_bounds__(256) void k_wta_u8(const uint8_t *__restrict__ vol, int HW, int d_begin, int Dloc,
                                               long long *__restrict__ keys, uint8_t *__restrict__ map)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= HW) return;
    int minCost = 256, minDis = 0;  // assets/dispsel.cl:41-62, initial minimum 256 so 255 can win
    int dl = (d_begin == 0) ? 1 : 0;
    const uint8_t *p = vol + (size_t)dl * HW + i;
    for (; dl < Dloc; ++dl, p += HW) {
        int c = *p;
        if (c < minCost) {
            minCost = c;
            minDis = d_begin + dl;
        }
    }
    // same key format as the float path / the fused select kernel (which carries q8 as a float): shards may mix both
    if (keys) keys[i] = pack_key_f32(minCost == 256 ? __builtin_inff() : (float)minCost, minDis);
    if (map) map[i] = (uint8_t)minDis;
}
void launch_wta_u8(hipStream_t s, const uint8_t *vol, int W, int H, int d_begin, int Dloc, long long *keys, uint8_t *map)
{
    int HW = W * H;
    hipLaunchKernelGGL(k_wta_u8, dim3((HW + 255) / 256), dim3(256), 0, s, vol, HW, d_begin, Dloc, keys, map);
}

}  // namespace psm
