/*
 * psm_oracle.c - CPU restatement of the DispEst hot path (CVC -> CVF -> DispSel).
 * TEST INFRASTRUCTURE ONLY - see psm_oracle.h.  "parity unpinned" (no reference binary,
 * tests or golden vectors exist; OpenCV semantics per SURVEY.md Appendix A).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fPIC -shared -pthread (oracle/Makefile).
 */
#include "psm_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------------- */
/* helpers                                                                          */
/* ------------------------------------------------------------------------------- */

/* cv::BORDER_REFLECT_101 index map (gfedcb|abcdefgh|gfedcba). */
static inline int r101(int k, int n)
{
    if (k < 0) k = -k;
    if (k >= n) k = 2 * (n - 1) - k;
    return k;
}

static double now_ms(void)
{ /* include/ComFunc.h:67-71 get_rt(), kept in double */
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec * 1e3 + (double)ts.tv_nsec * 1e-6;
}

#define T8(t0, t1, t2, t3, t4, t5, t6, t7) \
    ((((t0) + (t1)) + ((t2) + (t3))) + (((t4) + (t5)) + ((t6) + (t7))))

/* ------------------------------------------------------------------------------- */
/* input conditioning                                                               */
/* ------------------------------------------------------------------------------- */

void psmo_u8_to_f32(const uint8_t *src, size_t n, float *dst)
{
    /* src/StereoMatch.cpp:195: convertTo(CV_32F, 1/255.0f) -> (float)u8 * alpha, fp32 */
    const float alpha = 1 / 255.0f;
    for (size_t i = 0; i < n; ++i) dst[i] = (float)src[i] * alpha;
}

/* ------------------------------------------------------------------------------- */
/* CVC (float)                                                                      */
/* ------------------------------------------------------------------------------- */

#define BC_32F 1.0     /* include/CVC.h:12  (a double literal) */
#define ALPHA_32F 0.9f /* include/CVC.h:23 */

/* Alternative READINGS of two reference lines whose meaning depends on the toolchain that compiles the reference
 * (psmo_set_variant; never the canon - tests/test_oracle.py uses them to put a number on how far "the reference binary"
 * can be from the canonical arithmetic):
 *   PSMO_VAR_FABS_DOUBLE  src/CVC.cpp:21-23 calls unqualified fabs() on float differences with only <math.h> in scope
 *                         (include/ComFunc.h:18): with libstdc++ >= 6 that resolves to the float overload (canon); with the
 *                         2016-era toolchains the project targeted it is ::fabs(double) - the three terms are doubles, their
 *                         sum is formed in double and rounded once when assigned to `float clrDiff`.
 *   PSMO_VAR_FMA_SOLVE    the scalar solve loop src/CVF.cpp:129-147 compiled with -ffp-contract=fast on an FMA target (GCC's
 *                         default outside ISO mode; ARM VFPv4 / NEON boards are what the project ran on): every a*b-c*d /
 *                         a*b+c*d there becomes a fused multiply-add. */
static int g_variant = 0;
void psmo_set_variant(int bits) { g_variant = bits; }
int psmo_get_variant(void) { return g_variant; }

/* src/CVC.cpp:18-27 */
static inline float myCostGrd2(const float *lC, const float *rC, const float *lG, const float *rG)
{
    if (g_variant & PSMO_VAR_FABS_DOUBLE) {
        float cd = (float)(fabs((double)(lC[0] - rC[0])) + fabs((double)(lC[1] - rC[1])) + fabs((double)(lC[2] - rC[2])));
        float gd = (float)fabs((double)(*lG - *rG));
        return ALPHA_32F * cd + (1 - ALPHA_32F) * gd;
    }
    float clrDiff = fabsf(lC[0] - rC[0]) + fabsf(lC[1] - rC[1]) + fabsf(lC[2] - rC[2]);
    float grdDiff = fabsf(*lG - *rG);
    return ALPHA_32F * clrDiff + (1 - ALPHA_32F) * grdDiff;
}

/* src/CVC.cpp:30-39: BC_32F is a double, so the differences and their sum are double and
 * are rounded once when assigned to the float locals. */
static inline float myCostGrd1(const float *lC, const float *lG)
{
    float clrDiff = (float)(fabs(lC[0] - BC_32F) + fabs(lC[1] - BC_32F) + fabs(lC[2] - BC_32F));
    float grdDiff = (float)fabs(*lG - BC_32F);
    return ALPHA_32F * clrDiff + (1 - ALPHA_32F) * grdDiff;
}

void psmo_cvc_preprocess(const float *img, int H, int W, float *grdx)
{
    /* src/CVC.cpp:43: cvtColor(Img, GrdX, CV_RGB2GRAY) on BGR data: 0.299 multiplies c0.
     * src/CVC.cpp:44: Sobel(GrdX, GrdX, CV_32F, 1, 0, 1): [-1 0 1], REFLECT_101. */
    float *gray = (float *)malloc((size_t)H * W * sizeof(float));
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const float *c = img + ((size_t)y * W + x) * 3;
            gray[(size_t)y * W + x] = (c[0] * 0.299f + c[1] * 0.587f) + c[2] * 0.114f;
        }
    for (int y = 0; y < H; ++y) {
        const float *g = gray + (size_t)y * W;
        for (int x = 0; x < W; ++x) grdx[(size_t)y * W + x] = g[r101(x + 1, W)] - g[r101(x - 1, W)];
    }
    free(gray);
}

void psmo_cvc_build_left(const float *lImg, const float *rImg, const float *lGrdX,
                         const float *rGrdX, int H, int W, int d, float *cost)
{
    /* src/CVC.cpp:127-147 */
    for (int y = 0; y < H; ++y) {
        const float *lData = lImg + (size_t)y * W * 3;
        const float *rData = rImg + (size_t)y * W * 3;
        const float *lGData = lGrdX + (size_t)y * W;
        const float *rGData = rGrdX + (size_t)y * W;
        float *c = cost + (size_t)y * W;
        for (int x = d; x < W; ++x)
            c[x] = myCostGrd2(lData + 3 * x, rData + 3 * (x - d), lGData + x, rGData + x - d);
        for (int x = 0; x < d && x < W; ++x) c[x] = myCostGrd1(lData + 3 * x, lGData + x);
    }
}

void psmo_cvc_build_right(const float *lImg, const float *rImg, const float *lGrdX,
                          const float *rGrdX, int H, int W, int d, float *cost)
{
    /* src/CVC.cpp:157-177 */
    int border = W - d;
    if (border < 0) border = 0;
    for (int y = 0; y < H; ++y) {
        const float *lData = lImg + (size_t)y * W * 3;
        const float *rData = rImg + (size_t)y * W * 3;
        const float *lGData = lGrdX + (size_t)y * W;
        const float *rGData = rGrdX + (size_t)y * W;
        float *c = cost + (size_t)y * W;
        for (int x = 0; x < border; ++x)
            c[x] = myCostGrd2(lData + 3 * x, rData + 3 * (x + d), lGData + x, rGData + x + d);
        for (int x = border; x < W; ++x) c[x] = myCostGrd1(lData + 3 * x, lGData + x);
    }
}

/* ------------------------------------------------------------------------------- */
/* CVF                                                                              */
/* ------------------------------------------------------------------------------- */

/* Which summation order the box filters use (psmo_set_box_order):
 *   PSMO_BOX_TREE (default) - the canonical balanced tree of psm_oracle.h (segment independent; what the HIP kernels
 *                             evaluate bit for bit);
 *   PSMO_BOX_OCV            - the order OpenCV's own engines execute for a CV_32F boxFilter / blur
 *                             (modules/imgproc box_filter: RowSum<float,double> + ColumnSum<double,float>):
 *                             a running sum along each row, s += (double)S[i+k] - (double)S[i], and a running column
 *                             accumulator SUM over the whole image height, out = (float)((SUM + newest) * scale),
 *                             SUM = (SUM + newest) - oldest.
 * Read by every thread of the pipelines; set it before starting one. */
static int g_box_order = PSMO_BOX_TREE;
void psmo_set_box_order(int order) { g_box_order = order == PSMO_BOX_OCV ? PSMO_BOX_OCV : PSMO_BOX_TREE; }
int psmo_get_box_order(void) { return g_box_order; }

/* cv::boxFilter / cv::blur with a k x k kernel on a CV_32F plane in OpenCV's own evaluation order.
 * anchor = k/2 (the default anchor (-1,-1)), BORDER_REFLECT_101, normalised.
 *   RowSum<float,double>::operator() (generic branch; OpenCV >= 3.4 special-cases ksize 3 and 5 as a plain
 *   left-to-right sum per output, which is what `direct_rows` selects):
 *       s = 0; for i < k: s += (double)S[i];  D[0] = s;  then  s += (double)S[i+k] - (double)S[i];  D[i+1] = s
 *     on the border-extended row S (FilterEngine pads `anchor` pixels left, k-1-anchor right).
 *   ColumnSum<double,float>::operator(): SUM = 0; the first k-1 (border-extended) rows are added top to bottom;
 *     per output row: s0 = SUM + Sp; D = (float)(s0 * scale); SUM = s0 - Sm.   scale = 1./(k*k) (double). */
static void box_ocv(const float *src, int H, int W, int k, float *dst, double *hs)
{
    const int an = k / 2;
    const double scale = 1. / ((double)k * k);
    const int direct_rows = (k == 3 || k == 5);
    double *ext = (double *)malloc((size_t)(W + k) * sizeof(double));
    for (int y = 0; y < H; ++y) {
        const float *s = src + (size_t)y * W;
        double *h = hs + (size_t)y * W;
        for (int i = 0; i < W + k - 1; ++i) ext[i] = (double)s[r101(i - an, W)];
        if (direct_rows) {
            for (int x = 0; x < W; ++x) {
                double a = ext[x];
                for (int i = 1; i < k; ++i) a += ext[x + i];
                h[x] = a;
            }
        } else {
            double a = 0;
            for (int i = 0; i < k; ++i) a += ext[i];
            h[0] = a;
            for (int x = 0; x < W - 1; ++x) {
                a += ext[x + k] - ext[x];
                h[x + 1] = a;
            }
        }
    }
    double *SUM = (double *)calloc((size_t)W, sizeof(double));
    for (int j = 0; j < k - 1; ++j) {
        const double *Sp = hs + (size_t)r101(j - an, H) * W;
        for (int x = 0; x < W; ++x) SUM[x] += Sp[x];
    }
    for (int y = 0; y < H; ++y) {
        const double *Sp = hs + (size_t)r101(y + k - 1 - an, H) * W;
        const double *Sm = hs + (size_t)r101(y - an, H) * W;
        float *o = dst + (size_t)y * W;
        for (int x = 0; x < W; ++x) {
            double s0 = SUM[x] + Sp[x];
            o[x] = (float)(s0 * scale);
            SUM[x] = s0 - Sm[x];
        }
    }
    free(SUM);
    free(ext);
}

/* box with caller-provided double scratch (H*W) so threads do not malloc per call */
static void box8_ws(const float *src, int H, int W, float *dst, double *hs)
{
    if (g_box_order == PSMO_BOX_OCV) { box_ocv(src, H, W, 8, dst, hs); return; }
    for (int y = 0; y < H; ++y) {
        const float *s = src + (size_t)y * W;
        double *h = hs + (size_t)y * W;
        for (int x = 0; x < W; ++x) {
            double t0 = s[r101(x - 4, W)], t1 = s[r101(x - 3, W)], t2 = s[r101(x - 2, W)],
                   t3 = s[r101(x - 1, W)], t4 = s[x], t5 = s[r101(x + 1, W)],
                   t6 = s[r101(x + 2, W)], t7 = s[r101(x + 3, W)];
            h[x] = T8(t0, t1, t2, t3, t4, t5, t6, t7);
        }
    }
    for (int y = 0; y < H; ++y) {
        const double *r0 = hs + (size_t)r101(y - 4, H) * W, *r1 = hs + (size_t)r101(y - 3, H) * W,
                     *r2 = hs + (size_t)r101(y - 2, H) * W, *r3 = hs + (size_t)r101(y - 1, H) * W,
                     *r4 = hs + (size_t)y * W, *r5 = hs + (size_t)r101(y + 1, H) * W,
                     *r6 = hs + (size_t)r101(y + 2, H) * W, *r7 = hs + (size_t)r101(y + 3, H) * W;
        float *o = dst + (size_t)y * W;
        for (int x = 0; x < W; ++x)
            o[x] = (float)(T8(r0[x], r1[x], r2[x], r3[x], r4[x], r5[x], r6[x], r7[x]) * (1.0 / 64));
    }
}

/* Tolerance-form MODELS of the per-slice box filters (psmo_set_variant; never the canon - they put a number on what a cheaper
 * HIP form of the sliding trees would cost in accuracy before any kernel is written, tests/test_oracle.py):
 *   PSMO_VAR_F32_L1   level 1 of the horizontal tree on the fp32 inputs: (double)(float)(t0 + t1) ...
 *   PSMO_VAR_F32_L2   levels 1 and 2 in fp32
 *   PSMO_VAR_RUNCOL   the vertical pass as OpenCV's running ColumnSum (fp64) instead of the balanced tree
 * The guidance precompute (d-invariant) keeps the canonical filter. */
static void box8_slice(const float *src, int H, int W, float *dst, double *hs)
{
    if (!(g_variant & (PSMO_VAR_F32_L1 | PSMO_VAR_F32_L2 | PSMO_VAR_RUNCOL)) || g_box_order == PSMO_BOX_OCV) { box8_ws(src, H, W, dst, hs); return; }
    for (int y = 0; y < H; ++y) {
        const float *s = src + (size_t)y * W;
        double *h = hs + (size_t)y * W;
        for (int x = 0; x < W; ++x) {
            const float t0 = s[r101(x - 4, W)], t1 = s[r101(x - 3, W)], t2 = s[r101(x - 2, W)], t3 = s[r101(x - 1, W)],
                        t4 = s[x], t5 = s[r101(x + 1, W)], t6 = s[r101(x + 2, W)], t7 = s[r101(x + 3, W)];
            if (g_variant & PSMO_VAR_F32_L2) {
                const float a = (t0 + t1) + (t2 + t3), b = (t4 + t5) + (t6 + t7);
                h[x] = (double)a + (double)b;
            } else if (g_variant & PSMO_VAR_F32_L1) {
                const float a = t0 + t1, b = t2 + t3, c = t4 + t5, d = t6 + t7;
                h[x] = ((double)a + (double)b) + ((double)c + (double)d);
            } else h[x] = T8((double)t0, (double)t1, (double)t2, (double)t3, (double)t4, (double)t5, (double)t6, (double)t7);
        }
    }
    if (g_variant & PSMO_VAR_RUNCOL) {
        double *SUM = (double *)calloc((size_t)W, sizeof(double));
        for (int j = 0; j < 7; ++j) {
            const double *Sp = hs + (size_t)r101(j - 4, H) * W;
            for (int x = 0; x < W; ++x) SUM[x] += Sp[x];
        }
        for (int y = 0; y < H; ++y) {
            const double *Sp = hs + (size_t)r101(y + 3, H) * W, *Sm = hs + (size_t)r101(y - 4, H) * W;
            float *o = dst + (size_t)y * W;
            for (int x = 0; x < W; ++x) {
                const double s0 = SUM[x] + Sp[x];
                o[x] = (float)(s0 * (1.0 / 64));
                SUM[x] = s0 - Sm[x];
            }
        }
        free(SUM);
        return;
    }
    for (int y = 0; y < H; ++y) {
        const double *r0 = hs + (size_t)r101(y - 4, H) * W, *r1 = hs + (size_t)r101(y - 3, H) * W,
                     *r2 = hs + (size_t)r101(y - 2, H) * W, *r3 = hs + (size_t)r101(y - 1, H) * W,
                     *r4 = hs + (size_t)y * W, *r5 = hs + (size_t)r101(y + 1, H) * W,
                     *r6 = hs + (size_t)r101(y + 2, H) * W, *r7 = hs + (size_t)r101(y + 3, H) * W;
        float *o = dst + (size_t)y * W;
        for (int x = 0; x < W; ++x)
            o[x] = (float)(T8(r0[x], r1[x], r2[x], r3[x], r4[x], r5[x], r6[x], r7[x]) * (1.0 / 64));
    }
}

void psmo_box8(const float *src, int H, int W, float *dst)
{
    double *hs = (double *)malloc((size_t)H * W * sizeof(double));
    box8_ws(src, H, W, dst, hs);
    free(hs);
}

void psmo_cvf_preprocess(const float *img, int H, int W, float *rgb, float *mean, float *var)
{
    const size_t N = (size_t)H * W;
    float *tmp = (float *)malloc(N * sizeof(float));
    double *hs = (double *)malloc(N * sizeof(double));
    /* src/CVF.cpp:47 split */
    for (size_t i = 0; i < N; ++i)
        for (int c = 0; c < 3; ++c) rgb[c * N + i] = img[i * 3 + c];
    /* src/CVF.cpp:49-51 */
    for (int c = 0; c < 3; ++c) box8_ws(rgb + c * N, H, W, mean + c * N, hs);
    /* src/CVF.cpp:58-68 */
    int varIdx = 0;
    for (int c = 0; c < 3; ++c)
        for (int cp = c; cp < 3; ++cp) {
            float *v = var + (size_t)varIdx * N;
            for (size_t i = 0; i < N; ++i) tmp[i] = rgb[c * N + i] * rgb[cp * N + i];
            box8_ws(tmp, H, W, v, hs);
            for (size_t i = 0; i < N; ++i) {
                float m = mean[c * N + i] * mean[cp * N + i];
                v[i] = v[i] - m;
            }
            ++varIdx;
        }
    free(tmp);
    free(hs);
}

/* src/CVF.cpp:102-149: the per-pixel 3x3 solve of GuidedFilter_cv on planar inputs (var_I: 6 planes of N, cov: 3 planes, a: 3
 * planes).  Exported so that tests can put the FMA reading next to what GCC emits for the same expressions (oracle/fma_probe.c). */
void psmo_solve_models(const float *var_I, const float *cov, size_t N, float *a)
{
    for (size_t i = 0; i < N; ++i) {
        float c0 = cov[0 * N + i];
        float c1 = cov[1 * N + i];
        float c2 = cov[2 * N + i];
        float a11 = var_I[0 * N + i] + PSMO_GIF_EPS;
        float a12 = var_I[1 * N + i];
        float a13 = var_I[2 * N + i];
        float a21 = var_I[1 * N + i];
        float a22 = var_I[3 * N + i] + PSMO_GIF_EPS;
        float a23 = var_I[4 * N + i];
        float a31 = var_I[2 * N + i];
        float a32 = var_I[4 * N + i];
        float a33 = var_I[5 * N + i] + PSMO_GIF_EPS;
        if (g_variant & PSMO_VAR_FMA_SOLVE) {
            /* What GCC's -ffp-contract=fast makes of these lines on an FMA target - read off gcc 11.4 -O2 -mfma on the same
             * expressions and re-checked against a live compile by tests/test_oracle.py (oracle/fma_probe.c):
             *   x*y - z*w        -> fma(x, y, -RN(z*w))                 (the SECOND product is the rounded one)
             *   p0 + p1 + p2     -> fma(x2, y2, fma(x0, y0, RN(x1*y1))) (the MIDDLE product is the rounded one)
             *   p0 - p1 + p2     -> fma(x2, y2, fma(x0, y0, -RN(x1*y1)))
             * and common subexpressions are shared first: a31*a23 and a32*a13 are one rounded product P, so the two minors
             * (a31*a23 - a33*a21) and (a32*a13 - a33*a12) are both fma(-a33, a21, P) - the exact negative of DET's middle minor. */
#define M2(x, y, z, w) fmaf((x), (y), -((z) * (w)))
            float m00 = M2(a33, a22, a32, a23), m01 = M2(a33, a12, a32, a13), m02 = M2(a23, a12, a22, a13);
            float DETf = fmaf(a31, m02, fmaf(a11, m00, -(a21 * m01)));
            DETf = 1 / DETf;
            float n01 = fmaf(-a33, a21, a31 * a23);
            float m11 = M2(a33, a11, a31, a13), m12 = M2(a21, a13, a23, a11), m22 = M2(a22, a11, a21, a12);
            a[0 * N + i] = DETf * fmaf(c2, m02, fmaf(c0, m00, c1 * n01));
            a[1 * N + i] = DETf * fmaf(c2, m12, fmaf(c0, n01, c1 * m11));
            a[2 * N + i] = DETf * fmaf(c2, m22, fmaf(c0, m02, c1 * m12));
#undef M2
            continue;
        }
        float DET = a11 * (a33 * a22 - a32 * a23) - a21 * (a33 * a12 - a32 * a13) +
                    a31 * (a23 * a12 - a22 * a13);
        DET = 1 / DET;
        a[0 * N + i] = DET * (c0 * (a33 * a22 - a32 * a23) + c1 * (a31 * a23 - a33 * a21) +
                              c2 * (a32 * a21 - a31 * a22));
        a[1 * N + i] = DET * (c0 * (a32 * a13 - a33 * a12) + c1 * (a33 * a11 - a31 * a13) +
                              c2 * (a31 * a12 - a32 * a11));
        a[2 * N + i] = DET * (c0 * (a23 * a12 - a22 * a13) + c1 * (a21 * a13 - a23 * a11) +
                              c2 * (a22 * a11 - a21 * a12));
    }
}

/* workspace-taking core of GuidedFilter_cv; ws: 9*N floats, hs: N doubles */
static void guided_filter_ws(const float *rgb, const float *mean_I, const float *var_I, int H,
                             int W, float *p, float *ab_out, float *ws, double *hs)
{
    const size_t N = (size_t)H * W;
    float *mean_p = ws;             /* N */
    float *tmp = ws + N;            /* N */
    float *mean_Ip = ws + 2 * N;    /* 3N, becomes cov_Ip */
    float *a = ws + 5 * N;          /* 3N */
    float *q = ws + 8 * N;          /* N */

    /* src/CVF.cpp:81-82 */
    box8_slice(p, H, W, mean_p, hs);
    /* src/CVF.cpp:86-89 */
    for (int c = 0; c < 3; ++c) {
        for (size_t i = 0; i < N; ++i) tmp[i] = rgb[c * N + i] * p[i];
        box8_slice(tmp, H, W, mean_Ip + c * N, hs);
    }
    /* src/CVF.cpp:91-95: cov_Ip = mean_Ip - mean_I*mean_p */
    for (int c = 0; c < 3; ++c)
        for (size_t i = 0; i < N; ++i) {
            float t = mean_I[c * N + i] * mean_p[i];
            mean_Ip[c * N + i] = mean_Ip[c * N + i] - t;
        }
    /* src/CVF.cpp:102-149 */
    psmo_solve_models(var_I, mean_Ip, N, a);
    /* src/CVF.cpp:152-155: mean_p -= a[c]*mean_I[c], sequentially */
    for (int c = 0; c < 3; ++c)
        for (size_t i = 0; i < N; ++i) {
            float t = a[c * N + i] * mean_I[c * N + i];
            mean_p[i] = mean_p[i] - t;
        }
    if (ab_out) {
        memcpy(ab_out, a, 3 * N * sizeof(float));
        memcpy(ab_out + 3 * N, mean_p, N * sizeof(float));
    }
    /* src/CVF.cpp:157-163 */
    box8_slice(mean_p, H, W, q, hs);
    for (int c = 0; c < 3; ++c) {
        box8_slice(a + c * N, H, W, tmp, hs);
        for (size_t i = 0; i < N; ++i) {
            float t = tmp[i] * rgb[c * N + i];
            q[i] = q[i] + t;
        }
    }
    memcpy(p, q, N * sizeof(float));
}

void psmo_guided_filter(const float *rgb, const float *mean, const float *var, int H, int W,
                        float *p, float *ab)
{
    const size_t N = (size_t)H * W;
    float *ws = (float *)malloc(9 * N * sizeof(float));
    double *hs = (double *)malloc(N * sizeof(double));
    guided_filter_ws(rgb, mean, var, H, W, p, ab, ws, hs);
    free(ws);
    free(hs);
}

/* ------------------------------------------------------------------------------- */
/* CVF, Fast Guided Filter variant (src/fastguidedfilter.cpp)                       */
/* ------------------------------------------------------------------------------- */

/* cv::blur(src, dst, Size(k,k)): normalised box, anchor centre, BORDER_REFLECT_101 */
static void blur_k(const float *src, int H, int W, int k, float *dst, double *hs)
{
    const int r = k / 2;
    const double scale = 1.0 / (k * k);
    if (g_box_order == PSMO_BOX_OCV) { box_ocv(src, H, W, k, dst, hs); return; }
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            double a = 0.0;
            for (int i = -r; i <= r; ++i) a += (double)src[(size_t)y * W + r101(x + i, W)];
            hs[(size_t)y * W + x] = a;
        }
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            double a = 0.0;
            for (int j = -r; j <= r; ++j) a += hs[(size_t)r101(y + j, H) * W + x];
            dst[(size_t)y * W + x] = (float)(a * scale);
        }
}

/* cv::resize(src, dst, Size(W/s, H/s), 0, 0, INTER_NN) index maps */
static void nn_maps(int H, int W, int s, int *yofs, int *xofs)
{
    const int hs = H / s, ws = W / s;
    const double ifx = 1. / ((double)ws / W), ify = 1. / ((double)hs / H);
    for (int x = 0; x < ws; ++x) { int sx = (int)floor(x * ifx); xofs[x] = sx < W - 1 ? sx : W - 1; }
    for (int y = 0; y < hs; ++y) { int sy = (int)floor(y * ify); yofs[y] = sy < H - 1 ? sy : H - 1; }
}

/* cv::resize(..., INTER_LINEAR) coefficient tables for one axis (CV_32F, 1 channel) */
static void lin_maps(int ssize, int dsize, int *ofs, float *w1)
{
    const double scale = (double)ssize / dsize;
    for (int d = 0; d < dsize; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int sx = (int)floorf(f);
        f -= sx;
        if (sx < 0) { f = 0.f; sx = 0; }
        if (sx >= ssize - 1) { f = 0.f; sx = ssize - 1; }
        ofs[d] = sx;
        w1[d] = f;
    }
}

void psmo_fgf_setup(const float *img, int H, int W, int s, float *setup)
{
    const int hs = H / s, ws = W / s, k = 2 * (8 / s) + 1; /* FastGuidedFilter(I, GIF_R_WIN, ...): 2*(r/s)+1 */
    const size_t n = (size_t)hs * ws;
    int *yofs = (int *)malloc(hs * sizeof(int)), *xofs = (int *)malloc(ws * sizeof(int));
    float *tmp = (float *)malloc(n * sizeof(float)), *var = (float *)malloc(6 * n * sizeof(float));
    double *hsum = (double *)malloc(n * sizeof(double));
    float *I = setup, *mean = setup + 3 * n, *inv = setup + 6 * n;
    const float eps = PSMO_GIF_EPS; /* double eps added to a CV_32F Mat: the work type is float */
    nn_maps(H, W, s, yofs, xofs);
    for (int c = 0; c < 3; ++c)
        for (int y = 0; y < hs; ++y)
            for (int x = 0; x < ws; ++x) I[c * n + (size_t)y * ws + x] = img[((size_t)yofs[y] * W + xofs[x]) * 3 + c];
    for (int c = 0; c < 3; ++c) blur_k(I + c * n, hs, ws, k, mean + c * n, hsum);
    /* var_I_cc' = boxfilter(Ic.mul(Ic')) - mean_c.mul(mean_c') (+ eps on the diagonal), src/fastguidedfilter.cpp:150-155 */
    int vi = 0;
    for (int c = 0; c < 3; ++c)
        for (int cp = c; cp < 3; ++cp) {
            for (size_t i = 0; i < n; ++i) tmp[i] = I[c * n + i] * I[cp * n + i];
            blur_k(tmp, hs, ws, k, var + vi * n, hsum);
            for (size_t i = 0; i < n; ++i) {
                float m = mean[c * n + i] * mean[cp * n + i];
                float v = var[vi * n + i] - m;
                if (c == cp) v = v + eps;
                var[vi * n + i] = v;
            }
            ++vi;
        }
    /* var order: rr rg rb gg gb bb = 0..5;  src/fastguidedfilter.cpp:158-172 */
    for (size_t i = 0; i < n; ++i) {
        float rr = var[0 * n + i], rg = var[1 * n + i], rb = var[2 * n + i], gg = var[3 * n + i], gb = var[4 * n + i],
              bb = var[5 * n + i];
        float irr = gg * bb - gb * gb;
        float irg = gb * rb - rg * bb;
        float irb = rg * gb - gg * rb;
        float igg = rr * bb - rb * rb;
        float igb = rb * rg - rr * gb;
        float ibb = rr * gg - rg * rg;
        float covDet = (irr * rr + irg * rg) + irb * rb;
        inv[0 * n + i] = irr / covDet;
        inv[1 * n + i] = irg / covDet;
        inv[2 * n + i] = irb / covDet;
        inv[3 * n + i] = igg / covDet;
        inv[4 * n + i] = igb / covDet;
        inv[5 * n + i] = ibb / covDet;
    }
    free(yofs); free(xofs); free(tmp); free(var); free(hsum);
}

static void fgf_filter_ws(const float *img, const float *setup, int H, int W, int s, float *p, float *ws_f, double *hsum,
                          int *imaps, float *fmaps)
{
    const int hs = H / s, wsm = W / s, k = 2 * (8 / s) + 1;
    const size_t n = (size_t)hs * wsm;
    const float *I = setup, *mean = setup + 3 * n, *inv = setup + 6 * n;
    float *ps = ws_f, *tmp = ws_f + n, *mp = ws_f + 2 * n, *mIp = ws_f + 3 * n /* 3n */, *a = ws_f + 6 * n /* 3n */,
          *b = ws_f + 9 * n, *ma = ws_f + 10 * n /* 3n */, *mb = ws_f + 13 * n;
    int *yofs = imaps, *xofs = imaps + hs, *lxo = imaps + hs + wsm, *lyo = lxo + W;
    float *lxw = fmaps, *lyw = fmaps + W;
    nn_maps(H, W, s, yofs, xofs);
    lin_maps(wsm, W, lxo, lxw);
    lin_maps(hs, H, lyo, lyw);
    for (int y = 0; y < hs; ++y)
        for (int x = 0; x < wsm; ++x) ps[(size_t)y * wsm + x] = p[(size_t)yofs[y] * W + xofs[x]];
    blur_k(ps, hs, wsm, k, mp, hsum);
    for (int c = 0; c < 3; ++c) {
        for (size_t i = 0; i < n; ++i) tmp[i] = I[c * n + i] * ps[i];
        blur_k(tmp, hs, wsm, k, mIp + c * n, hsum);
    }
    for (size_t i = 0; i < n; ++i) {
        /* src/fastguidedfilter.cpp:184-194 */
        float cr = mIp[0 * n + i] - mean[0 * n + i] * mp[i];
        float cg = mIp[1 * n + i] - mean[1 * n + i] * mp[i];
        float cb = mIp[2 * n + i] - mean[2 * n + i] * mp[i];
        float ar = (inv[0 * n + i] * cr + inv[1 * n + i] * cg) + inv[2 * n + i] * cb;
        float ag = (inv[1 * n + i] * cr + inv[3 * n + i] * cg) + inv[4 * n + i] * cb;
        float ab = (inv[2 * n + i] * cr + inv[4 * n + i] * cg) + inv[5 * n + i] * cb;
        a[0 * n + i] = ar; a[1 * n + i] = ag; a[2 * n + i] = ab;
        b[i] = ((mp[i] - ar * mean[0 * n + i]) - ag * mean[1 * n + i]) - ab * mean[2 * n + i];
    }
    for (int c = 0; c < 3; ++c) blur_k(a + c * n, hs, wsm, k, ma + c * n, hsum);
    blur_k(b, hs, wsm, k, mb, hsum);
    /* src/fastguidedfilter.cpp:196-204: bilinear upsampling of mean_a_*, mean_b, then the linear model */
    for (int y = 0; y < H; ++y) {
        const int sy = lyo[y], sy1 = sy + 1 < hs ? sy + 1 : hs - 1;
        const float fy = lyw[y], b0 = 1.f - fy, b1 = fy;
        for (int x = 0; x < W; ++x) {
            const int sx = lxo[x], sx1 = sx + 1 < wsm ? sx + 1 : wsm - 1;
            const float fx = lxw[x], a0 = 1.f - fx, a1 = fx;
            float up[4];
            for (int c = 0; c < 4; ++c) {
                const float *src = c < 3 ? ma + c * n : mb;
                float r0 = src[(size_t)sy * wsm + sx] * a0 + src[(size_t)sy * wsm + sx1] * a1;
                float r1 = src[(size_t)sy1 * wsm + sx] * a0 + src[(size_t)sy1 * wsm + sx1] * a1;
                up[c] = r0 * b0 + r1 * b1;
            }
            const float *px = img + ((size_t)y * W + x) * 3;
            p[(size_t)y * W + x] = ((up[0] * px[0] + up[1] * px[1]) + up[2] * px[2]) + up[3];
        }
    }
}

void psmo_fgf_filter(const float *img, const float *setup, int H, int W, int s, float *p)
{
    const int hs = H / s, wsm = W / s;
    const size_t n = (size_t)hs * wsm;
    float *ws_f = (float *)malloc(14 * n * sizeof(float));
    double *hsum = (double *)malloc(n * sizeof(double));
    int *imaps = (int *)malloc((hs + wsm + W + H) * sizeof(int));
    float *fmaps = (float *)malloc((W + H) * sizeof(float));
    fgf_filter_ws(img, setup, H, W, s, p, ws_f, hsum, imaps, fmaps);
    free(ws_f); free(hsum); free(imaps); free(fmaps);
}

/* ------------------------------------------------------------------------------- */
/* DispSel                                                                          */
/* ------------------------------------------------------------------------------- */

void psmo_wta(const float *vol, int D, int H, int W, uint8_t *disp)
{
    /* src/DispSel.cpp:88-107.  "float minCost = DBL_MAX" converts to +inf. */
    const size_t N = (size_t)H * W;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            float minCost = INFINITY;
            int minDis = 0;
            for (int d = 1; d < D; ++d) {
                float c = vol[(size_t)d * N + (size_t)y * W + x];
                if (c < minCost) {
                    minCost = c;
                    minDis = d;
                }
            }
            disp[(size_t)y * W + x] = (uint8_t)minDis;
        }
}

void psmo_wta_partial(const float *vol, int d_begin, int d_end, int H, int W, float *min_cost,
                      int32_t *min_disp)
{
    const size_t N = (size_t)H * W;
    for (size_t i = 0; i < N; ++i) {
        float minCost = INFINITY;
        int minDis = 0;
        for (int d = (d_begin < 1 ? 1 : d_begin); d < d_end; ++d) {
            float c = vol[(size_t)(d - d_begin) * N + i];
            if (c < minCost) {
                minCost = c;
                minDis = d;
            }
        }
        min_cost[i] = minCost;
        min_disp[i] = minDis;
    }
}

/* ------------------------------------------------------------------------------- */
/* pthreads driver: one joinable thread per disparity, in blocks of `threads`       */
/* (src/DispEst.cpp:235-268)                                                        */
/* ------------------------------------------------------------------------------- */

typedef struct {
    /* buildCV_TD, include/CVC.h:46-53 */
    const float *lImg, *rImg, *lGrdX, *rGrdX;
    int H, W, d, right;
    float *costVol;
} buildCV_TD;

static void *buildCV_thread(void *arg)
{
    buildCV_TD *t = (buildCV_TD *)arg;
    if (t->right)
        psmo_cvc_build_right(t->lImg, t->rImg, t->lGrdX, t->rGrdX, t->H, t->W, t->d, t->costVol);
    else
        psmo_cvc_build_left(t->lImg, t->rImg, t->lGrdX, t->rGrdX, t->H, t->W, t->d, t->costVol);
    return NULL;
}

typedef struct {
    /* filterCV_TD, include/CVF.h:28 */
    const float *Img_rgb, *mean_Img, *var_Img;
    int H, W;
    float *costVol;
    float *ws;
    double *hs;
} filterCV_TD;

static void *filterCV_thread(void *arg)
{ /* src/CVF.cpp:28-41 */
    filterCV_TD *t = (filterCV_TD *)arg;
    guided_filter_ws(t->Img_rgb, t->mean_Img, t->var_Img, t->H, t->W, t->costVol, NULL, t->ws,
                     t->hs);
    return NULL;
}

typedef struct {
    void *(*fn)(void *);
    void *arg;
} job;

/* the level/block_size pattern of src/DispEst.cpp:235-251 */
static void run_blocked(job *jobs, int n, int threads)
{
    pthread_t *tid = (pthread_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(pthread_t));
    pthread_attr_t attr;
    pthread_attr_init(&attr);
    pthread_attr_setdetachstate(&attr, PTHREAD_CREATE_JOINABLE);
    for (int level = 0; level <= n / threads; ++level) {
        int block_size = (level < n / threads) ? threads : (n % threads);
        for (int iter = 0; iter < block_size; ++iter) {
            int d = level * threads + iter;
            pthread_create(&tid[d], &attr, jobs[d].fn, jobs[d].arg);
        }
        for (int iter = 0; iter < block_size; ++iter) {
            int d = level * threads + iter;
            pthread_join(tid[d], NULL);
        }
    }
    pthread_attr_destroy(&attr);
    free(tid);
}

typedef struct {
    const float *vol;
    int D, H, W, y0, y1;
    uint8_t *disp;
} wta_TD;

static void *wta_rows_thread(void *arg)
{
    wta_TD *t = (wta_TD *)arg;
    const size_t N = (size_t)t->H * t->W;
    for (int y = t->y0; y < t->y1; ++y)
        for (int x = 0; x < t->W; ++x) {
            float minCost = INFINITY;
            int minDis = 0;
            for (int d = 1; d < t->D; ++d) {
                float c = t->vol[(size_t)d * N + (size_t)y * t->W + x];
                if (c < minCost) {
                    minCost = c;
                    minDis = d;
                }
            }
            t->disp[(size_t)y * t->W + x] = (uint8_t)minDis;
        }
    return NULL;
}

/* DispSel::CVSelect is "#pragma omp parallel for" over rows (src/DispSel.cpp:88); here the
 * rows are split statically over `threads` pthreads - same arithmetic, same result. */
static void wta_parallel(const float *vol, int D, int H, int W, uint8_t *disp, int threads)
{
    pthread_t tid[PSMO_MAX_CPU_THREADS * 32];
    wta_TD td[PSMO_MAX_CPU_THREADS * 32];
    if (threads > PSMO_MAX_CPU_THREADS * 32) threads = PSMO_MAX_CPU_THREADS * 32;
    for (int t = 0; t < threads; ++t) {
        td[t] = (wta_TD){vol, D, H, W, (int)((long)H * t / threads), (int)((long)H * (t + 1) / threads), disp};
        pthread_create(&tid[t], NULL, wta_rows_thread, &td[t]);
    }
    for (int t = 0; t < threads; ++t) pthread_join(tid[t], NULL);
}

int psmo_pipeline_f32(const uint8_t *l_bgr, const uint8_t *r_bgr, int H, int W, int D, int threads,
                      uint8_t *ldisp, uint8_t *rdisp, float *lvol, float *rvol, float *raw_l,
                      float *raw_r, psmo_times *times)
{
    if (!l_bgr || !r_bgr || H < 8 || W < 8 || D < 1 || D > 256 || threads < 1) return -1;
    const size_t N = (size_t)H * W;
    int rc = -1;
    float *lImg = (float *)malloc(N * 3 * sizeof(float)), *rImg = (float *)malloc(N * 3 * sizeof(float));
    float *lG = (float *)malloc(N * sizeof(float)), *rG = (float *)malloc(N * sizeof(float));
    float *lv = lvol ? lvol : (float *)malloc(N * D * sizeof(float));
    float *rv = rvol ? rvol : (float *)malloc(N * D * sizeof(float));
    float *guide = (float *)malloc(12 * N * sizeof(float));
    float *ws = (float *)malloc((size_t)threads * 9 * N * sizeof(float));
    double *hs = (double *)malloc((size_t)threads * N * sizeof(double));
    job *jobs = (job *)malloc((size_t)D * sizeof(job));
    buildCV_TD *btd = (buildCV_TD *)malloc((size_t)D * sizeof(buildCV_TD));
    filterCV_TD *ftd = (filterCV_TD *)malloc((size_t)D * sizeof(filterCV_TD));
    if (!lImg || !rImg || !lG || !rG || !lv || !rv || !guide || !ws || !hs || !jobs || !btd || !ftd)
        goto done;

    /* src/StereoMatch.cpp:193-198 */
    psmo_u8_to_f32(l_bgr, N * 3, lImg);
    psmo_u8_to_f32(r_bgr, N * 3, rImg);

    /* ---- CostConst_CPU, src/DispEst.cpp:222-270 ---- */
    double t0 = now_ms();
    psmo_cvc_preprocess(lImg, H, W, lG);
    psmo_cvc_preprocess(rImg, H, W, rG);
    for (int d = 0; d < D; ++d) {
        btd[d] = (buildCV_TD){lImg, rImg, lG, rG, H, W, d, 0, lv + (size_t)d * N};
        jobs[d] = (job){buildCV_thread, &btd[d]};
    }
    run_blocked(jobs, D, threads);
    for (int d = 0; d < D; ++d) {
        btd[d] = (buildCV_TD){rImg, lImg, rG, lG, H, W, d, 1, rv + (size_t)d * N};
        jobs[d] = (job){buildCV_thread, &btd[d]};
    }
    run_blocked(jobs, D, threads);
    double t1 = now_ms();
    if (raw_l) memcpy(raw_l, lv, N * D * sizeof(float));
    if (raw_r) memcpy(raw_r, rv, N * D * sizeof(float));

    /* ---- cost filter: CVF::preprocess once per side + one filterCV_thread per d in the
     * same level/block pattern (driver absent from the snapshot; order
     * L-preprocess, L-filter, R-preprocess, R-filter as src/DispEst.cpp:302-305) ---- */
    double t2 = now_ms();
    for (int side = 0; side < 2; ++side) {
        float *vol = side ? rv : lv;
        psmo_cvf_preprocess(side ? rImg : lImg, H, W, guide, guide + 3 * N, guide + 6 * N);
        for (int d = 0; d < D; ++d) {
            int slot = d % threads; /* threads of one block run concurrently: distinct slots */
            ftd[d] = (filterCV_TD){guide, guide + 3 * N, guide + 6 * N, H, W, vol + (size_t)d * N,
                                   ws + (size_t)slot * 9 * N, hs + (size_t)slot * N};
            jobs[d] = (job){filterCV_thread, &ftd[d]};
        }
        run_blocked(jobs, D, threads);
    }
    double t3 = now_ms();

    /* ---- DispSelect_CPU, src/DispEst.cpp:311-321 ---- */
    wta_parallel(lv, D, H, W, ldisp, threads);
    wta_parallel(rv, D, H, W, rdisp, threads);
    double t4 = now_ms();

    if (times) {
        times->cvc_ms = t1 - t0;
        times->cvf_ms = t3 - t2;
        times->dispsel_ms = t4 - t3;
    }
    rc = 0;
done:
    free(lImg); free(rImg); free(lG); free(rG);
    if (!lvol) free(lv);
    if (!rvol) free(rv);
    free(guide); free(ws); free(hs); free(jobs); free(btd); free(ftd);
    return rc;
}

/* ---- the same pipeline, maps only, without holding the volumes ----
 * psmo_pipeline_f32 keeps both [D][H][W] float volumes (17 GB at 3840 x 2160 x 256).  The checker of the largest
 * configuration only needs the two maps: here every block of `threads` disparities is built, filtered (the same
 * buildCV_thread / filterCV_thread jobs on the same arithmetic) and folded into the running DispSel::CVSelect minimum in
 * ascending d with strict '<' (src/DispSel.cpp:96-104: identical to the loop over a stored volume), then its slices are
 * reused.  Memory: threads x (1 + 9 + 2) planes.  Test infrastructure like the rest of this file. */
typedef struct {
    const float *slices;   /* [nd][H][W], disparities d0 .. d0+nd-1 */
    int d0, nd, W, y0, y1;
    size_t N;
    float *minCost;
    uint8_t *disp;
} fold_TD;

static void *fold_rows_thread(void *arg)
{
    fold_TD *t = (fold_TD *)arg;
    for (int y = t->y0; y < t->y1; ++y)
        for (int x = 0; x < t->W; ++x) {
            const size_t i = (size_t)y * t->W + x;
            float m = t->minCost[i];
            int k = t->disp[i];
            for (int j = 0; j < t->nd; ++j) {
                const int d = t->d0 + j;
                if (d == 0) continue;                    /* the loop of src/DispSel.cpp:96 starts at d = 1 */
                const float c = t->slices[(size_t)j * t->N + i];
                if (c < m) { m = c; k = d; }
            }
            t->minCost[i] = m;
            t->disp[i] = (uint8_t)k;
        }
    return NULL;
}

int psmo_pipeline_f32_maps(const uint8_t *l_bgr, const uint8_t *r_bgr, int H, int W, int D, int threads,
                           uint8_t *ldisp, uint8_t *rdisp)
{
    if (!l_bgr || !r_bgr || !ldisp || !rdisp || H < 8 || W < 8 || D < 1 || D > 256 || threads < 1) return -1;
    if (threads > PSMO_MAX_CPU_THREADS * 32) threads = PSMO_MAX_CPU_THREADS * 32;
    const size_t N = (size_t)H * W;
    int rc = -1;
    float *lImg = (float *)malloc(N * 3 * sizeof(float)), *rImg = (float *)malloc(N * 3 * sizeof(float));
    float *lG = (float *)malloc(N * sizeof(float)), *rG = (float *)malloc(N * sizeof(float));
    float *blk = (float *)malloc((size_t)threads * N * sizeof(float));
    float *minCost = (float *)malloc(N * sizeof(float));
    float *guide = (float *)malloc(12 * N * sizeof(float));
    float *ws = (float *)malloc((size_t)threads * 9 * N * sizeof(float));
    double *hs = (double *)malloc((size_t)threads * N * sizeof(double));
    job *jobs = (job *)malloc((size_t)threads * sizeof(job));
    buildCV_TD *btd = (buildCV_TD *)malloc((size_t)threads * sizeof(buildCV_TD));
    filterCV_TD *ftd = (filterCV_TD *)malloc((size_t)threads * sizeof(filterCV_TD));
    fold_TD *wtd = (fold_TD *)malloc((size_t)threads * sizeof(fold_TD));
    pthread_t *tid = (pthread_t *)malloc((size_t)threads * sizeof(pthread_t));
    if (!lImg || !rImg || !lG || !rG || !blk || !minCost || !guide || !ws || !hs || !jobs || !btd || !ftd || !wtd || !tid) goto done;
    psmo_u8_to_f32(l_bgr, N * 3, lImg);
    psmo_u8_to_f32(r_bgr, N * 3, rImg);
    psmo_cvc_preprocess(lImg, H, W, lG);
    psmo_cvc_preprocess(rImg, H, W, rG);
    for (int side = 0; side < 2; ++side) {
        uint8_t *disp = side ? rdisp : ldisp;
        psmo_cvf_preprocess(side ? rImg : lImg, H, W, guide, guide + 3 * N, guide + 6 * N);
        for (size_t i = 0; i < N; ++i) { minCost[i] = INFINITY; disp[i] = 0; }
        for (int d0 = 0; d0 < D; d0 += threads) {
            const int nd = D - d0 < threads ? D - d0 : threads;
            for (int j = 0; j < nd; ++j) {
                btd[j] = side ? (buildCV_TD){rImg, lImg, rG, lG, H, W, d0 + j, 1, blk + (size_t)j * N}
                              : (buildCV_TD){lImg, rImg, lG, rG, H, W, d0 + j, 0, blk + (size_t)j * N};
                jobs[j] = (job){buildCV_thread, &btd[j]};
            }
            run_blocked(jobs, nd, threads);
            for (int j = 0; j < nd; ++j) {
                ftd[j] = (filterCV_TD){guide, guide + 3 * N, guide + 6 * N, H, W, blk + (size_t)j * N, ws + (size_t)j * 9 * N, hs + (size_t)j * N};
                jobs[j] = (job){filterCV_thread, &ftd[j]};
            }
            run_blocked(jobs, nd, threads);
            for (int t = 0; t < threads; ++t) {
                wtd[t] = (fold_TD){blk, d0, nd, W, (int)((long)H * t / threads), (int)((long)H * (t + 1) / threads), N, minCost, disp};
                pthread_create(&tid[t], NULL, fold_rows_thread, &wtd[t]);
            }
            for (int t = 0; t < threads; ++t) pthread_join(tid[t], NULL);
        }
    }
    rc = 0;
done:
    free(lImg); free(rImg); free(lG); free(rG); free(blk); free(minCost); free(guide); free(ws); free(hs);
    free(jobs); free(btd); free(ftd); free(wtd); free(tid);
    return rc;
}

typedef struct {
    const float *img, *setup;
    int H, W, s;
    float *costVol;
} fgf_TD;

static void *fgf_thread(void *arg)
{
    fgf_TD *t = (fgf_TD *)arg;
    psmo_fgf_filter(t->img, t->setup, t->H, t->W, t->s, t->costVol);
    return NULL;
}

/* CostConst() -> CostFilter_FGF() -> DispSelect_CPU(): the snapshot's live CPU branch (src/StereoMatch.cpp:207-224) */
int psmo_pipeline_fgf(const uint8_t *l_bgr, const uint8_t *r_bgr, int H, int W, int D, int threads, int s,
                      uint8_t *ldisp, uint8_t *rdisp, float *lvol, float *rvol, psmo_times *times)
{
    if (!l_bgr || !r_bgr || H < 8 || W < 8 || D < 1 || D > 256 || threads < 1 || (s != 2 && s != 4 && s != 8)) return -1;
    if (H / s < 2 * (8 / s) + 1 || W / s < 2 * (8 / s) + 1) return -1;
    const size_t N = (size_t)H * W, n = (size_t)(H / s) * (W / s);
    int rc = -1;
    float *lImg = (float *)malloc(N * 3 * sizeof(float)), *rImg = (float *)malloc(N * 3 * sizeof(float));
    float *lG = (float *)malloc(N * sizeof(float)), *rG = (float *)malloc(N * sizeof(float));
    float *lv = lvol ? lvol : (float *)malloc(N * D * sizeof(float));
    float *rv = rvol ? rvol : (float *)malloc(N * D * sizeof(float));
    float *setup = (float *)malloc(12 * n * sizeof(float));
    job *jobs = (job *)malloc((size_t)D * sizeof(job));
    buildCV_TD *btd = (buildCV_TD *)malloc((size_t)D * sizeof(buildCV_TD));
    fgf_TD *ftd = (fgf_TD *)malloc((size_t)D * sizeof(fgf_TD));
    if (!lImg || !rImg || !lG || !rG || !lv || !rv || !setup || !jobs || !btd || !ftd) goto done;
    psmo_u8_to_f32(l_bgr, N * 3, lImg);
    psmo_u8_to_f32(r_bgr, N * 3, rImg);
    double t0 = now_ms();
    psmo_cvc_preprocess(lImg, H, W, lG);
    psmo_cvc_preprocess(rImg, H, W, rG);
    for (int d = 0; d < D; ++d) {
        btd[d] = (buildCV_TD){lImg, rImg, lG, rG, H, W, d, 0, lv + (size_t)d * N};
        jobs[d] = (job){buildCV_thread, &btd[d]};
    }
    run_blocked(jobs, D, threads);
    for (int d = 0; d < D; ++d) {
        btd[d] = (buildCV_TD){rImg, lImg, rG, lG, H, W, d, 1, rv + (size_t)d * N};
        jobs[d] = (job){buildCV_thread, &btd[d]};
    }
    run_blocked(jobs, D, threads);
    double t1 = now_ms();
    for (int side = 0; side < 2; ++side) {
        const float *img = side ? rImg : lImg;
        psmo_fgf_setup(img, H, W, s, setup);
        for (int d = 0; d < D; ++d) {
            ftd[d] = (fgf_TD){img, setup, H, W, s, (side ? rv : lv) + (size_t)d * N};
            jobs[d] = (job){fgf_thread, &ftd[d]};
        }
        run_blocked(jobs, D, threads);
    }
    double t2 = now_ms();
    wta_parallel(lv, D, H, W, ldisp, threads);
    wta_parallel(rv, D, H, W, rdisp, threads);
    double t3 = now_ms();
    if (times) { times->cvc_ms = t1 - t0; times->cvf_ms = t2 - t1; times->dispsel_ms = t3 - t2; }
    rc = 0;
done:
    free(lImg); free(rImg); free(lG); free(rG);
    if (!lvol) free(lv);
    if (!rvol) free(rv);
    free(setup); free(jobs); free(btd); free(ftd);
    return rc;
}

/* ------------------------------------------------------------------------------- */
/* 8-bit char mode (build-defined, see header)                                      */
/* ------------------------------------------------------------------------------- */

void psmo_gray_grad_u8(const uint8_t *img, int H, int W, uint8_t *gray, uint8_t *grdx)
{
    /* OpenCV 8-bit RGB2GRAY fixed point (14 fractional bits): R2Y=4899 G2Y=9617 B2Y=1868,
     * applied in memory order (so 4899 multiplies c0, as in the float path). */
    for (size_t i = 0; i < (size_t)H * W; ++i) {
        const uint8_t *c = img + i * 3;
        gray[i] = (uint8_t)((c[0] * 4899 + c[1] * 9617 + c[2] * 1868 + (1 << 13)) >> 14);
    }
    /* Sobel(gray, grd, CV_8U, 1, 0, 1) (commented host code src/CVC_cl.cpp:127-128):
     * saturate_cast<uchar>(g[x+1]-g[x-1]), REFLECT_101. */
    for (int y = 0; y < H; ++y) {
        const uint8_t *g = gray + (size_t)y * W;
        for (int x = 0; x < W; ++x) {
            int v = (int)g[r101(x + 1, W)] - (int)g[r101(x - 1, W)];
            grdx[(size_t)y * W + x] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
}

/* assets/cvc.cl:279-301: integer colour term /3; TAU_1_US/TAU_2_US (1835/524) can never
 * clip an 8-bit value; (uchar)(ALPHA*clr + (1-ALPHA)*grd) in fp32, truncating cast. */
static inline uint8_t cost_u8(int clr3, int grd)
{
    unsigned short clrDiff = (unsigned short)(clr3 / 3);
    unsigned short grdDiff = (unsigned short)grd;
    float f = 0.9f * (float)clrDiff + (1 - 0.9f) * (float)grdDiff;
    return (uint8_t)f;
}

void psmo_cvc_build_left_u8(const uint8_t *lImg, const uint8_t *rImg, const uint8_t *lGrdX,
                            const uint8_t *rGrdX, int H, int W, int d, uint8_t *cost)
{
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const uint8_t *lC = lImg + ((size_t)y * W + x) * 3;
            int lG = lGrdX[(size_t)y * W + x];
            int clr, grd;
            if (x >= d) {
                const uint8_t *rC = rImg + ((size_t)y * W + x - d) * 3;
                clr = abs(lC[0] - rC[0]) + abs(lC[1] - rC[1]) + abs(lC[2] - rC[2]);
                grd = abs(lG - rGrdX[(size_t)y * W + x - d]);
            } else {
                clr = abs(lC[0] - 255) + abs(lC[1] - 255) + abs(lC[2] - 255);
                grd = abs(lG - 255);
            }
            cost[(size_t)y * W + x] = cost_u8(clr, grd);
        }
}

void psmo_cvc_build_right_u8(const uint8_t *lImg, const uint8_t *rImg, const uint8_t *lGrdX,
                             const uint8_t *rGrdX, int H, int W, int d, uint8_t *cost)
{
    /* predicate corrected to x < W-d (assets/cvc.cl:400-408 uses x>=d with +d reads and runs
     * off the row - SURVEY.md Appendix B); argument order as buildCV_right. */
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const uint8_t *lC = lImg + ((size_t)y * W + x) * 3;
            int lG = lGrdX[(size_t)y * W + x];
            int clr, grd;
            if (x < W - d) {
                const uint8_t *rC = rImg + ((size_t)y * W + x + d) * 3;
                clr = abs(lC[0] - rC[0]) + abs(lC[1] - rC[1]) + abs(lC[2] - rC[2]);
                grd = abs(lG - rGrdX[(size_t)y * W + x + d]);
            } else {
                clr = abs(lC[0] - 255) + abs(lC[1] - 255) + abs(lC[2] - 255);
                grd = abs(lG - 255);
            }
            cost[(size_t)y * W + x] = cost_u8(clr, grd);
        }
}

void psmo_wta_u8(const uint8_t *vol, int D, int H, int W, uint8_t *disp)
{
    /* assets/dispsel.cl:41-62 with minCost initialised to 256 instead of UCHAR_MAX */
    const size_t N = (size_t)H * W;
    for (size_t i = 0; i < N; ++i) {
        int minCost = 256, minDis = 0;
        for (int d = 1; d < D; ++d) {
            int c = vol[(size_t)d * N + i];
            if (c < minCost) {
                minCost = c;
                minDis = d;
            }
        }
        disp[i] = (uint8_t)minDis;
    }
}

static inline uint8_t quant_u8(float q)
{
    float r = rintf(q * 255.0f); /* round-half-even; NaN -> 0 */
    if (!(r > 0.0f)) return 0;
    if (r > 255.0f) return 255;
    return (uint8_t)r;
}

typedef struct {
    const float *Img_rgb, *mean_Img, *var_Img;
    int H, W;
    uint8_t *costVol;
    float *ws;
    double *hs;
} filterCV8_TD;

static void *filterCV8_thread(void *arg)
{
    filterCV8_TD *t = (filterCV8_TD *)arg;
    const size_t N = (size_t)t->H * t->W;
    float *p = t->ws + 9 * N;
    psmo_u8_to_f32(t->costVol, N, p);
    guided_filter_ws(t->Img_rgb, t->mean_Img, t->var_Img, t->H, t->W, p, NULL, t->ws, t->hs);
    for (size_t i = 0; i < N; ++i) t->costVol[i] = quant_u8(p[i]);
    return NULL;
}

typedef struct {
    const uint8_t *lImg, *rImg, *lGrdX, *rGrdX;
    int H, W, d, right;
    uint8_t *costVol;
} buildCV8_TD;

static void *buildCV8_thread(void *arg)
{
    buildCV8_TD *t = (buildCV8_TD *)arg;
    if (t->right)
        psmo_cvc_build_right_u8(t->lImg, t->rImg, t->lGrdX, t->rGrdX, t->H, t->W, t->d, t->costVol);
    else
        psmo_cvc_build_left_u8(t->lImg, t->rImg, t->lGrdX, t->rGrdX, t->H, t->W, t->d, t->costVol);
    return NULL;
}

int psmo_pipeline_u8(const uint8_t *l_bgr, const uint8_t *r_bgr, int H, int W, int D, int threads,
                     uint8_t *ldisp, uint8_t *rdisp, uint8_t *lvol, uint8_t *rvol, uint8_t *raw_l,
                     uint8_t *raw_r, psmo_times *times)
{
    if (!l_bgr || !r_bgr || H < 8 || W < 8 || D < 1 || D > 256 || threads < 1) return -1;
    const size_t N = (size_t)H * W;
    int rc = -1;
    uint8_t *gray = (uint8_t *)malloc(N), *lG = (uint8_t *)malloc(N), *rG = (uint8_t *)malloc(N);
    uint8_t *lv = lvol ? lvol : (uint8_t *)malloc(N * D), *rv = rvol ? rvol : (uint8_t *)malloc(N * D);
    float *img = (float *)malloc(N * 3 * sizeof(float));
    float *guide = (float *)malloc(12 * N * sizeof(float));
    float *ws = (float *)malloc((size_t)threads * 10 * N * sizeof(float));
    double *hs = (double *)malloc((size_t)threads * N * sizeof(double));
    job *jobs = (job *)malloc((size_t)D * sizeof(job));
    buildCV8_TD *btd = (buildCV8_TD *)malloc((size_t)D * sizeof(buildCV8_TD));
    filterCV8_TD *ftd = (filterCV8_TD *)malloc((size_t)D * sizeof(filterCV8_TD));
    if (!gray || !lG || !rG || !lv || !rv || !img || !guide || !ws || !hs || !jobs || !btd || !ftd)
        goto done;

    double t0 = now_ms();
    psmo_gray_grad_u8(l_bgr, H, W, gray, lG);
    psmo_gray_grad_u8(r_bgr, H, W, gray, rG);
    for (int d = 0; d < D; ++d) {
        btd[d] = (buildCV8_TD){l_bgr, r_bgr, lG, rG, H, W, d, 0, lv + (size_t)d * N};
        jobs[d] = (job){buildCV8_thread, &btd[d]};
    }
    run_blocked(jobs, D, threads);
    for (int d = 0; d < D; ++d) {
        btd[d] = (buildCV8_TD){r_bgr, l_bgr, rG, lG, H, W, d, 1, rv + (size_t)d * N};
        jobs[d] = (job){buildCV8_thread, &btd[d]};
    }
    run_blocked(jobs, D, threads);
    double t1 = now_ms();
    if (raw_l) memcpy(raw_l, lv, N * D);
    if (raw_r) memcpy(raw_r, rv, N * D);

    for (int side = 0; side < 2; ++side) {
        uint8_t *vol = side ? rv : lv;
        psmo_u8_to_f32(side ? r_bgr : l_bgr, N * 3, img);
        psmo_cvf_preprocess(img, H, W, guide, guide + 3 * N, guide + 6 * N);
        for (int d = 0; d < D; ++d) {
            int slot = d % threads;
            ftd[d] = (filterCV8_TD){guide, guide + 3 * N, guide + 6 * N, H, W, vol + (size_t)d * N,
                                    ws + (size_t)slot * 10 * N, hs + (size_t)slot * N};
            jobs[d] = (job){filterCV8_thread, &ftd[d]};
        }
        run_blocked(jobs, D, threads);
    }
    double t2 = now_ms();
    psmo_wta_u8(lv, D, H, W, ldisp);
    psmo_wta_u8(rv, D, H, W, rdisp);
    double t3 = now_ms();
    if (times) {
        times->cvc_ms = t1 - t0;
        times->cvf_ms = t2 - t1;
        times->dispsel_ms = t3 - t2;
    }
    rc = 0;
done:
    free(gray); free(lG); free(rG);
    if (!lvol) free(lv);
    if (!rvol) free(rv);
    free(img); free(guide); free(ws); free(hs); free(jobs); free(btd); free(ftd);
    return rc;
}

/* ------------------------------------------------------------------------------- */
/* left-right check + invalid fill ("next" row)                                     */
/* ------------------------------------------------------------------------------- */

void psmo_lr_check(const uint8_t *ldis, const uint8_t *rdis, int H, int W, uint8_t *lvalid,
                   uint8_t *rvalid)
{
    /* src/PP.cpp:17-50 */
    memset(lvalid, 0, (size_t)H * W);
    memset(rvalid, 0, (size_t)H * W);
    for (int y = 0; y < H; ++y) {
        const uint8_t *l = ldis + (size_t)y * W, *r = rdis + (size_t)y * W;
        uint8_t *lv = lvalid + (size_t)y * W, *rv = rvalid + (size_t)y * W;
        for (int x = 0; x < W; ++x) {
            int lDep = l[x];
            int rLoc = (x - lDep + W) % W;
            int rDep = r[rLoc];
            if (lDep == rDep && lDep >= 2) lv[x] = 1;
            rDep = r[x];
            int lLoc = (x + rDep + W) % W;
            lDep = l[lLoc];
            if (rDep == lDep && rDep >= 2) rv[x] = 1;
        }
    }
}

void psmo_fill_inv(uint8_t *dis, const uint8_t *valid, int H, int W)
{
    /* src/PP.cpp:58-98 (left) / 100-141 (right): identical logic per map.  The scan reads
     * dis[] values of *valid* pixels only, which the fill never modifies, so the in-place
     * update is order independent. */
    for (int y = 0; y < H; ++y) {
        uint8_t *d = dis + (size_t)y * W;
        const uint8_t *v = valid + (size_t)y * W;
        for (int x = 0; x < W; ++x) {
            if (v[x] != 0) continue;
            int lFirst = x, lFind = 0;
            while (lFirst >= 0) {
                if (v[lFirst]) { lFind = 1; break; }
                lFirst--;
            }
            int rFirst = x, rFind = 0;
            while (rFirst < W) {
                if (v[rFirst]) { rFind = 1; break; }
                rFirst++;
            }
            if (lFind && rFind)
                d[x] = (d[lFirst] <= d[rFirst]) ? d[lFirst] : d[rFirst];
            else if (lFind)
                d[x] = d[lFirst];
            else if (rFind)
                d[x] = d[rFirst];
        }
    }
}

/* ------------------------------------------------------------------------------- */
/* weighted-median post-filter ("next" row; src/PP.cpp:145-247 wgtMedian)           */
/* ------------------------------------------------------------------------------- */

#define MED_SZ 19   /* include/PP.h:12 */
#define SIG_CLR 0.1 /* include/PP.h:13 (double) */
#define SIG_DIS 9   /* include/PP.h:14 (int) */

/* one bilateral weight exactly as the loop below forms it (src/PP.cpp:169-175 left / 216-224 right): lets a test compare the
 * device's weights with the host's operand by operand (roots and exp are the two places a toolchain can differ) */
float psmo_wm_weight(const float *p3, const float *q3, int wx, int wy, int right)
{
    float disWgt = (float)(wx * wx + wy * wy);
    if (right) disWgt = sqrtf(disWgt);
    float d0 = p3[0] - q3[0], d1 = p3[1] - q3[1], d2 = p3[2] - q3[2];
    float clrWgt = d0 * d0 + d1 * d1 + d2 * d2;
    if (right) clrWgt = sqrtf(clrWgt);
    return (float)exp((double)(-disWgt / (SIG_DIS * SIG_DIS)) - (double)clrWgt / (SIG_CLR * SIG_CLR));
}

void psmo_wgt_median(const float *img, uint8_t *dis, const uint8_t *valid, int H, int W, int maxDis, int right)
{
    /* src/PP.cpp:155-196 (left map) / 199-245 (right map: the two distances go through sqrt, :218,:223).
     * The map is updated IN PLACE in raster order, so a filtered pixel sees the already filtered pixels
     * above / to the left of it (and, through the modulo wrap, the still unfiltered ones at the far end). */
    const int wndR = MED_SZ / 2;
    float *disHist = (float *)malloc((size_t)(maxDis > 0 ? maxDis : 1) * sizeof(float));
    for (int y = 0; y < H; ++y) {
        uint8_t *disData = dis + (size_t)y * W;
        const float *p = img + (size_t)y * W * 3;
        const uint8_t *validData = valid + (size_t)y * W;
        for (int x = 0; x < W; ++x) {
            if (validData[x] != 0) continue; /* just filter invalid pixels */
            memset(disHist, 0, sizeof(float) * (size_t)maxDis);
            float sumWgt = 0.0f;
            for (int wy = -wndR; wy <= wndR; ++wy) {
                int qy = (y + wy + H) % H;
                const float *q = img + (size_t)qy * W * 3;
                const uint8_t *qDisData = dis + (size_t)qy * W;
                for (int wx = -wndR; wx <= wndR; ++wx) {
                    int qx = (x + wx + W) % W;
                    int qDep = qDisData[qx];
                    if (qDep != 0) {
                        float disWgt = (float)(wx * wx + wy * wy);
                        if (right) disWgt = sqrtf(disWgt); /* sqrt(float) of <cmath>: float */
                        float d0 = p[3 * x] - q[3 * qx], d1 = p[3 * x + 1] - q[3 * qx + 1], d2 = p[3 * x + 2] - q[3 * qx + 2];
                        float clrWgt = d0 * d0 + d1 * d1 + d2 * d2;
                        if (right) clrWgt = sqrtf(clrWgt);
                        /* -disWgt / (SIG_DIS*SIG_DIS): float / int -> float; clrWgt / (SIG_CLR*SIG_CLR): float / double
                         * -> double; exp(double) of the host libm; result narrowed to float */
                        float biWgt = (float)exp((double)(-disWgt / (SIG_DIS * SIG_DIS)) - (double)clrWgt / (SIG_CLR * SIG_CLR));
                        disHist[qDep] += biWgt;
                        sumWgt += biWgt;
                    }
                }
            }
            float halfWgt = sumWgt / 2.0f;
            sumWgt = 0.0f;
            int filterDep = 0;
            for (int d = 0; d < maxDis; ++d) {
                sumWgt += disHist[d];
                if (sumWgt >= halfWgt) {
                    filterDep = d;
                    break;
                }
            }
            disData[x] = (uint8_t)filterDep; /* set new disparity */
        }
    }
    free(disHist);
}

/* ------------------------------------------------------------------------------- */
/* harness evaluation (src/StereoMatch.cpp:248-249,275-311)                         */
/* ------------------------------------------------------------------------------- */

unsigned psmo_eval_bad_pixels(const uint8_t *disp, const uint8_t *gt, const uint8_t *mask, int H,
                              int W, int maxDis, int scale_factor, int error_threshold,
                              float *avg_err)
{
    const int unit = 127 / maxDis; /* CHAR_MAX/maxDis, integer division */
    const int thresh = error_threshold * unit;
    unsigned bad = 0;
    double sum = 0.0;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            size_t i = (size_t)y * W + x;
            int s = disp[i] * scale_factor; /* convertTo(CV_8U, scale_factor) */
            if (s > 255) s = 255;
            int e = abs(s - (int)gt[i]);       /* absdiff */
            if (x <= maxDis) e = 0;            /* Rect(0,0,maxDis+1,rows) -> 0 */
            if (e <= thresh) e = 0;            /* THRESH_TOZERO */
            if (mask) e = mask[i] ? (mask[i] == 255 ? e : (int)lrint(e * (mask[i] / 255.0))) : 0;
            if (e) ++bad;
            sum += e;
        }
    if (avg_err) *avg_err = unit ? (float)(sum / ((double)H * W) / unit) : 0.0f;
    return bad;
}
