/*
 * psm_oracle.h - CPU restatement ("oracle") of the PRiMEStereoMatch DispEst hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under primestereomatch_amd/ (the product) may
 * include, link or call this.  Allowed users: tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py.
 *
 * PARITY STATUS: "parity unpinned".  The reference (/root/reference) is C++ on cv::Mat
 * and cannot be compiled in this image (OpenCV is not installed), it has
 * no tests and no golden vectors.  This file restates the reference's CPU arithmetic
 * (src/CVC.cpp, src/CVF.cpp, src/DispSel.cpp, src/PP.cpp) with the OpenCV primitives
 * replaced by the canonical definitions of SURVEY.md Appendix A.  Every function cites
 * the reference file:line it follows.
 *
 * Conventions
 *   - images are interleaved 3-channel rows (like CV_32FC3 / CV_8UC3), channel order as
 *     delivered by cv::imread: c0=B, c1=G, c2=R.
 *   - planes are dense row-major H x W.
 *   - cost volumes are dense [D][H][W] float (the reference keeps D separate H x W Mats,
 *     src/DispEst.cpp:31-37; the slices are the same data).
 *   - compile with -ffp-contract=off and without -ffast-math: every expression below is
 *     evaluated op-for-op in the precision written.
 */
#ifndef PSM_ORACLE_H
#define PSM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSMO_GIF_EPS 0.0001f /* include/ComFunc.h:50 */
#define PSMO_MAX_CPU_THREADS 8 /* include/ComFunc.h:52 */

/* ---- input conditioning -------------------------------------------------------- */

/* src/StereoMatch.cpp:195-196: frame.convertTo(frame, CV_32F, 1/255.0f) */
void psmo_u8_to_f32(const uint8_t *src, size_t n, float *dst);

/* ---- CVC: cost volume construction (float mode) -------------------------------- */

/* src/CVC.cpp:41-46 CVC::preprocess: gray (CV_RGB2GRAY applied to BGR data) then
 * x-Sobel ksize=1 ([-1 0 1], BORDER_REFLECT_101).  img: H x W x 3 float. */
void psmo_cvc_preprocess(const float *img, int H, int W, float *grdx);

/* src/CVC.cpp:122-149 CVC::buildCV_left.  Argument names follow the reference. */
void psmo_cvc_build_left(const float *lImg, const float *rImg, const float *lGrdX,
                         const float *rGrdX, int H, int W, int d, float *cost);
/* src/CVC.cpp:151-179 CVC::buildCV_right.  NOTE the caller passes the images swapped:
 * buildCV_right(rImg, lImg, rGrdX, lGrdX, d, rcostVol[d]) (src/DispEst.cpp:217,260). */
void psmo_cvc_build_right(const float *lImg, const float *rImg, const float *lGrdX,
                          const float *rGrdX, int H, int W, int d, float *cost);

/* ---- CVF: guided image filter --------------------------------------------------- */

/* cv::boxFilter(src, dst, -1, Size(8,8)) as used at src/CVF.cpp:50,63,82,88,158,160.
 * Canonical definition (order fixed here):
 *   hs[y][x] = T8_i (double)src[y][r101(x-4+i)]            i = 0..7
 *   dst[y][x] = (float)( T8_j hs[r101(y-4+j)][x] * (1.0/64) )
 *   T8(t0..t7) = ((t0+t1)+(t2+t3)) + ((t4+t5)+(t6+t7))      all in double
 * H, W >= 8. */
void psmo_box8(const float *src, int H, int W, float *dst);

/* Summation order of every box filter below (psmo_box8, the guided filters, the pipelines):
 * PSMO_BOX_TREE = the canonical order above (default; what the HIP kernels evaluate bit for bit);
 * PSMO_BOX_OCV  = the order OpenCV's engines execute for CV_32F (RowSum<float,double>: running sum along the
 * row, s += (double)S[i+k] - (double)S[i]; ColumnSum<double,float>: running column accumulator over the whole image,
 * out = (float)((SUM + newest) * (1./(k*k))), SUM = (SUM + newest) - oldest) - i.e. what the reference binary executes
 * at src/CVF.cpp:50,63,82,88,158,160 when OpenCV's generic (non-IPP) path runs.  Both orders sum the same 64 taps
 * in double; they differ only where a double addition rounds.  Process-global: set it before starting a pipeline. */
enum { PSMO_BOX_TREE = 0, PSMO_BOX_OCV = 1 };
void psmo_set_box_order(int order);
int psmo_get_box_order(void);

/* Toolchain-dependent readings of two reference lines (see psm_oracle.c: myCostGrd2, guided_filter_ws).  0 = canon.
 * Used by tests only, to bound how far the reference binary can be from the canonical arithmetic. */
enum { PSMO_VAR_FABS_DOUBLE = 1, PSMO_VAR_FMA_SOLVE = 2,
       /* tolerance-form models of the per-slice box filters (psm_oracle.c: box8_slice) */
       PSMO_VAR_F32_L1 = 4, PSMO_VAR_F32_L2 = 8, PSMO_VAR_RUNCOL = 16 };
void psmo_set_variant(int bits);
int psmo_get_variant(void);

/* src/CVF.cpp:44-70 CVF::preprocess.  rgb: 3 planes, mean: 3 planes, var: 6 planes
 * (order 00,01,02,11,12,22), each H*W floats, stored back to back. */
void psmo_cvf_preprocess(const float *img, int H, int W, float *rgb, float *mean, float *var);

/* src/CVF.cpp:72-165 GuidedFilter_cv.  p (H*W) is replaced by q (CVF::filterCV,
 * src/CVF.cpp:22-26).  If ab != NULL it receives the four intermediate planes
 * a0,a1,a2,b (4*H*W floats, back to back) - the values the second box-filter round
 * consumes (src/CVF.cpp:102-155). */
void psmo_guided_filter(const float *rgb, const float *mean, const float *var, int H, int W,
                        float *p, float *ab);

/* src/CVF.cpp:102-149 alone: the per-pixel 3x3 solve on planar inputs (var: 6 planes of n, cov: 3 planes -> a: 3 planes);
 * honours psmo_set_variant (PSMO_VAR_FMA_SOLVE). */
void psmo_solve_models(const float *var, const float *cov, size_t n, float *a);

/* ---- CVF, Fast Guided Filter variant ("next" row: what the snapshot's live CPU branch runs) ------- */
/* src/fastguidedfilter.cpp (FastGuidedFilterColor) as used by DispEst::CostFilter_FGF
 * (src/DispEst.cpp:281-296): r = GIF_R_WIN = 8, eps = GIF_EPS, s = subsample_rate (2, 4 or 8).
 * OpenCV pieces restated canonically:
 *   cv::resize(..., INTER_NN):     dst(y,x) = src(min(floor(y*ify),H-1), min(floor(x*ifx),W-1)), ifx = 1/((W/s)/(double)W)
 *   cv::blur(I, Size(k,k)), k = 2*(r/s)+1: taps -k/2..+k/2, REFLECT_101, fp64 sums (x taps left to right,
 *                                  then y taps top to bottom), (float)(sum * (1.0/(k*k)))
 *   cv::resize(..., INTER_LINEAR): fx = (float)((dx+0.5)*scale-0.5), sx = floor(fx), fx -= sx, clamped at both
 *                                  ends; row pass S[sx]*(1.f-fx) + S[sx+1]*fx, then column pass, fp32, no FMA.
 * setup: 12 planes of (H/s)*(W/s) floats: I0,I1,I2 (subsampled), mean0..2, invrr,invrg,invrb,invgg,invgb,invbb. */
void psmo_fgf_setup(const float *img, int H, int W, int s, float *setup);
/* filters one slice in place (FastGuidedFilterColor::filterSingleChannel + the NN subsampling of p) */
void psmo_fgf_filter(const float *img, const float *setup, int H, int W, int s, float *p);

/* ---- DispSel: winner takes all --------------------------------------------------- */

/* src/DispSel.cpp:83-109 DispSel::CVSelect on a dense [D][H][W] volume. */
void psmo_wta(const float *vol, int D, int H, int W, uint8_t *disp);

/* Shard form used by the multi-GPU path: argmin over the global disparities
 * [d_begin, d_end) of which this volume holds slices (slice 0 == d_begin); d = 0 is
 * never a candidate (src/DispSel.cpp:96).  Writes per-pixel (min cost, min d); no
 * candidate -> (+inf, 0).  Merging shards in ascending d with strict '<' reproduces
 * psmo_wta exactly. */
void psmo_wta_partial(const float *vol, int d_begin, int d_end, int H, int W, float *min_cost,
                      int32_t *min_disp);

/* ---- whole path, driven like the reference pthreads path ------------------------ */

typedef struct {
    double cvc_ms, cvf_ms, dispsel_ms; /* src/StereoMatch.cpp:209-219 stage timers */
} psmo_times;

/* l_bgr/r_bgr: H x W x 3 uint8 (cv::imread order).  threads: pthreads per block
 * (src/DispEst.cpp:235-268 level/block_size pattern).  lvol/rvol (optional, may be NULL):
 * receive the filtered [D][H][W] volumes.  raw_l/raw_r (optional): receive the
 * unfiltered cost volumes.  Returns 0, or -1 on bad arguments / allocation failure. */
int psmo_pipeline_f32(const uint8_t *l_bgr, const uint8_t *r_bgr, int H, int W, int D, int threads,
                      uint8_t *ldisp, uint8_t *rdisp, float *lvol, float *rvol, float *raw_l,
                      float *raw_r, psmo_times *times);

/* The same pipeline, maps only, without holding the two [D][H][W] volumes (17 GB at 3840 x 2160 x 256): blocks of `threads`
 * disparities are built, filtered and folded into the running DispSel::CVSelect minimum (ascending d, strict '<',
 * src/DispSel.cpp:96-104), so the maps are those of psmo_pipeline_f32 bit for bit. */
int psmo_pipeline_f32_maps(const uint8_t *l_bgr, const uint8_t *r_bgr, int H, int W, int D, int threads,
                           uint8_t *ldisp, uint8_t *rdisp);

/* CostConst() -> CostFilter_FGF() -> DispSelect_CPU(): the snapshot's live CPU branch
 * (src/StereoMatch.cpp:207-224), s = subsample_rate in {2,4,8}. */
int psmo_pipeline_fgf(const uint8_t *l_bgr, const uint8_t *r_bgr, int H, int W, int D, int threads, int s,
                      uint8_t *ldisp, uint8_t *rdisp, float *lvol, float *rvol, psmo_times *times);

/* ---- 8-bit char mode (build-defined; the reference has no CPU 8-bit path) -------- */
/* Contract (DESIGN.md "8-bit mode"): u8 planar colour, u8 gray/gradient, u8 cost volume
 * per assets/cvc.cl:250-329 (cvc_uchar_nv) with the right-volume predicate corrected to
 * x < W-d; CVF = the float guided filter on cost*(1/255.0f) with the f32 guidance image,
 * re-quantised q8 = sat_u8(rintf(q*255.0f)); WTA per assets/dispsel.cl:22-63 with the
 * initial minimum set to 256 so that a cost of 255 can win. */
void psmo_gray_grad_u8(const uint8_t *img, int H, int W, uint8_t *gray, uint8_t *grdx);
void psmo_cvc_build_left_u8(const uint8_t *lImg, const uint8_t *rImg, const uint8_t *lGrdX,
                            const uint8_t *rGrdX, int H, int W, int d, uint8_t *cost);
void psmo_cvc_build_right_u8(const uint8_t *lImg, const uint8_t *rImg, const uint8_t *lGrdX,
                             const uint8_t *rGrdX, int H, int W, int d, uint8_t *cost);
void psmo_wta_u8(const uint8_t *vol, int D, int H, int W, uint8_t *disp);
int psmo_pipeline_u8(const uint8_t *l_bgr, const uint8_t *r_bgr, int H, int W, int D, int threads,
                     uint8_t *ldisp, uint8_t *rdisp, uint8_t *lvol, uint8_t *rvol, uint8_t *raw_l,
                     uint8_t *raw_r, psmo_times *times);

/* ---- "next" row: left-right check + invalid fill --------------------------------- */
/* src/PP.cpp:17-50 lrCheck */
void psmo_lr_check(const uint8_t *ldis, const uint8_t *rdis, int H, int W, uint8_t *lvalid,
                   uint8_t *rvalid);
/* src/PP.cpp:52-143 fillInv (one map at a time) */
void psmo_fill_inv(uint8_t *dis, const uint8_t *valid, int H, int W);

/* ---- "next" row: weighted-median post-filter (plain bilateral-histogram form, not JointWMF) ------ */
/* src/PP.cpp:145-247 wgtMedian, one map at a time: every pixel with valid == 0 is replaced by the weighted median of
 * the disparities in its 19 x 19 window (MED_SZ, include/PP.h:12; modulo wrap at the image border; pixels of
 * disparity 0 do not vote), weights exp(-disWgt/81 - clrWgt/0.01) (SIG_DIS 9, SIG_CLR 0.1) with
 *   right == 0 (left map,  :169-175): disWgt = wx^2+wy^2,        clrWgt = |p-q|^2
 *   right != 0 (right map, :216-224): disWgt = sqrt(wx^2+wy^2),  clrWgt = sqrt(|p-q|^2)
 * img: H x W x 3 float (the CV_32FC3 image PP::processDM receives), dis: H x W, updated IN PLACE in raster order
 * (later pixels see earlier results - the sequential semantics of the reference's single-threaded form).
 * exp is the host libm's double exp, narrowed to float, as in the reference. */
float psmo_wm_weight(const float *p3, const float *q3, int wx, int wy, int right);   /* one weight of the loop below */
void psmo_wgt_median(const float *img, uint8_t *dis, const uint8_t *valid, int H, int W, int maxDis, int right);

/* ---- evaluation recipe of the harness (src/StereoMatch.cpp:275-311) -------------- */
/* disp: raw WTA map; gt: ground-truth map (disparity*scale); mask: 0/255 or NULL.
 * Returns number of bad pixels; *avg_err receives the "Avg Err" figure. */
unsigned psmo_eval_bad_pixels(const uint8_t *disp, const uint8_t *gt, const uint8_t *mask, int H,
                              int W, int maxDis, int scale_factor, int error_threshold,
                              float *avg_err);

#ifdef __cplusplus
}
#endif
#endif
