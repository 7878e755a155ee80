// psm_kernels.h - launcher interface between the C-ABI layer (psm_api_*.cpp) and the gfx950
// kernels (psm_kernels.hip, psm_pc.hip, psm_fgf.hip, psm_pp.hip).  Internal to libprimesm_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace psm {

// Per-pixel guidance of one side, packed for 16-byte lane loads (DESIGN.md "HBM layout").
//   g1[y][x] = { I0, I1, I2, GrdX }            I_c = image channel c (c0=B), GrdX = x-gradient of gray
//   g2[y][x] = { mI0, mI1, mI2, 1/DET }        mI_c = box(I_c)
//   g3[y][x] = { A00, A01, A02, A11 }          A_rc = the distinct adjugate entries of Sigma+eps*I
//   g4[y][x] = { A12, A22 }
struct Guidance {
    float4 *g1;
    float4 *g2;
    float4 *g3;
    float2 *g4;
};

struct March {       // geometry of the marching kernels
    int seg_rows;    // output rows per y-segment (0 = auto)
    int waves;       // waves (= disparity slices) per workgroup: 1,2,4,8
    int flags;       // PSM_OPT_FLAGS (include/primesm_hip.h)
    int dstep = 1;            // psm_create_shard_strided: local slice i is the global disparity d_begin + i * dstep (select forms only)
    int inflight = 1;         // PSM_OPT_FRAMES_IN_FLIGHT: pairs other contexts filter at the same time on their own streams (a planning hint only)
    int ybeg = 0, yend = 0;   // row stripe of the select-form filter (psm_set_rows): output rows [ybeg, yend) of the whole
                              // image; yend <= ybeg: all rows
    int y0(int H) const { (void)H; return yend > ybeg ? ybeg : 0; }
    int y1(int H) const { return yend > ybeg ? yend : H; }
    int rows(int H) const { return y1(H) - y0(H); }
};

// ---- batched launches: B stereo pairs of one geometry share every launch of the default path (psm_compute_batch; the
// reference's use on Middlebury-size data is a loop over pairs, src/main.cpp:64-73) ----
struct PcPair {                  // one pair of the batch: an entry of a device table the kernels index with the pair number
    const void *raw[2];          // staged interleaved images (left, right)
    Guidance g[2];
    uint8_t *p4[2];              // 8-bit char mode: {c0,c1,c2,grad} byte planes (else null)
    void *scratch;               // chunk planes of the select kernel (both sides)
    long long *keys;             // [2][H][W] packed minima
    uint8_t *maps;               // [2][H][W]
};
void launch_prep_batch(hipStream_t s, const PcPair *tab, int npairs, size_t pitch, int depth_f32, int W, int H, bool u8_planes);
void launch_guidance_batch(hipStream_t s, const PcPair *tab, int npairs, int W, int H, size_t pitch = 0, int src = 0);   // src 1 / 2: + image preparation (8-bit / float staged images)
void launch_merge_batch(hipStream_t s, const PcPair *tab, int npairs, int W, int H);   // keys -> maps of every pair

// device <-> page-locked host copy as a kernel (both pointers 16-byte aligned)
void launch_copy_bytes(hipStream_t s, void *dst, const void *src, size_t bytes);
// biased-exponent range of n floats: out[0] = max, out[1] = min over the non-zero values (initialise out to {0, 255})
void launch_range_f32(hipStream_t s, const float *p, size_t n, unsigned *out);
// image -> g1 (planarise, scale, gray, x-gradient).  src: device copy of the interleaved image.
void launch_prep(hipStream_t s, const void *src, size_t pitch, int depth_f32, int W, int H, float4 *g1, const void *src1 = nullptr,
                 float4 *g11 = nullptr);
// g1 -> g2,g3,g4 (second != NULL: both images in one launch; [ybeg, yend): rows of g2..g4 to produce, default all)
void launch_guidance(hipStream_t s, Guidance g, int W, int H, const Guidance *second = nullptr, int ybeg = 0, int yend = 0,
                     bool fma = false,    // fma: PSM_FLAG_FMA_SOLVE - minors and DET in their fused forms
                     const void *raw0 = nullptr, const void *raw1 = nullptr, size_t pitch = 0, int raw_f32 = 0);
// raw0 / raw1 (device copies of the interleaved images, row pitch `pitch`): image preparation (launch_prep) in the same launch -
// the g1 rows [ybeg, yend) of both images are written from them
// cost volume slices [d_begin, d_begin+Dloc) of one side.  base: g1 of the side's own image.
void launch_cvc(hipStream_t s, const float4 *g1_base, const float4 *g1_other, float *vol, int W, int H,
                int d_begin, int Dloc, int right, int ybeg, int yend);
// Stage A of the guided filter alone (the debug entry psm_filter_stage_a; tests compare (a0,a1,a2,b) with the oracle's
// intermediates).  variant 0 = marching, 1 = direct per-voxel.  [ybeg, yend): output rows of this launch (the arithmetic
// always refers to the full H-row planes).  Stage B alone exists in the direct formulation only (cross-check).
void launch_cvf_a(hipStream_t s, int variant, March m, const float *vol, float4 *ab, Guidance g, int W,
                  int H, int Dloc, int ybeg, int yend);
void launch_cvf_b_direct(hipStream_t s, const float4 *ab, float *vol, Guidance g, int W, int H, int Dloc);
// fused stage A+B: vin -> vout (distinct buffers), output rows [ybeg, yend) within [4, H-3)
// cvc_mode 0: read the cost slices from vin; 1/2: build the left/right costs on the fly from the g1 planes
void launch_cvf_fused(hipStream_t s, March m, const float *vin, float *vout, Guidance g, int W, int H, int Dloc,
                      int ybeg, int yend, const float4 *g1_other, int d_begin, int cvc_mode, unsigned long long *ts = nullptr);
// The same kernel with the winner-takes-all fused in ("select" forms): the filtered volume is never written.
//   plane form: per pixel the running minimum over chunks of DC slices in scratch planes -> launch_chunk_min* -> packed WTA keys
//   key form:   64-bit atomicMin on a key plane that already holds good bounds (second phase of the two-phase selection)
struct PcDev { int nxcd = 0, cus_per_xcd = 0; };
PcDev pc_dev();                                                  // of the device the calling thread is bound to
enum { PC_STORE = 0, PC_PLANES = 1, PC_KEYS = 2, PC_BOTH = 4 };  // forms of pc_plan (PC_BOTH: both volumes per launch)
struct PcPlan {
    int ngroups, nsegs, seg_rows, DC, nchunks, nbmax, nxcd;
    int cols;                                                    // output columns per column group (107; 50 in the narrow layout; 96 storing form)
    bool narrow;                                                 // two-wave workgroups (one producer + one consumer wave)
    size_t rec_per_chunk;                                        // records per chunk plane
    int rec_bytes;                                               // bytes per record (costs + disparities)
    size_t scratch_bytes() const { return rec_per_chunk * (size_t)nchunks * rec_bytes; }
};
PcPlan pc_plan(int W, int rows, int Dloc, int seg_rows_opt, int form, int batch = 1, int inflight = 1);   // batch: pairs per launch (psm_compute_batch); inflight: March::inflight
int pc_seed_stride(int W, int rows, bool u8);                     // S of the two-phase selection: every S-th slice seeds the key plane (rows: of the stripe being filtered)
// ts (may be NULL): slot of this launch in a buffer of 3 x PC_TS_SLOTS 64-bit words {first workgroup start | last workgroup
// end | form} in ticks of the device's constant-rate clock (PSM_OPT_PROFILE 2, psm_filter_launch_times)
constexpr int PC_TS_SLOTS = 4096;
// one volume per launch (costs read from vin, cvc_mode 0, or built on the fly, 1 / 2; p4_*: 8-bit char mode)
void launch_cvf_select(hipStream_t s, March m, const float *vin, Guidance g, int W, int H, int Dloc, const float4 *g1_other,
                       int d_begin, int cvc_mode, void *scratch, unsigned long long *ts = nullptr, const uint8_t *p4_own = nullptr,
                       const uint8_t *p4_other = nullptr);
void launch_chunk_min(hipStream_t s, March m, int W, int H, int Dloc, void *scratch, long long *keys, uint8_t *map);
// both volumes per launch (costs on the fly): g[0] / g[1] = guidance of the left / right image; keys / map: [2][H][W];
// Dloc slices, which ones: (sel, step) - 0: all, 1: every step-th, 2: the others
void launch_cvf_select2(hipStream_t s, March m, const Guidance *g, int W, int H, int Dloc, int d_begin, void *scratch,
                        unsigned long long *ts = nullptr, const uint8_t *const *p4 = nullptr, int sel = 0, int step = 1);
void launch_chunk_min2sides(hipStream_t s, March m, int W, int H, int Dloc, void *scratch, long long *keys, uint8_t *map);
void launch_cvf_select_keys2(hipStream_t s, March m, const Guidance *g, int W, int H, int Dloc, int d_begin, long long *keys,
                             unsigned long long *ts = nullptr, const uint8_t *const *p4 = nullptr, int init = 1, int sel = 0, int step = 1);
// the three launches above for `npairs` pairs at once (blockIdx.z = pair, pointers from the device table `tab`; every pair's
// scratch holds 2 x pc_plan(..., PC_PLANES | PC_BOTH, npairs).scratch_bytes()); to_maps: the reduction also writes the maps
void launch_cvf_select2_batch(hipStream_t s, March m, const PcPair *tab, int npairs, int W, int H, int Dloc, int d_begin,
                              unsigned long long *ts, bool u8, int sel = 0, int step = 1);
void launch_chunk_min2sides_batch(hipStream_t s, March m, const PcPair *tab, int npairs, int W, int H, int Dloc, bool to_maps);
void launch_cvf_select_keys2_batch(hipStream_t s, March m, const PcPair *tab, int npairs, int W, int H, int Dloc, int d_begin,
                                   unsigned long long *ts, bool u8, int sel, int step);
// plain 8x8 box filter of every slice (the north-star kernel in isolation)
void launch_box8(hipStream_t s, int variant, March m, const float *vol, float *out, int W, int H, int Dloc);
// WTA over local slices -> packed keys (keys != NULL) and/or final map (map != NULL)
void launch_wta(hipStream_t s, const float *vol, int W, int H, int d_begin, int Dloc, long long *keys,
                uint8_t *map);
// min over nranks key planes -> map
void launch_merge(hipStream_t s, const long long *keys_all, size_t rank_stride, int nranks, int n,
                  uint8_t *map);
// left-right check on two maps
void launch_lr_check(hipStream_t s, const uint8_t *l, const uint8_t *r, int W, int H, uint8_t *lv,
                     uint8_t *rv);

// fill invalid pixels of one map in place (valid: 0/1 per pixel)
void launch_fill_inv(hipStream_t s, uint8_t *dis, const uint8_t *valid, size_t side, int W, int H);   // both maps: the right one `side` bytes behind

// weighted-median post-filter of one map, in place, with the reference's raster-order semantics (psm_pp.hip).
// nxt: scratch of H*(W+1) ints, prog: H ints, err: 1 int (set to 1 if the dataflow watchdog fired)
void launch_wgt_median(hipStream_t s, uint8_t *dis, const uint8_t *valid, const float4 *g1, int W, int H, int maxDis, int right,
                       int *nxt, int *prog, int *err);
// parallel form (sweeps to the fixed point of the in-place recursion; psm_pp.hip)
constexpr int WM_LANE_MIN = 8192;       // active pixels from which a sweep evaluates one pixel per LANE (k_wm_eval) instead of per wave
constexpr int WM_WROW = 20;             // floats per window row of the weight cache (19 weights + 1 pad: five float4)
constexpr int WM_WPIX = 19 * WM_WROW;   // floats per cached pixel
// One map's share of the sweep state.  Sweep k (0-based) evaluates the list act = k ? list[(k + 1) & 1] : inv, whose length is
// cnt[2k]; it leaves the number of changed pixels in cnt[2k + 1], the changed pixels in chg, and the next sweep's list in
// list[k & 1] / cnt[2k + 2]; cnt[0] is the number of invalid pixels (inv).  Both maps go through every launch side by side
// (blockIdx.y): below the first sweeps a kernel is a few hundred waves, and the two maps are independent.
struct WmSide {
    uint8_t *cur;                 // the map, filtered in place
    const uint8_t *valid;
    const float4 *g1;
    uint8_t *orig, *newv, *chgb, *rowany;      // input copy, new values of changed pixels, byte maps of the gather form
    int *stamp, *list[2], *chg, *slot_of, *inv, *cnt;
    const float *wts;             // weight cache of this map (null: none, or an empty map)
};
struct WmPair { WmSide s[2]; };
void launch_wm_seed(hipStream_t s, const WmPair &p, int W, int H);
void launch_wm_sweep(hipStream_t s, const WmPair &p, int W, int H, int maxDis, int sw, bool cached, bool tail);
void launch_wm_weights(hipStream_t s, const float4 *g1, int W, int H, int right, const int *inv, const int *n_inv, int n, float *wts, int *slot_of);

// ---- Fast Guided Filter variant (psm_fgf.hip); sub = subsample rate, small planes are (H/sub) x (W/sub) ----
// g1 -> subsampled guidance ism, its means msm and the inverse covariance planes v1 = {irr,irg,irb,igg}, v2 = {igb,ibb}
void launch_fgf_setup(hipStream_t s, const float4 *g1, int W, int H, int sub, float4 *ism, float4 *msm, float4 *v1, float2 *v2);
// First half of the filter for Dloc slices: subsampled cost (cvc_mode 0: read from vol; 1/2: left/right costs of the
// sampled pixels built from the g1 planes) -> linear models -> smoothed models mab (Dloc*(H/sub)*(W/sub) float4;
// ab: scratch of the same size).
void launch_fgf_model(hipStream_t s, const float *vol, const float4 *g1, const float4 *g1_other, int W, int H, int Dloc, int d_begin,
                      int sub, int cvc_mode, const float4 *msm, const float4 *v1, const float2 *v2, float4 *ab, float4 *mab);
// Second half: bilinear upsampling of mab + q = a.I + b, written to vol ...
void launch_fgf_apply(hipStream_t s, float *vol, const float4 *g1, int W, int H, int Dloc, int sub, const float4 *mab);
// ... or consumed on the fly by the WTA: keys[H*W] receives the packed (cost, d) minimum over the local slices and the
// filtered volume is never written (needs fgf_can_fuse_wta(W))
bool fgf_can_fuse_wta(int W);
void launch_fgf_apply_wta(hipStream_t s, const float4 *g1, int W, int H, int Dloc, int d_begin, int sub, const float4 *mab,
                          long long *keys);

// ---- 8-bit char mode ----
// ({c0,c1,c2,grad} byte planes; the same word also into g1[..].w - bit pattern - of the image's g1 plane: launch_prep first)
void launch_prep_u8(hipStream_t s, const uint8_t *src, size_t pitch, int W, int H, uint8_t *planes4, float4 *g1);
void launch_cvc_u8(hipStream_t s, const uint8_t *base4, const uint8_t *other4, uint8_t *vol, int W, int H,
                   int d_begin, int Dloc, int right);
void launch_u8_to_f32(hipStream_t s, const uint8_t *src, float *dst, size_t n);
void launch_f32_to_u8(hipStream_t s, const float *src, uint8_t *dst, size_t n);
void launch_wta_u8(hipStream_t s, const uint8_t *vol, int W, int H, int d_begin, int Dloc, long long *keys,
                   uint8_t *map);

}  // namespace psm
