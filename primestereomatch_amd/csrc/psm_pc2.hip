// psm_pc2.hip - fused CVC + guided filter, producer/consumer form with TWO image columns per lane.
//
// Same algorithm, schedule and arithmetic order as k_cvf_pc (psm_kernels.hip: producer waves build the linear
// models of GuidedFilter_cv, src/CVF.cpp:72-165, into an LDS ring; consumer waves box-filter the models and write
// q), but every lane owns two adjacent columns:
//   * a wave spans 128 columns instead of 64, so the 7-column halo of each 8-tap window costs 6 % of the lanes
//     instead of 19-25 %: a workgroup of 2 producer + 2 consumer waves yields 224 output columns (7 full 128-byte
//     lines per row) from 232 model columns;
//   * the first level of the horizontal tree (p[x]+p[x+1]) is lane-local for the even windows and needs one float
//     from the neighbour lane for the odd ones; the remaining exchanges are one-lane and two-lane rotations;
//   * the per-voxel fp32 arithmetic (I*p products, 3x3 model solve, q = a.I + b) runs on (even,odd) column pairs
//     as packed fp32 (v_pk_mul_f32 / v_pk_add_f32: same IEEE roundings, half the issue slots).  To make the pairs
//     loadable as 8-byte words the guidance is read from planar copies of the g1..g4 planes (k_soa below).
// The kernel is VALU-issue bound like k_cvf_pc (there: 4.0 VALU instructions per output voxel; here 2.6).
// Needs W even (pairs) and W % 4 == 0 (16-byte output stores); other widths use k_cvf_pc.
#include "psm_kernels.h"
#include "psm_cost.h"
#include "psm_dev.h"

#pragma clang fp contract(off)

namespace psm {

typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
typedef float f4v __attribute__((ext_vector_type(4)));

// ---- planar copies of the guidance: soa[c][y][x], c = 0..3 g1 {I0,I1,I2,GrdX}, 4..7 g2 {mI0,mI1,mI2,1/DET},
//      8..11 g3 {A00,A01,A02,A11}, 12..13 g4 {A12,A22}
__global__ __launch_bounds__(256) void k_soa(const float4 *__restrict__ g1, const float4 *__restrict__ g2,
                                            const float4 *__restrict__ g3, const float2 *__restrict__ g4, size_t n,
                                            float *__restrict__ soa, int only_g1)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 a = g1[i];
    soa[i] = a.x; soa[n + i] = a.y; soa[2 * n + i] = a.z; soa[3 * n + i] = a.w;
    if (only_g1) return;
    const float4 b = g2[i], c = g3[i];
    const float2 d = g4[i];
    soa[4 * n + i] = b.x; soa[5 * n + i] = b.y; soa[6 * n + i] = b.z; soa[7 * n + i] = b.w;
    soa[8 * n + i] = c.x; soa[9 * n + i] = c.y; soa[10 * n + i] = c.z; soa[11 * n + i] = c.w;
    soa[12 * n + i] = d.x; soa[13 * n + i] = d.y;
}

void launch_soa(hipStream_t s, Guidance g, int W, int H, float *soa, int only_g1)
{
    const size_t n = (size_t)W * H;
    hipLaunchKernelGGL(k_soa, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float4 *)g.g1, (const float4 *)g.g2,
                       (const float4 *)g.g3, (const float2 *)g.g4, n, soa, only_g1);
}

namespace {

constexpr int P2_NA = 2, P2_NB = 2;        // producer / consumer waves
constexpr int P2_OUT_A = 116;              // model columns per producer wave (58 lanes x 2)
constexpr int P2_OUT_B = 112;              // output columns per consumer wave (56 lanes x 2)
constexpr int P2_COLS = P2_NB * P2_OUT_B;  // 224 output columns per workgroup = 7 lines of 128 bytes
constexpr int P2_MCOLS = P2_NA * P2_OUT_A; // 232 model columns per workgroup (>= P2_COLS + 7)
constexpr int P2_RING = 4;                 // batches of four model rows kept in LDS
static_assert(P2_MCOLS >= P2_COLS + 7 && P2_OUT_A <= 121 && P2_OUT_B <= 121, "bad layout");

#ifndef PSM_PC2_X2
#define PSM_PC2_X2 0    // two-lane exchange of the doubles: 0 ds_bpermute, 1 two DPP rotations
#endif

__device__ __forceinline__ double rol1d(double v)
{
    return __hiloint2double(rol1(__double2hiint(v)), rol1(__double2loint(v)));
}
__device__ __forceinline__ double rol2d(double v, int i2)
{
#if PSM_PC2_X2
    (void)i2;
    return rol1d(rol1d(v));
#else
    return lane_get(v, i2);
#endif
}

// Horizontal 8-tap window sums for the two columns of a lane: lane l owns columns 2l (E) and 2l+1 (O) of the wave's
// 128-column span; hE / hO are the sums over columns 2l..2l+7 / 2l+1..2l+8, each as the balanced tree
// ((t0+t1)+(t2+t3))+((t4+t5)+(t6+t7)) of oracle/psm_oracle.h (box8).
__device__ __forceinline__ void hsum8x2(float vE, float vO, int i2, double &hE, double &hO)
{
    const double dO = (double)vO;
    const double a = __dadd_rn((double)vE, dO);                  // t[2l] + t[2l+1]
    const double u = __dadd_rn(dO, (double)rol1(vE));            // t[2l+1] + t[2l+2]
    const double sE = __dadd_rn(a, rol1d(a));                    // columns 2l .. 2l+3
    const double sO = __dadd_rn(u, rol1d(u));                    // columns 2l+1 .. 2l+4
    hE = __dadd_rn(sE, rol2d(sE, i2));
    hO = __dadd_rn(sO, rol2d(sO, i2));
}

__device__ __forceinline__ f2v mk2(float a, float b) { f2v r = {a, b}; return r; }

// Raw buffer loads: descriptor and row/plane offset live in scalar registers, the lane contributes one 32-bit byte
// offset that is constant over the whole march - no per-load 64-bit vector address arithmetic.
typedef unsigned u2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float bload1(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ f2v bload2(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    const u2v v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return mk2(__uint_as_float(v.x), __uint_as_float(v.y));
}

// GuidedFilter_cv model solve (src/CVF.cpp:91-155) for a column pair; same operation order as solve_ab (psm_dev.h)
__device__ __forceinline__ void solve_ab2(f2v mp, f2v mIp0, f2v mIp1, f2v mIp2, const f2v *g2, const f2v *g3, const f2v *g4,
                                          f2v &a0, f2v &a1, f2v &a2, f2v &b)
{
    const f2v mI0 = g2[0], mI1 = g2[1], mI2 = g2[2], inv = g2[3];
    const f2v A00 = g3[0], A01 = g3[1], A02 = g3[2], A11 = g3[3], A12 = g4[0], A22 = g4[1];
    const f2v c0 = mIp0 - mI0 * mp;
    const f2v c1 = mIp1 - mI1 * mp;
    const f2v c2 = mIp2 - mI2 * mp;
    a0 = inv * ((c0 * A00 + c1 * A01) + c2 * A02);
    a1 = inv * ((c0 * A01 + c1 * A11) + c2 * A12);
    a2 = inv * ((c0 * A02 + c1 * A12) + c2 * A22);
    b = ((mp - a0 * mI0) - a1 * mI1) - a2 * mI2;
}

struct VTree2 { VTree e, o; };

template <int CVC>
__global__ __launch_bounds__(64 * (P2_NA + P2_NB)) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_cvf_pc2(const float *__restrict__ vin, float *__restrict__ vout,
                                                                 const float *__restrict__ S, const float *__restrict__ So,
                                                                 int W, int H, int Dloc, int ngroups, int nsegs, int seg_rows,
                                                                 int ybeg, int yend, int d_begin, int dsplit)
{
    __shared__ __attribute__((aligned(16))) float ring[P2_RING][4][4][P2_MCOLS];   // [batch][row][a0,a1,a2,b][column]
    __shared__ __attribute__((aligned(16))) float qbuf[2][4][P2_COLS];             // output rows, two batches
    // workgroup -> (column group, segment, slice); blocks go round-robin over the 8 XCDs (block b -> XCD b % 8).
    // dsplit: every XCD owns Dloc/8 slices of every (group, segment) pair (balanced for any pair count); otherwise
    // every XCD owns a contiguous range of pairs and walks all slices (as k_cvf_pc).  Either way the workgroups
    // resident on one XCD read the same few guidance rows.  Speed only.
    const int id = blockIdx.x, xcd = id & 7, jj = id >> 3;
    const int npairs = ngroups * nsegs;
    int d, pair;
    if (dsplit) {
        const int dch = (Dloc + 7) >> 3;
        d = xcd * dch + jj % dch;
        pair = jj / dch;
        if (d >= Dloc || pair >= npairs) return;
    } else {
        const int ppx = (npairs + 7) >> 3;
        d = jj % Dloc;
        const int pl = jj / Dloc;
        pair = xcd * ppx + pl;
        if (pl >= ppx || pair >= npairs) return;
    }
    const int g = pair % ngroups, seg = pair / ngroups;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool is_a = wave < P2_NA;
    const int xg = g * P2_COLS;                       // first output column of the workgroup
    const int xm0 = xg - 4;                           // first model column of the workgroup
    const int y0 = ybeg + seg * seg_rows, y1 = min(yend, y0 + seg_rows);   // output rows [y0, y1)
    const int mstart = max(0, y0 - 4);                // model rows produced: mstart .. mend
    const int mend = min(H - 1, y1 + 2);
    const int nbA = (mend - mstart + 1 + 3) >> 2;     // producer batches
    const int nf = (y1 - y0) + 7;                     // consumer feeds (model rows y0-4 .. y1+2, reflected)
    const int nbB = (nf + 3) >> 2;                    // consumer batches
    const int iters = nbB + 3;                        // barriers executed by every wave
    const int i2 = ((lane + 2) & 63) << 2;
    const size_t HW = (size_t)H * W;

    if (is_a) {
        // ---------------- producer: stage A ----------------
        // step s reads input row mstart-5+s; from step 8 on it yields model row mstart+(s-8)
        const int xa0 = xm0 + wave * P2_OUT_A;        // first model column of this wave (even)
        const int ciE = xa0 - 4 + 2 * lane;           // input columns of this lane before reflection
        const int cE = r101c(ciE, W), cO = r101c(ciE + 1, W);
        const bool rev = cO < cE;                     // a reflected pair is adjacent but reversed
        int cb = rev ? cO : cE;                       // first column of the pair in memory
        cb = cb > W - 2 ? W - 2 : cb;
        const bool any_rev = __builtin_amdgcn_ballot_w64(rev) != 0;
        int xab = xa0 + 2 * lane;                     // model column pair (clamped: models outside the image are never read)
        xab = xab < 0 ? 0 : (xab > W - 2 ? W - 2 : xab);
        const bool mvalid = lane < P2_OUT_A / 2;
        const float *vd = vin + (size_t)d * HW;
        const int dg = d_begin + d;                   // global disparity of this slice
        // buildCV_left: partner x-d while x >= d; buildCV_right: partner x+d while x < W-d (src/CVC.cpp:135-146,165-176)
        const bool inbE = CVC == 2 ? (cE < W - dg) : (cE >= dg), inbO = CVC == 2 ? (cO < W - dg) : (cO >= dg);
        const int cpE = CVC == 2 ? min(cE + dg, W - 1) : max(cE - dg, 0), cpO = CVC == 2 ? min(cO + dg, W - 1) : max(cO - dg, 0);
        const bool any_border = CVC != 0 && __builtin_amdgcn_ballot_w64(!(inbE && inbO)) != 0;
        VTree2 t0 = {}, t1 = {}, t2 = {}, t3 = {};
        f2v pin[2], gin[2][4], o2[4], o3[4], o4[2];
        float othE[2][4], othO[2][4];
        const unsigned plane = (unsigned)HW * 4u;     // bytes per plane
        const __amdgpu_buffer_rsrc_t rS = make_rsrc(S, 14u * plane), rO = make_rsrc(So, 4u * plane);
        const __amdgpu_buffer_rsrc_t rV = make_rsrc(CVC == 0 ? (const void *)vd : (const void *)S, plane);
#ifndef PSM_PC2_ABL
#define PSM_PC2_ABL 0
#endif
#if PSM_PC2_ABL & 1   // experiment: all lanes and rows read the same words (same instruction count, no data movement; invalid results)
        const int vcb = 0, vcpE = 0, vcpO = 0, vxab = 0;
#else
        const int vcb = cb * 4, vcpE = cpE * 4, vcpO = cpO * 4, vxab = xab * 4;
#endif
        // input row of step STEP (image/cost pair): issued one step ahead
#define PSM_ISSUE_A2(SLOT, STEP)                                                        \
    {                                                                                   \
        const int row_ = (PSM_PC2_ABL & 1) ? 0 : r101c(mstart - 5 + (STEP), H) * W * 4; \
        if (CVC == 0) pin[SLOT] = bload2(rV, vcb, row_);                                \
        _Pragma("unroll") for (int c_ = 0; c_ < (CVC == 0 ? 3 : 4); ++c_) {             \
            gin[SLOT][c_] = bload2(rS, vcb, c_ * plane + row_);                         \
            if (CVC != 0) {                                                             \
                othE[SLOT][c_] = bload1(rO, vcpE, c_ * plane + row_);                   \
                othO[SLOT][c_] = bload1(rO, vcpO, c_ * plane + row_);                   \
            }                                                                           \
        }                                                                               \
    }
        // guidance of the model row step STEP completes: requested during that step, consumed by the solve at the
        // start of the next one (the solve runs one step behind the trees)
#define PSM_ISSUE_G2(STEP)                                                              \
    {                                                                                   \
        int ya_ = mstart - 8 + (STEP);                                                  \
        ya_ = ya_ < 0 ? 0 : (ya_ > H - 1 ? H - 1 : ya_);                                \
        const int oa_ = (PSM_PC2_ABL & 1) ? 0 : ya_ * W * 4;                            \
        _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_) {                              \
            o2[c_] = bload2(rS, vxab, (4 + c_) * plane + oa_);                          \
            o3[c_] = bload2(rS, vxab, (8 + c_) * plane + oa_);                          \
        }                                                                               \
        o4[0] = bload2(rS, vxab, 12 * plane + oa_);                                     \
        o4[1] = bload2(rS, vxab, 13 * plane + oa_);                                     \
    }
        f2v cm0 = {0.f, 0.f}, cm1 = cm0, cm2 = cm0, cm3 = cm0;   // window means of the previous step
        float *cdst = nullptr;                                   // where its model row goes (nullptr: warm-up)
        // solve + hand-over of the model row the PREVIOUS step completed
#define PSM_SOLVE_A2()                                                                              \
    {                                                                                               \
        f2v a0_, a1_, a2_, b_;                                                                      \
        solve_ab2(cm0, cm1, cm2, cm3, o2, o3, o4, a0_, a1_, a2_, b_);                               \
        if (cdst != nullptr && mvalid) {                                                            \
            *(f2v *)(cdst) = a0_;                                                                   \
            *(f2v *)(cdst + P2_MCOLS) = a1_;                                                        \
            *(f2v *)(cdst + 2 * P2_MCOLS) = a2_;                                                    \
            *(f2v *)(cdst + 3 * P2_MCOLS) = b_;                                                     \
        }                                                                                           \
    }
        // one step: finish the previous row, consume the loads of step S (slot K&1), issue those of step S+1
#define PSM_STEP_A2(K, STEPNO, DST)                                                                 \
    {                                                                                               \
        PSM_SOLVE_A2()                                                                              \
        PSM_ISSUE_G2(STEPNO)                                                                        \
        PSM_ISSUE_A2((K + 1) & 1, (STEPNO) + 1)                                                     \
        f2v gi_[4];                                                                                 \
        _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_) gi_[c_] = gin[K & 1][c_];                  \
        f2v p_;                                                                                     \
        if (CVC == 0) p_ = pin[K & 1];                                                              \
        if (any_rev) {   /* image border workgroups only */                                        \
            asm volatile("; reversed pairs");                                                       \
            _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_) gi_[c_] = rev ? gi_[c_].yx : gi_[c_];  \
            if (CVC == 0) p_ = rev ? p_.yx : p_;                                                    \
        }                                                                                           \
        if (CVC != 0) {                                                                             \
            const float4 aE_ = make_float4(gi_[0].x, gi_[1].x, gi_[2].x, gi_[3].x);                 \
            const float4 aO_ = make_float4(gi_[0].y, gi_[1].y, gi_[2].y, gi_[3].y);                 \
            float pe_ = cost_pair(aE_, make_float4(othE[K & 1][0], othE[K & 1][1], othE[K & 1][2], othE[K & 1][3])); \
            float po_ = cost_pair(aO_, make_float4(othO[K & 1][0], othO[K & 1][1], othO[K & 1][2], othO[K & 1][3])); \
            if (any_border) {   /* only where x < d (left) or x >= W-d (right) occurs in this wave */ \
                asm volatile("; border cost");                                                      \
                const float be_ = cost_border(aE_), bo_ = cost_border(aO_);                         \
                pe_ = inbE ? pe_ : be_;                                                             \
                po_ = inbO ? po_ : bo_;                                                             \
            }                                                                                       \
            p_ = mk2(pe_, po_);                                                                     \
        }                                                                                           \
        const f2v q1_ = gi_[0] * p_, q2_ = gi_[1] * p_, q3_ = gi_[2] * p_;                          \
        double h0e, h0o, h1e, h1o, h2e, h2o, h3e, h3o;                                              \
        hsum8x2(p_.x, p_.y, i2, h0e, h0o);                                                          \
        hsum8x2(q1_.x, q1_.y, i2, h1e, h1o);                                                        \
        hsum8x2(q2_.x, q2_.y, i2, h2e, h2o);                                                        \
        hsum8x2(q3_.x, q3_.y, i2, h3e, h3o);                                                        \
        cm0 = mk2(box_out(vstep<K>(t0.e, h0e)), box_out(vstep<K>(t0.o, h0o)));                      \
        cm1 = mk2(box_out(vstep<K>(t1.e, h1e)), box_out(vstep<K>(t1.o, h1o)));                      \
        cm2 = mk2(box_out(vstep<K>(t2.e, h2e)), box_out(vstep<K>(t2.o, h2o)));                      \
        cm3 = mk2(box_out(vstep<K>(t3.e, h3e)), box_out(vstep<K>(t3.o, h3o)));                      \
        cdst = (DST) != nullptr ? (DST) + K * (4 * P2_MCOLS) : nullptr;                             \
        __builtin_amdgcn_sched_barrier(0);                                                          \
    }
        PSM_ISSUE_G2(0) PSM_ISSUE_A2(0, 0) __builtin_amdgcn_sched_barrier(0);
        {   // warm-up: 8 rows fill the trees (no conditionals around the tree updates, see k_cvf_pc)
            float *const none = nullptr;
            PSM_STEP_A2(0, 0, none) PSM_STEP_A2(1, 1, none) PSM_STEP_A2(2, 2, none) PSM_STEP_A2(3, 3, none)
            PSM_STEP_A2(0, 4, none) PSM_STEP_A2(1, 5, none) PSM_STEP_A2(2, 6, none) PSM_STEP_A2(3, 7, none)
        }
        // The model row of a batch's last step is handed over during the first step of the next iteration (or by
        // the flush below): consumers run two batches behind, so it is in the ring long before it is read.
        for (int b = 0; b < nbA; ++b) {
            const int s0 = 8 + b * 4;
            float *dst = &ring[b & (P2_RING - 1)][0][0][wave * P2_OUT_A + 2 * lane];
            PSM_STEP_A2(0, s0, dst) PSM_STEP_A2(1, s0 + 1, dst) PSM_STEP_A2(2, s0 + 2, dst) PSM_STEP_A2(3, s0 + 3, dst)
            __syncthreads();
        }
        PSM_SOLVE_A2()
        for (int b = nbA; b < iters; ++b) __syncthreads();
#undef PSM_SOLVE_A2
#undef PSM_STEP_A2
#undef PSM_ISSUE_A2
#undef PSM_ISSUE_G2
    } else {
        // ---------------- consumer: stage B ----------------
        // feed j (j = 0 .. nf-1) is model row r101(y0-4+j); from feed 7 on the trees yield output row y0+j-7
        const int wb = wave - P2_NA;
        const int xb0 = xg + wb * P2_OUT_B;           // first output column of this wave
        const int xmod = xb0 - 4 + 2 * lane;          // model columns (window starts) of this lane
        int mcE = r101(xmod, W) - xm0, mcO = r101(xmod + 1, W) - xm0;   // REFLECT_101 of the model planes, as ring columns
        mcE = mcE < 0 ? 0 : (mcE > P2_MCOLS - 1 ? P2_MCOLS - 1 : mcE);
        mcO = mcO < 0 ? 0 : (mcO > P2_MCOLS - 1 ? P2_MCOLS - 1 : mcO);
        if (lane >= P2_OUT_B / 2 + 4) { mcE = 0; mcO = 1; }   // windows of the output lanes end at lane 59: don't-care lanes
        const bool pairs_ok = __builtin_amdgcn_ballot_w64(mcO != mcE + 1) == 0;   // interior workgroups: aligned pairs
        int xbb = xb0 + 2 * lane;                     // output column pair of this lane
        xbb = xbb > W - 2 ? W - 2 : xbb;
        const bool ovalid = lane < P2_OUT_B / 2;
        float *od = vout + (size_t)d * HW;
        const int amax = 4 * nbA - 1;
        VTree2 t0 = {}, t1 = {}, t2 = {}, t3 = {};
        f2v o1[4][3];                                 // I0,I1,I2 at (output row, output column pair), one batch ahead
        const unsigned plane = (unsigned)HW * 4u;
        const __amdgpu_buffer_rsrc_t rS = make_rsrc(S, 3u * plane);
        const int vxbb = xbb * 4;
#define PSM_ISSUE_B2(SLOT, J)                                                           \
    {                                                                                   \
        int yb_ = y0 + (J) - 7;                                                         \
        yb_ = yb_ < 0 ? 0 : (yb_ > H - 1 ? H - 1 : yb_);                                \
        const int ob_ = yb_ * W * 4;                                                    \
        o1[SLOT][0] = bload2(rS, vxbb, ob_);                                            \
        o1[SLOT][1] = bload2(rS, vxbb, plane + ob_);                                    \
        o1[SLOT][2] = bload2(rS, vxbb, 2 * plane + ob_);                                \
    }
        // ring row (four quantity planes) that feed J consumes (wave-uniform arithmetic)
        auto model_of = [&](int J) -> const float * {
            int a = r101(y0 - 4 + J, H) - mstart;
            a = a < 0 ? 0 : (a > amax ? amax : a);
            return &ring[(a >> 2) & (P2_RING - 1)][a & 3][0][0];
        };
        auto fetch = [&](const float *row, f2v *m) {
            if (pairs_ok) {
#pragma unroll
                for (int q = 0; q < 4; ++q) m[q] = *(const f2v *)(row + q * P2_MCOLS + mcE);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) m[q] = mk2(row[q * P2_MCOLS + mcE], row[q * P2_MCOLS + mcO]);
            }
        };
        // merged store of output batch `c` (rows parked in qbuf[c & 1] one iteration earlier)
        auto store_batch = [&](int c) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {          // row k of the batch is stored by consumer wave k % P2_NB
                const int j = 4 * c + k;
                const int yo = y0 + j - 7;
                if (k % P2_NB == wb && j >= 7 && yo < y1) {
                    float *row = od + (size_t)yo * W + xg;
                    const float *src = &qbuf[c & 1][k][0];
                    const int cc = lane * 4;
                    if (lane < P2_COLS / 4 && xg + cc < W)
                        __builtin_nontemporal_store(*reinterpret_cast<const f4v *>(src + cc), reinterpret_cast<f4v *>(row + cc));
                }
            }
        };
        PSM_ISSUE_B2(0, 0) PSM_ISSUE_B2(1, 1) PSM_ISSUE_B2(2, 2) PSM_ISSUE_B2(3, 3)
        __syncthreads();                               // iteration 0
        __syncthreads();                               // iteration 1
        for (int b = 2; b <= nbB + 1; ++b) {           // iteration b: consume feed batch c = b-2
            const int c = b - 2;
            if (c >= 1) store_batch(c - 1);
            {
                const int j0 = 4 * c;
                f2v mrow[2][4];                    // model rows of the current / next feed, alternating
                fetch(model_of(j0), mrow[0]);
                f2v qv[4];
#define PSM_STEP_B2(K)                                                                              \
    {                                                                                               \
        if (K < 3) fetch(model_of(j0 + K + 1), mrow[(K + 1) & 1]);   /* model row of the next feed, one step ahead */ \
        double h0e, h0o, h1e, h1o, h2e, h2o, h3e, h3o;                                              \
        hsum8x2(mrow[K & 1][0].x, mrow[K & 1][0].y, i2, h0e, h0o);                                  \
        hsum8x2(mrow[K & 1][1].x, mrow[K & 1][1].y, i2, h1e, h1o);                                  \
        hsum8x2(mrow[K & 1][2].x, mrow[K & 1][2].y, i2, h2e, h2o);                                  \
        hsum8x2(mrow[K & 1][3].x, mrow[K & 1][3].y, i2, h3e, h3o);                                  \
        const f2v ma0_ = mk2(box_out(vstep<K>(t0.e, h0e)), box_out(vstep<K>(t0.o, h0o)));           \
        const f2v ma1_ = mk2(box_out(vstep<K>(t1.e, h1e)), box_out(vstep<K>(t1.o, h1o)));           \
        const f2v ma2_ = mk2(box_out(vstep<K>(t2.e, h2e)), box_out(vstep<K>(t2.o, h2o)));           \
        const f2v mb_ = mk2(box_out(vstep<K>(t3.e, h3e)), box_out(vstep<K>(t3.o, h3o)));            \
        /* q = ((box(b) + box(a0)*I0) + box(a1)*I1) + box(a2)*I2   (src/CVF.cpp:157-163) */          \
        qv[K] = ((mb_ + ma0_ * o1[K][0]) + ma1_ * o1[K][1]) + ma2_ * o1[K][2];                      \
        PSM_ISSUE_B2(K, j0 + K + 4)                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                          \
    }
                PSM_STEP_B2(0) PSM_STEP_B2(1) PSM_STEP_B2(2) PSM_STEP_B2(3)
#undef PSM_STEP_B2
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (ovalid) *(f2v *)&qbuf[c & 1][k][wb * P2_OUT_B + 2 * lane] = qv[k];
            }
            __syncthreads();
        }
        store_batch(nbB - 1);                          // iteration nbB+2
        __syncthreads();
#undef PSM_ISSUE_B2
    }
}

}  // namespace

int pc2_cols() { return P2_COLS; }

void launch_cvf_pc2(hipStream_t s, const float *vin, float *vout, const float *soa, const float *soa_other, int W, int H, int Dloc,
                    int ybeg, int yend, int d_begin, int cvc_mode, int seg_rows)
{
    if (yend <= ybeg) return;
    const int rows = yend - ybeg;
    const int ngroups = (W + P2_COLS - 1) / P2_COLS;
    if (seg_rows <= 0) {
        // 11 halo rows per segment: long segments are cheaper (~360 rows), but a disparity shard with few slices
        // needs more segments to keep >= ~2048 workgroups in the launch
        const int per_seg = ngroups * Dloc;
        int k = (rows + 399) / 400;
        const int kmin = (2048 + per_seg - 1) / per_seg, kmax = rows / 64 > 1 ? rows / 64 : 1;
        if (k < kmin) k = kmin;
        if (k > kmax) k = kmax;
        seg_rows = (rows + k - 1) / k;
    }
    if (seg_rows > rows) seg_rows = rows;
    const int nsegs = (rows + seg_rows - 1) / seg_rows;
    const int npairs = ngroups * nsegs;
    const int dsplit = Dloc >= 16;
    const int nblocks = dsplit ? 8 * ((Dloc + 7) / 8) * npairs : 8 * ((npairs + 7) / 8) * Dloc;
    const dim3 blk(64 * (P2_NA + P2_NB));
#define PSM_LAUNCH_PC2(CV)                                                                                        \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cvf_pc2<CV>), dim3(nblocks), blk, 0, s, vin, vout, soa, soa_other, W, H, \
                       Dloc, ngroups, nsegs, seg_rows, ybeg, yend, d_begin, dsplit)
    if (cvc_mode == 1) PSM_LAUNCH_PC2(1); else if (cvc_mode == 2) PSM_LAUNCH_PC2(2); else PSM_LAUNCH_PC2(0);
#undef PSM_LAUNCH_PC2
}

}  // namespace psm
