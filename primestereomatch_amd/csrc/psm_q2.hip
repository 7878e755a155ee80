// psm_q2.hip - k_cvf_q2: the select-mode fused kernel (CVC + guided filter + WTA, psm_pc.hip MODE 1) with TWO image
// columns per lane and the filter's channels split over the waves of a workgroup.
//
// Why: k_cvf_pc is VALU-issue bound (psm_pc.hip), and per output voxel most of its instructions are the cross-lane part
// of the horizontal 8-tap trees (5 DPP moves + 2 ds_bpermute + 2 conversions per channel) plus the 7 halo lanes every
// 64-lane wave carries.  With two adjacent columns per lane the first tree level is lane-local for the even window and
// needs one float from the neighbour for the odd one, the second level is a one-lane rotation of a double, the third a
// two-lane one: per channel and voxel 2.5 DPP + 2 ds_bpermute + 1.5 conversions, and the halo costs 7 of 128 columns.
// Two columns per lane double the sliding-tree state, though (k_cvf_pc2, psm_pc2.hip: 256 VGPRs, two waves per SIMD,
// slower).  Here every wave carries only TWO of the four channels of its stage:
//   A1: cost p (myCostGrd, src/CVC.cpp:18-39, from the two g1 planes) and I0*p  -> window means mp, mIp0;  p, means -> LDS
//   A2: I1*p, I2*p (p from A1, one batch later) -> means mIp1, mIp2; + A1's means -> model solve (src/CVF.cpp:91-155)
//       -> model rows (a0,a1,a2,b) into the LDS ring
//   B1: box means of a0, a1 (ring, REFLECT_101 by index arithmetic as in k_cvf_pc) -> LDS
//   B2: box means of a2, b; one batch later + B1's means: q = ((mb + ma0*I0) + ma1*I1) + ma2*I2 (src/CVF.cpp:157-163)
//       -> running strict-'<' minimum over the slices of the chunk (src/DispSel.cpp:96-104) in the chunk plane
// so a wave holds 4 trees x 14 registers as in k_cvf_pc and three workgroups fit a CU.  One barrier per batch of four
// rows; interval t: A1 batch t, A2 batch t-1, ring batch t-3, B feed batch t-5, recombination of feed batch t-6.
// A workgroup covers 128 input columns -> 121 model columns -> 114 output columns (89 % useful lanes in every wave).
// Arithmetic, operation order and results are exactly those of k_cvf_pc / the oracle (tests compare bit for bit).
#include "psm_kernels.h"
#include "psm_cost.h"
#include "psm_dev.h"

#ifndef PSM_Q2_ABL
#define PSM_Q2_ABL 0      // ablation experiments (invalid results): 1 no model solve, 2 no guidance loads in A2
#endif
#ifndef PSM_Q2_TIMING
#define PSM_Q2_TIMING 0   // 1: every wave accumulates its cycles between barriers (work) and inside them (wait) per role
#endif

namespace psm {

#if PSM_Q2_TIMING
__device__ unsigned long long g_q2_dbg[8];   // work cycles of roles A1, A2, B1, B2, then their barrier-wait cycles
#define Q_SYNC()                                                                 \
    {                                                                            \
        const unsigned long long a_ = __builtin_readcyclecounter();              \
        __syncthreads();                                                         \
        const unsigned long long b_ = __builtin_readcyclecounter();              \
        q_work += a_ - q_mark; q_wait += b_ - a_; q_mark = b_;                   \
    }
#else
#define Q_SYNC() __syncthreads()
#endif

namespace {

constexpr int Q_COLS = 114;    // output columns per workgroup
constexpr int Q_MCOLS = 121;   // model columns per workgroup
constexpr int Q_MPAD = 122;    // ring row pitch (floats per channel)
constexpr int Q_RING = 4;      // batches of four model rows kept in LDS

typedef unsigned q_u2 __attribute__((ext_vector_type(2)));
typedef unsigned q_u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t q_rsrc(const void *p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float4 q_load4(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    const q_u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ float2 q_load2(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    const q_u2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
}

__device__ __forceinline__ double q_rol1d(double v)
{
    return __hiloint2double(rol1(__double2hiint(v)), rol1(__double2loint(v)));
}
#ifndef PSM_Q2_X2
#define PSM_Q2_X2 0    // two-lane exchange of the doubles: 0 ds_bpermute, 1 two DPP rotations
#endif
__device__ __forceinline__ double q_rol2d(double v, int i2)
{
#if PSM_Q2_X2
    (void)i2;
    return q_rol1d(q_rol1d(v));
#else
    return lane_get(v, i2);
#endif
}

// Horizontal 8-tap window sums for the two columns of a lane: lane l owns columns 2l (E) and 2l+1 (O) of the wave's
// 128-column span; hE / hO are the sums over columns 2l..2l+7 / 2l+1..2l+8, each as the balanced tree
// ((t0+t1)+(t2+t3))+((t4+t5)+(t6+t7)) of oracle/psm_oracle.h (box8).
__device__ __forceinline__ void q_hsum8x2(float vE, float vO, int i2, double &hE, double &hO)
{
    const double dO = (double)vO;
    const double a = __dadd_rn((double)vE, dO);                  // t[2l] + t[2l+1]
    const double u = __dadd_rn(dO, (double)rol1(vE));            // t[2l+1] + t[2l+2]
    const double sE = __dadd_rn(a, q_rol1d(a));                  // columns 2l .. 2l+3
    const double sO = __dadd_rn(u, q_rol1d(u));                  // columns 2l+1 .. 2l+4
    hE = __dadd_rn(sE, q_rol2d(sE, i2));
    hO = __dadd_rn(sO, q_rol2d(sO, i2));
}

struct QTree2 { VTree e, o; };

}  // namespace

// CVC = 1 (left volume) / 2 (right volume): costs built on the fly from the g1 planes (the volume stays virtual).
template <int CVC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_cvf_q2(const float4 *__restrict__ G1, const float4 *__restrict__ G2, const float4 *__restrict__ G3,
                                               const float2 *__restrict__ G4, const float4 *__restrict__ Gother, int W, int H, int Dloc,
                                               int ngroups, int nsegs, int seg_rows, int d_begin, int DC, float *__restrict__ kcost,
                                               unsigned *__restrict__ kdisp, int nbmax)
{
    __shared__ __attribute__((aligned(16))) float ring[Q_RING][4][4][Q_MPAD];   // [batch][row][a0,a1,a2,b][model column]
    __shared__ __attribute__((aligned(16))) float2 abuf[2][4][3][64];           // A1 -> A2: {pE,pO}, {mpE,mpO}, {mIp0E,mIp0O}
    __shared__ __attribute__((aligned(16))) float4 bbuf[2][4][64];              // B1 -> B2: {ma0E, ma0O, ma1E, ma1O}
    const int nchunks = (Dloc + DC - 1) / DC;
    const int id = blockIdx.x;
    const int npairs = ngroups * nsegs;
    // work items (pair, chunk), pair-major; XCD x (blocks go round-robin over the XCDs) takes the x-th eighth of them:
    // a contiguous range of pairs whose chunks run back to back (guidance stays in that XCD's L2), equal work per XCD
    // whatever the pair count
    const int nitems = npairs * nchunks;
    const int ipx = (nitems + 7) >> 3;
    const int xcd = id & 7, jj = id >> 3;
    const int item = xcd * ipx + jj;
    if (jj >= ipx || item >= nitems) return;
    const int pair = item / nchunks, ch = item % nchunks;
    const int g = pair % ngroups, seg = pair / ngroups;
#ifndef PSM_Q2_ROT
#define PSM_Q2_ROT 1     // rotate the role <-> hardware wave assignment with the workgroup index (spreads the four roles over the SIMDs)
#endif
    const int wave = (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) + (PSM_Q2_ROT ? (int)blockIdx.x >> 3 : 0)) & 3, lane = threadIdx.x & 63;
    const int xg = g * Q_COLS;                        // first output column of the workgroup
    const int xm0 = xg - 4;                           // first model column
    const int xin0 = xg - 8;                          // first input column
    const int y0 = seg * seg_rows, y1 = min(H, y0 + seg_rows);   // output rows [y0, y1)
    const int mstart = max(0, y0 - 4);                // model rows produced: mstart .. mend
    const int mend = min(H - 1, y1 + 2);
    const int nbA = (mend - mstart + 1 + 3) >> 2;     // model-row batches
    const int nf = (y1 - y0) + 7;                     // consumer feeds (model rows y0-4 .. y1+2, reflected)
    const int nbB = (nf + 3) >> 2;                    // consumer batches
    const int T = nbB + 6;                            // barriers executed by every wave per slice
    const int i2 = ((lane + 2) & 63) << 2;
    const size_t HW = (size_t)H * W;
    const unsigned plane16 = (unsigned)HW * 16u;

#if PSM_Q2_TIMING
    unsigned long long q_work = 0, q_wait = 0, q_mark = __builtin_readcyclecounter();
#endif
    for (int ds = 0; ds < DC; ++ds) {                 // the slices of this chunk, ascending d
    const int d = ch * DC + ds;
    if (d >= Dloc) break;                             // uniform over the workgroup
    const int dg = d_begin + d;                       // global disparity of this slice
    if (wave == 0) {
        // ---------------- A1: cost, channels p and I0*p ----------------
        // step s reads input row r101c(mstart-5+s); interval t = steps 4t .. 4t+3; 2 + nbA intervals
        const int cE = r101c(xin0 + 2 * lane, W), cO = r101c(xin0 + 2 * lane + 1, W);
        const bool inbE = CVC == 2 ? (cE < W - dg) : (cE >= dg), inbO = CVC == 2 ? (cO < W - dg) : (cO >= dg);
        const int cpE = CVC == 2 ? min(cE + dg, W - 1) : max(cE - dg, 0), cpO = CVC == 2 ? min(cO + dg, W - 1) : max(cO - dg, 0);
        const bool any_border = __builtin_amdgcn_ballot_w64(!(inbE && inbO)) != 0;
        const __amdgpu_buffer_rsrc_t rG1 = q_rsrc(G1, plane16), rGo = q_rsrc(Gother, plane16);
        const int vE = cE * 16, vO = cO * 16, vpE = cpE * 16, vpO = cpO * 16;
        QTree2 t0 = {}, t1 = {};
        float4 ginE[2], ginO[2], othE[2], othO[2];
#define Q_ISSUE_A1(SLOT, STEP)                                                          \
    {                                                                                   \
        const int row_ = r101c(mstart - 5 + (STEP), H) * W * 16;                        \
        ginE[SLOT] = q_load4(rG1, vE, row_);                                            \
        ginO[SLOT] = q_load4(rG1, vO, row_);                                            \
        othE[SLOT] = q_load4(rGo, vpE, row_);                                           \
        othO[SLOT] = q_load4(rGo, vpO, row_);                                           \
    }
#define Q_STEP_A1(K, S, PAR)                                                                        \
    {                                                                                               \
        Q_ISSUE_A1((K + 1) & 1, (S) + 1)                                                            \
        float pE = cost_pair(ginE[K & 1], othE[K & 1]), pO = cost_pair(ginO[K & 1], othO[K & 1]);   \
        if (any_border) {                                                                           \
            asm volatile("; border cost");                                                          \
            const float bE_ = cost_border(ginE[K & 1]), bO_ = cost_border(ginO[K & 1]);             \
            pE = inbE ? pE : bE_;                                                                   \
            pO = inbO ? pO : bO_;                                                                   \
        }                                                                                           \
        double h0E, h0O, h1E, h1O;                                                                  \
        q_hsum8x2(pE, pO, i2, h0E, h0O);                                                            \
        q_hsum8x2(__fmul_rn(ginE[K & 1].x, pE), __fmul_rn(ginO[K & 1].x, pO), i2, h1E, h1O);        \
        const float mpE = box_out(vstep<K>(t0.e, h0E)), mpO = box_out(vstep<K>(t0.o, h0O));         \
        const float m0E = box_out(vstep<K>(t1.e, h1E)), m0O = box_out(vstep<K>(t1.o, h1O));         \
        abuf[PAR][K][0][lane] = make_float2(pE, pO);                                                \
        abuf[PAR][K][1][lane] = make_float2(mpE, mpO);                                              \
        abuf[PAR][K][2][lane] = make_float2(m0E, m0O);                                              \
        __builtin_amdgcn_sched_barrier(0);                                                          \
    }
        Q_ISSUE_A1(0, 0) __builtin_amdgcn_sched_barrier(0);
        const int nA1 = 2 + nbA;
        for (int t = 0; t < nA1; ++t) {
            const int s0 = 4 * t, par = t & 1;
            Q_STEP_A1(0, s0, par) Q_STEP_A1(1, s0 + 1, par) Q_STEP_A1(2, s0 + 2, par) Q_STEP_A1(3, s0 + 3, par)
            Q_SYNC();
        }
        for (int t = nA1; t < T; ++t) Q_SYNC();
#undef Q_STEP_A1
#undef Q_ISSUE_A1
    } else if (wave == 1) {
        // ---------------- A2: channels I1*p and I2*p, model solve ----------------
        // interval t (>= 1) = A1's steps of interval t-1; from step 8 on a step yields model row mstart+(s-8)
        const int cE = r101c(xin0 + 2 * lane, W), cO = r101c(xin0 + 2 * lane + 1, W);
        int xaE = xm0 + 2 * lane, xaO = xaE + 1;      // model columns of this lane (clamped for the loads; models outside the image are never read)
        xaE = xaE < 0 ? 0 : (xaE > W - 1 ? W - 1 : xaE);
        xaO = xaO < 0 ? 0 : (xaO > W - 1 ? W - 1 : xaO);
        const bool mvE = 2 * lane < Q_MCOLS, mvO = 2 * lane + 1 < Q_MCOLS;
        const __amdgpu_buffer_rsrc_t rG1 = q_rsrc(G1, plane16), rG2 = q_rsrc(G2, plane16), rG3 = q_rsrc(G3, plane16);
        const __amdgpu_buffer_rsrc_t rG4 = q_rsrc(G4, (unsigned)HW * 8u);
        const int vE = cE * 16 + 4, vO = cO * 16 + 4;   // {I1, I2} of g1 = {I0, I1, I2, GrdX}
        const int vaE = xaE * 16, vaO = xaO * 16;
        QTree2 t2 = {}, t3 = {};
        float2 gyzE[2], gyzO[2], o4E[2] = {}, o4O[2] = {};
        float4 o2E[2] = {}, o2O[2] = {}, o3E[2] = {}, o3O[2] = {};
#define Q_ISSUE_A2(SLOT, STEP)                                                          \
    {                                                                                   \
        const int row_ = r101c(mstart - 5 + (STEP), H) * W * 16;                        \
        int ya_ = mstart - 8 + (STEP);                                                  \
        ya_ = ya_ < 0 ? 0 : (ya_ > H - 1 ? H - 1 : ya_);                                \
        const int oa_ = ya_ * W;                                                        \
        gyzE[SLOT] = q_load2(rG1, vE, row_);                                            \
        gyzO[SLOT] = q_load2(rG1, vO, row_);                                            \
        if (!(PSM_Q2_ABL & 2)) {   /* ablation bit 2: no guidance loads (invalid results) */ \
        o2E[SLOT] = q_load4(rG2, vaE, oa_ * 16);                                        \
        o2O[SLOT] = q_load4(rG2, vaO, oa_ * 16);                                        \
        o3E[SLOT] = q_load4(rG3, vaE, oa_ * 16);                                        \
        o3O[SLOT] = q_load4(rG3, vaO, oa_ * 16);                                        \
        o4E[SLOT] = q_load2(rG4, vaE >> 1, oa_ * 8);                                    \
        o4O[SLOT] = q_load2(rG4, vaO >> 1, oa_ * 8);                                    \
        }                                                                               \
    }
#define Q_STEP_A2(K, S, PAR, DST)                                                                   \
    {                                                                                               \
        Q_ISSUE_A2((K + 1) & 1, (S) + 1)                                                            \
        const float2 p_ = ap_, mp_ = amp_, m0_ = am0_;                                              \
        if (K < 3) { ap_ = abuf[PAR][(K + 1) & 3][0][lane]; amp_ = abuf[PAR][(K + 1) & 3][1][lane]; am0_ = abuf[PAR][(K + 1) & 3][2][lane]; } \
        double h2E, h2O, h3E, h3O;                                                                  \
        q_hsum8x2(__fmul_rn(gyzE[K & 1].x, p_.x), __fmul_rn(gyzO[K & 1].x, p_.y), i2, h2E, h2O);    \
        q_hsum8x2(__fmul_rn(gyzE[K & 1].y, p_.x), __fmul_rn(gyzO[K & 1].y, p_.y), i2, h3E, h3O);    \
        const float m1E = box_out(vstep<K>(t2.e, h2E)), m1O = box_out(vstep<K>(t2.o, h2O));         \
        const float m2E = box_out(vstep<K>(t3.e, h3E)), m2O = box_out(vstep<K>(t3.o, h3O));         \
        float4 rE, rO;                                                                              \
        if (PSM_Q2_ABL & 1) { rE = make_float4(mp_.x, m0_.x, m1E, m2E); rO = make_float4(mp_.y, m0_.y, m1O, m2O); } /* ablation: no solve (invalid results) */ \
        else {                                                                                      \
            rE = solve_ab(mp_.x, m0_.x, m1E, m2E, o2E[K & 1], o3E[K & 1], o4E[K & 1]);               \
            rO = solve_ab(mp_.y, m0_.y, m1O, m2O, o2O[K & 1], o3O[K & 1], o4O[K & 1]);               \
        }                                                                                           \
        if ((DST) != nullptr) {                                                                     \
            float *d_ = (DST) + K * 4 * Q_MPAD;                                                     \
            if (mvE) { d_[0] = rE.x; d_[Q_MPAD] = rE.y; d_[2 * Q_MPAD] = rE.z; d_[3 * Q_MPAD] = rE.w; } \
            if (mvO) { d_[1] = rO.x; d_[Q_MPAD + 1] = rO.y; d_[2 * Q_MPAD + 1] = rO.z; d_[3 * Q_MPAD + 1] = rO.w; } \
        }                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                          \
    }
        Q_SYNC();                               // interval 0
        Q_ISSUE_A2(0, 0) __builtin_amdgcn_sched_barrier(0);
        float2 ap_, amp_, am0_;                        // A1's hand-over of the next step (read one step ahead)
#define Q_PRE_A2(PAR) { ap_ = abuf[PAR][0][0][lane]; amp_ = abuf[PAR][0][1][lane]; am0_ = abuf[PAR][0][2][lane]; }
        {   // intervals 1, 2: the eight warm-up steps (loop bodies stay free of conditionals around the tree updates)
            float *const none = nullptr;
            Q_PRE_A2(0)
            Q_STEP_A2(0, 0, 0, none) Q_STEP_A2(1, 1, 0, none) Q_STEP_A2(2, 2, 0, none) Q_STEP_A2(3, 3, 0, none)
            Q_SYNC();
            Q_PRE_A2(1)
            Q_STEP_A2(0, 4, 1, none) Q_STEP_A2(1, 5, 1, none) Q_STEP_A2(2, 6, 1, none) Q_STEP_A2(3, 7, 1, none)
            Q_SYNC();
        }
        for (int b = 0; b < nbA; ++b) {                // interval 3 + b: ring batch b
            const int s0 = 8 + 4 * b, par = b & 1;     // A1 wrote these steps in interval 2 + b
            Q_PRE_A2(par)
            float *dst = &ring[b & (Q_RING - 1)][0][0][2 * lane];
            Q_STEP_A2(0, s0, par, dst) Q_STEP_A2(1, s0 + 1, par, dst) Q_STEP_A2(2, s0 + 2, par, dst) Q_STEP_A2(3, s0 + 3, par, dst)
            Q_SYNC();
        }
        for (int t = 3 + nbA; t < T; ++t) Q_SYNC();
#undef Q_STEP_A2
#undef Q_PRE_A2
#undef Q_ISSUE_A2
    } else {
        // ---------------- B1 / B2: second box-filter round on two of the four model channels each ----------------
        // feed j (j = 0 .. nf-1) is model row r101(y0-4+j); from feed 7 on the trees yield output row y0+j-7
        const bool is_b2 = wave == 3;
        // model columns this lane consumes (REFLECT_101 of the model planes, as ring columns): an adjacent pair, possibly
        // reversed by the reflection; lanes whose columns lie outside the ring read a harmless valid pair
        int mE = r101(xm0 + 2 * lane, W) - xm0, mO = r101(xm0 + 2 * lane + 1, W) - xm0;
        const bool swp = mO < mE;
        int mb = swp ? mO : mE;
        mb = mb < 0 ? 0 : (mb > Q_MPAD - 2 ? Q_MPAD - 2 : mb);   // (lane 60 reads columns 120, 121: the pad column only feeds unused windows)
        const int chan = is_b2 ? 2 : 0;                // B1: a0, a1   B2: a2, b
        const int amax = 4 * nbA - 1;
        QTree2 t0 = {}, t1 = {};
        auto model_of = [&](int J) -> const float * {  // ring row of the model row that feed J consumes
            int a = r101(y0 - 4 + J, H) - mstart;
            a = a < 0 ? 0 : (a > amax ? amax : a);
            return &ring[(a >> 2) & (Q_RING - 1)][a & 3][chan][mb];
        };
        // two channels of one feed: u = channel chan, v = channel chan+1, as (E, O) pairs
#define Q_READ_B(J, UE, UO, VE, VO)                                                                 \
    {                                                                                               \
        const float *r_ = model_of(J);                                                              \
        const float u0_ = r_[0], u1_ = r_[1], v0_ = r_[Q_MPAD], v1_ = r_[Q_MPAD + 1];               \
        UE = swp ? u1_ : u0_; UO = swp ? u0_ : u1_; VE = swp ? v1_ : v0_; VO = swp ? v0_ : v1_;     \
    }
        if (!is_b2) {
            // ---- B1 ----
            for (int t = 0; t < 5; ++t) Q_SYNC();
            for (int c = 0; c < nbB; ++c) {            // interval 5 + c
                const int j0 = 4 * c, par = c & 1;
#define Q_STEP_B1(K)                                                                                \
    {                                                                                               \
        float uE, uO, vE_, vO_;                                                                     \
        Q_READ_B(j0 + K, uE, uO, vE_, vO_)                                                          \
        double h0E, h0O, h1E, h1O;                                                                  \
        q_hsum8x2(uE, uO, i2, h0E, h0O);                                                            \
        q_hsum8x2(vE_, vO_, i2, h1E, h1O);                                                          \
        const float a0E = box_out(vstep<K>(t0.e, h0E)), a0O = box_out(vstep<K>(t0.o, h0O));         \
        const float a1E = box_out(vstep<K>(t1.e, h1E)), a1O = box_out(vstep<K>(t1.o, h1O));         \
        bbuf[par][K][lane] = make_float4(a0E, a0O, a1E, a1O);                                       \
    }
                Q_STEP_B1(0) Q_STEP_B1(1) Q_STEP_B1(2) Q_STEP_B1(3)
#undef Q_STEP_B1
                Q_SYNC();
            }
            Q_SYNC();                           // interval 5 + nbB
        } else {
            // ---- B2 ----
            const int xoE = xg + 2 * lane, xoO = xoE + 1;          // output columns of this lane
            const bool outE = lane < Q_COLS / 2 && xoE < W, outO = lane < Q_COLS / 2 && xoO < W;
            const __amdgpu_buffer_rsrc_t rG1 = q_rsrc(G1, plane16);
            const int voE = min(xoE, W - 1) * 16, voO = min(xoO, W - 1) * 16;
            // this wave's records of the chunk plane: [batch][lane]: costs E rows 0-3, costs O rows 0-3 | disparities E, O
            const size_t krec = (size_t)(ch * npairs + pair) * nbmax * 64;
            const __amdgpu_buffer_rsrc_t rKc = q_rsrc(reinterpret_cast<float4 *>(kcost) + 2 * krec, (unsigned)nbmax * 2048u);
            const __amdgpu_buffer_rsrc_t rKd = q_rsrc(kdisp + 2 * krec, (unsigned)nbmax * 512u);
            float sv[4][4];                            // {ma2E, ma2O, mbE, mbO} of the batch filtered in the previous interval
            float gE[4][3], gO[4][3];                  // g1.xyz at the output pixels of that batch
            float kqE[4], kqO[4];                      // running minima of that batch ...
            unsigned kdE = 0, kdO = 0;                 // ... and their disparities
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                kqE[k] = kqO[k] = __builtin_inff();
#pragma unroll
                for (int i = 0; i < 4; ++i) sv[k][i] = 0.f;
#pragma unroll
                for (int i = 0; i < 3; ++i) gE[k][i] = gO[k][i] = 0.f;
            }
            // loads for the recombination of feed batch C (issued one interval before it is needed)
#define Q_ISSUE_B2(C)                                                                               \
    {                                                                                               \
        _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                             \
            int yb_ = y0 + 4 * (C) + k - 7;                                                         \
            yb_ = yb_ < 0 ? 0 : (yb_ > H - 1 ? H - 1 : yb_);                                        \
            const float4 e_ = q_load4(rG1, voE, yb_ * W * 16), o_ = q_load4(rG1, voO, yb_ * W * 16); \
            gE[k][0] = e_.x; gE[k][1] = e_.y; gE[k][2] = e_.z;                                      \
            gO[k][0] = o_.x; gO[k][1] = o_.y; gO[k][2] = o_.z;                                      \
        }                                                                                           \
        if (ds > 0) {                                                                               \
            const q_u4 ce_ = __builtin_amdgcn_raw_buffer_load_b128(rKc, lane * 32, (C) * 2048, 16); \
            const q_u4 co_ = __builtin_amdgcn_raw_buffer_load_b128(rKc, lane * 32 + 16, (C) * 2048, 16); \
            const q_u2 dd_ = __builtin_amdgcn_raw_buffer_load_b64(rKd, lane * 8, (C) * 512, 16);    \
            kqE[0] = __uint_as_float(ce_.x); kqE[1] = __uint_as_float(ce_.y); kqE[2] = __uint_as_float(ce_.z); kqE[3] = __uint_as_float(ce_.w); \
            kqO[0] = __uint_as_float(co_.x); kqO[1] = __uint_as_float(co_.y); kqO[2] = __uint_as_float(co_.z); kqO[3] = __uint_as_float(co_.w); \
            kdE = dd_.x; kdO = dd_.y;                                                               \
        }                                                                                           \
    }
            // recombination + WTA of feed batch C (its box means are in sv, B1's in bbuf[C & 1])
#define Q_SELECT_B2(C)                                                                              \
    {                                                                                               \
        bool any_ = false;                                                                          \
        _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                             \
            const float4 m_ = bbuf[(C) & 1][k][lane];          /* ma0E, ma0O, ma1E, ma1O */        \
            const float qE_ = __fadd_rn(__fadd_rn(__fadd_rn(sv[k][2], __fmul_rn(m_.x, gE[k][0])), __fmul_rn(m_.z, gE[k][1])), __fmul_rn(sv[k][0], gE[k][2])); \
            const float qO_ = __fadd_rn(__fadd_rn(__fadd_rn(sv[k][3], __fmul_rn(m_.y, gO[k][0])), __fmul_rn(m_.w, gO[k][1])), __fmul_rn(sv[k][1], gO[k][2])); \
            const int j_ = 4 * (C) + k, yo_ = y0 + j_ - 7;                                          \
            const bool row_ = j_ >= 7 && yo_ < y1 && dg != 0;                                       \
            const bool bE_ = row_ && outE && qE_ < kqE[k], bO_ = row_ && outO && qO_ < kqO[k];      \
            kqE[k] = bE_ ? qE_ : kqE[k];                                                            \
            kqO[k] = bO_ ? qO_ : kqO[k];                                                            \
            kdE = bE_ ? ((kdE & ~(0xffu << (8 * k))) | ((unsigned)dg << (8 * k))) : kdE;            \
            kdO = bO_ ? ((kdO & ~(0xffu << (8 * k))) | ((unsigned)dg << (8 * k))) : kdO;            \
            any_ |= bE_ | bO_;                                                                      \
        }                                                                                           \
        if (ds == 0 || __builtin_amdgcn_ballot_w64(any_) != 0) {                                    \
            const q_u4 ce_ = {__float_as_uint(kqE[0]), __float_as_uint(kqE[1]), __float_as_uint(kqE[2]), __float_as_uint(kqE[3])}; \
            const q_u4 co_ = {__float_as_uint(kqO[0]), __float_as_uint(kqO[1]), __float_as_uint(kqO[2]), __float_as_uint(kqO[3])}; \
            const q_u2 dd_ = {kdE, kdO};                                                            \
            __builtin_amdgcn_raw_buffer_store_b128(ce_, rKc, lane * 32, (C) * 2048, 0);             \
            __builtin_amdgcn_raw_buffer_store_b128(co_, rKc, lane * 32 + 16, (C) * 2048, 0);        \
            __builtin_amdgcn_raw_buffer_store_b64(dd_, rKd, lane * 8, (C) * 512, 0);                \
        }                                                                                           \
    }
            for (int t = 0; t < 5; ++t) Q_SYNC();
            for (int c = 0; c < nbB; ++c) {            // interval 5 + c: recombine feed batch c-1, filter feed batch c
                const int j0 = 4 * c;
                if (c > 0) {
                    Q_SELECT_B2(c - 1)
                    if (ds == 0) {                     // the next batch starts from (+inf, 0)
#pragma unroll
                        for (int k = 0; k < 4; ++k) kqE[k] = kqO[k] = __builtin_inff();
                        kdE = kdO = 0;
                    }
                }
                Q_ISSUE_B2(c)
#define Q_STEP_B2(K)                                                                                \
    {                                                                                               \
        float uE, uO, vE_, vO_;                                                                     \
        Q_READ_B(j0 + K, uE, uO, vE_, vO_)                                                          \
        double h0E, h0O, h1E, h1O;                                                                  \
        q_hsum8x2(uE, uO, i2, h0E, h0O);                                                            \
        q_hsum8x2(vE_, vO_, i2, h1E, h1O);                                                          \
        sv[K][0] = box_out(vstep<K>(t0.e, h0E)); sv[K][1] = box_out(vstep<K>(t0.o, h0O));           \
        sv[K][2] = box_out(vstep<K>(t1.e, h1E)); sv[K][3] = box_out(vstep<K>(t1.o, h1O));           \
    }
                Q_STEP_B2(0) Q_STEP_B2(1) Q_STEP_B2(2) Q_STEP_B2(3)
#undef Q_STEP_B2
                Q_SYNC();
            }
            Q_SELECT_B2(nbB - 1)                       // interval 5 + nbB
            Q_SYNC();
            __builtin_amdgcn_s_waitcnt(0);             // the chunk plane of this slice is in the L2 before the next slice reads it
#undef Q_SELECT_B2
#undef Q_ISSUE_B2
        }
#undef Q_READ_B
    }
    }   // slices of the chunk
#if PSM_Q2_TIMING
    if (lane == 0) {
        atomicAdd(&g_q2_dbg[wave], q_work);
        atomicAdd(&g_q2_dbg[4 + wave], q_wait);
    }
#endif
}

// chunk planes of k_cvf_q2 -> packed WTA key and / or final map per pixel; one thread per record (pair, batch, lane) =
// four rows of two columns
__global__ __launch_bounds__(256) void k_chunk_min2(const float4 *__restrict__ kcost, const uint2 *__restrict__ kdisp, int nchunks, int npairs,
                                                   int nbmax, int ngroups, int seg_rows, int W, int H, long long *__restrict__ keys,
                                                   uint8_t *__restrict__ map)
{
    const size_t nrec = (size_t)npairs * nbmax * 64;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nrec) return;
    const int lane = (int)(idx & 63);
    size_t t = idx >> 6;
    const int c = (int)(t % nbmax);
    const int pair = (int)(t / nbmax);
    const int g = pair % ngroups, seg = pair / ngroups;
    const int x = g * Q_COLS + 2 * lane;
    if (lane >= Q_COLS / 2 || x >= W) return;
    const int y0 = seg * seg_rows, y1 = min(H, y0 + seg_rows);
    const int ya = y0 + 4 * c - 7;
    if (ya + 3 < y0 || ya >= y1) return;
    long long best[2][4];
    for (int chk = 0; chk < nchunks; ++chk) {
        const size_t r = (size_t)chk * nrec + idx;
        const float4 ce = kcost[2 * r], co = kcost[2 * r + 1];
        const uint2 dd = kdisp[r];
        const float cc[2][4] = {{ce.x, ce.y, ce.z, ce.w}, {co.x, co.y, co.z, co.w}};
        const unsigned d2[2] = {dd.x, dd.y};
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const long long key = pack_key_f32(cc[e][k], (d2[e] >> (8 * k)) & 0xff);
                best[e][k] = (chk == 0 || key < best[e][k]) ? key : best[e][k];
            }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int yo = ya + k;
        if (yo < y0 || yo >= y1) continue;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            if (x + e >= W) continue;
            const size_t o = (size_t)yo * W + x + e;
            if (keys) keys[o] = best[e][k];
            if (map) map[o] = (uint8_t)((unsigned long long)best[e][k] & 0xffull);
        }
    }
}

}  // namespace psm

// debug: read and clear the per-role cycle counters of k_cvf_q2 (all zero unless built with -DPSM_Q2_TIMING=1)
extern "C" int psm_debug_q2_cycles(unsigned long long *out8)
{
    for (int i = 0; i < 8; ++i) out8[i] = 0;
#if PSM_Q2_TIMING
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(psm::g_q2_dbg), 8 * sizeof(unsigned long long)) != hipSuccess) return 1;
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(psm::g_q2_dbg), z, sizeof z) != hipSuccess) return 1;
#endif
    return 0;
}

namespace psm {

PcPlan q2_plan(int W, int H, int Dloc, int seg_rows_opt)
{
    PcPlan pl = pc_plan_cols(W, H, Dloc, seg_rows_opt, 1, Q_COLS);
    pl.rec_per_chunk = (size_t)pl.ngroups * pl.nsegs * pl.nbmax * 64;   // one consumer-record wave per workgroup
    pl.rec_bytes = 40;                                                   // 2 x 16 bytes of costs + 8 bytes of disparities
    return pl;
}

void launch_cvf_q2(hipStream_t s, March m, Guidance gd, int W, int H, int Dloc, const float4 *g1_other, int d_begin, int cvc_mode,
                   void *scratch)
{
    const PcPlan pl = q2_plan(W, H, Dloc, m.seg_rows);
    float *kcost = (float *)scratch;                                              // nchunks * rec_per_chunk * 2 float4
    unsigned *kdisp = (unsigned *)(kcost + 8 * pl.rec_per_chunk * pl.nchunks);     // nchunks * rec_per_chunk * 2 dwords
    const int nblocks = 8 * ((pl.ngroups * pl.nsegs * pl.nchunks + 7) / 8);
    if (cvc_mode == 2)
        hipLaunchKernelGGL(k_cvf_q2<2>, dim3(nblocks), dim3(256), 0, s, (const float4 *)gd.g1, (const float4 *)gd.g2, (const float4 *)gd.g3,
                           (const float2 *)gd.g4, g1_other, W, H, Dloc, pl.ngroups, pl.nsegs, pl.seg_rows, d_begin, pl.DC, kcost, kdisp, pl.nbmax);
    else
        hipLaunchKernelGGL(k_cvf_q2<1>, dim3(nblocks), dim3(256), 0, s, (const float4 *)gd.g1, (const float4 *)gd.g2, (const float4 *)gd.g3,
                           (const float2 *)gd.g4, g1_other, W, H, Dloc, pl.ngroups, pl.nsegs, pl.seg_rows, d_begin, pl.DC, kcost, kdisp, pl.nbmax);
}

void launch_chunk_min2(hipStream_t s, March m, int W, int H, int Dloc, void *scratch, long long *keys, uint8_t *map)
{
    const PcPlan pl = q2_plan(W, H, Dloc, m.seg_rows);
    const float *kcost = (const float *)scratch;
    const unsigned *kdisp = (const unsigned *)(kcost + 8 * pl.rec_per_chunk * pl.nchunks);
    hipLaunchKernelGGL(k_chunk_min2, dim3((unsigned)((pl.rec_per_chunk + 255) / 256)), dim3(256), 0, s, (const float4 *)kcost, (const uint2 *)kdisp,
                       pl.nchunks, pl.ngroups * pl.nsegs, pl.nbmax, pl.ngroups, pl.seg_rows, W, H, keys, map);
}

}  // namespace psm
