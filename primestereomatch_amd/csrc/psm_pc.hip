// psm_pc.hip - k_cvf_pc: the product kernel of the DispEst hot path on gfx950 - cost build (CVC) + both box-filter
// rounds of the guided filter (CVF) + (optionally) the winner-takes-all (DispSel) in ONE kernel.
// Arithmetic contract as in psm_kernels.hip: op-for-op fp32, fp64 balanced-tree box sums, no FMA contraction.
// Reference arithmetic: src/CVC.cpp:18-39,122-179, src/CVF.cpp:72-165, src/DispSel.cpp:83-109.
//
// Stage A and stage B of the guided filter are chained without the (a0,a1,a2,b) round trip through HBM (16 B/voxel
// written and read again by the two-stage kernels).  Both sliding trees in one wave need 250 VGPRs (1-2 waves per
// SIMD; measured 8.3-9.3 ms per 1080p x 256 volume, latency bound), so the two halves run in DIFFERENT waves of
// one workgroup and the linear models are handed over through LDS:
//   producer waves ("A"): cost p (read, or built from the g1 planes), g1 -> window sums -> solve -> model rows
//                         into an LDS ring of PC_RING batches of four rows
//   consumer waves ("B"): model rows from the ring (REFLECT_101 of the model planes = ring index arithmetic,
//                         columns and rows, so the image border needs no separate kernel) -> window sums -> q
// One barrier per batch of four rows; B runs two batches behind A.
//
// MODE 0 ("store"): q is written to the filtered volume - merged 16-byte-per-lane rows one batch behind B; layout
//   2 A waves x 52 model columns -> 2 B waves x 48 output columns = 96 outputs = three whole 128-byte lines per row.
// MODE 1 ("select"): the filtered volume stays VIRTUAL.  A workgroup walks a chunk of DC consecutive slices of its
//   (column group, segment) and the consumer waves keep the running strict-'<' minimum of q over the chunk in a
//   per-chunk scratch plane private to the workgroup: per consumer wave and batch of four rows one 16-byte-per-lane
//   record of costs and one 4-byte-per-lane record of disparities, laid out [wave][batch][lane] so that a wave reads
//   and writes whole contiguous kilobytes (read - compare - store by the one lane that owns the pixel: no atomics, no
//   races).  The planes are touched once per slice and live in L2 / MALL; k_chunk_min then reduces the Dloc/DC chunk
//   planes to the packed WTA key (or the map) per pixel.  (First version: image-layout planes, one dword load + a
//   dword and a byte store per row step - slower than storing q, 4.50 vs 4.20 + 0.32 ms: with DC = 8 nearly every row
//   step improves some lane, and unaligned 4-byte-per-lane stores are the slowest thing the memory path can do.)  Per voxel this replaces a 4-byte HBM store + a 4-byte HBM read (k_wta) by a cached 4-byte
//   read and a rare store, and - nothing being stored per row - the 128-byte alignment rule of MODE 0 no longer binds:
//   2 x 57 model columns feed 54 + 53 output columns (84-89 % useful lanes instead of 75-81 %).
// MODE 2 ("select, shared keys"): as MODE 1, but without chunk planes: one plane of packed WTA keys per volume (8 bytes per
//   pixel, image layout - the final result), initialised to key(+inf, 0).  A consumer lane loads the current key of its
//   pixel one batch ahead (a stale value only costs a redundant atomic), compares, and only where q beats it issues a 64-bit
//   atomicMin - a handful per pixel over all 256 slices instead of a record per slice.  One slice per workgroup (the finest
//   work granularity, shortest tail), no reduction kernel afterwards.
// d = 0 is never a candidate, NaN never wins, the lowest d wins ties (src/DispSel.cpp:91-105): slices are visited in
// ascending d with strict '<' inside a chunk, and k_chunk_min takes the signed minimum of pack_key_f32 across chunks.
#include "psm_kernels.h"
#include "psm_cost.h"
#include "psm_dev.h"

#include <cstdlib>

#ifndef PSM_PC_TIMING
#define PSM_PC_TIMING 0   // 1: every wave accumulates its cycles between barriers (work) and inside them (wait) per role
#endif

namespace psm {

#if PSM_PC_TIMING
__device__ unsigned long long g_pc_dbg[8];   // work cycles of waves A0, A1, B0, B1, then their barrier-wait cycles
#define PC_SYNC()                                                                \
    {                                                                            \
        const unsigned long long a_ = __builtin_readcyclecounter();              \
        __syncthreads();                                                         \
        const unsigned long long b_ = __builtin_readcyclecounter();              \
        q_work += a_ - q_mark; q_wait += b_ - a_; q_mark = b_;                   \
    }
#else
#define PC_SYNC() __syncthreads()
#endif

typedef float f4v __attribute__((ext_vector_type(4)));

#ifndef PSM_PC_ATTR
#define PSM_PC_ATTR
#endif
// Occupancy of the key form (MODE 2, 5/6 of the slices of a 256-slice volume): four workgroups per CU instead of three.
// The kernel is latency-sensitive at 3 waves per SIMD (barrier waits, LDS round trips); 128 VGPRs need a shorter load
// look-ahead (PSM_PC_LEAN bit 0: consumer G1 / keys two rows ahead instead of a batch; bit 1: producer guidance planes issued
// at the start of their own step, partner pixels right after the cost is formed) and cost 4 spilled registers per producer batch.
// Measured at 1080p x 256: key phase 6.36 -> 5.86 ms (PSM_PC_LEAN 1 or 3; 0 with the cap: 65 spills, 12.7 ms per frame).
// PSM_PC_OCC4 == 2 caps the plane form (MODE 1) as well; its own register diet (PSM_PC_LEAN bits 4-7: G1 two rows ahead, the
// producer changes, selection row by row inside the steps, role constants recomputed per slice from an opaque lane index so
// that they do not stay alive through the other role's code) gets it from 159 to 139 registers / 9 spills under the cap, but
// there the shorter look-ahead costs what the fourth workgroup gains (720p x 128 1.91 vs 1.92 ms, 450 x 375 x 64 0.32 vs 0.31,
// 8-bit 450 x 375 0.45 vs 0.32): off by default.
#ifndef PSM_PC_OCC4
#define PSM_PC_OCC4 1
#endif
#ifndef PSM_PC_LEAN
#define PSM_PC_LEAN 3
#endif
constexpr int PC_RING = 4;   // batches of four model rows kept in LDS
#ifndef PSM_PC_NT
#define PSM_PC_NT 1          // MODE 0: nontemporal stores of the output rows (the volume is next read long after it left the L2)
#endif

template <int MODE> struct PcLayout;
template <> struct PcLayout<0> { static constexpr int NA = 2, NB = 2, OUT_A = 52, OUT_B = 48, COLS = 96; };
template <> struct PcLayout<1> { static constexpr int NA = 2, NB = 2, OUT_A = 57, OUT_B = 54, COLS = 107; };
template <> struct PcLayout<2> : PcLayout<1> {};
#ifndef PSM_K_LD_AUX
#define PSM_K_LD_AUX 0       // cache policy of MODE 1's record loads.  A record is only ever read by the lane that wrote it
#endif                       // (one slice earlier), and a thread always observes its own stores: plain cached loads are
                             // coherent here.  (16 = sc1 bypasses the L2 as well: measured 25 % slower kernel.)
#ifndef PSM_KEY_NOATOMIC
#define PSM_KEY_NOATOMIC 0   // experiment (invalid results): skip the atomics
#endif
#ifndef PSM_KEY_LD_AUX
#define PSM_KEY_LD_AUX 16    // cache policy of MODE 2's key loads: 16 = sc1 (agent scope: never from this CU's L1)
#endif

typedef unsigned pc_u2 __attribute__((ext_vector_type(2)));
typedef unsigned pc_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pc_rsrc(const void *p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float pc_load1(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ float pc_load1_l2(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{   // sc1: served by the L2, never by this CU's L1 (the line was last written by this same wave one slice earlier)
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 16));
}
__device__ __forceinline__ float2 pc_load2(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    const pc_u2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
}
__device__ __forceinline__ float4 pc_load4(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    const pc_u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

typedef unsigned pc_u3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ float3 pc_load3(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{   // the first three floats of a float4 element (the consumer waves never use g1.w)
    const pc_u3 v = __builtin_amdgcn_raw_buffer_load_b96(r, voff, soff, 0);
    return make_float3(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z));
}

// Second set of plane pointers for CVC == 3: one launch filters BOTH volumes (blockIdx.y = side; side 0 = left volume with
// the kernel's own arguments, side 1 = right volume with these) - twice the workgroups per launch, half the launches.
struct PcSide {
    const float4 *G1, *G2, *G3;
    const float2 *G4;
    const float4 *Gother;
    float *kcost;
    unsigned *kdisp;
};

// MODE 1, dynamic form: NW workgroups per (column group, segment) pair, all resident at once, take the pair's slices one
// after the other from a device counter (ascending d) and keep their running minima in ONE plane each - NW planes per
// volume instead of Dloc / DC, all workgroups finish within one slice of each other (no tail), and stores to the plane
// become rare (a lane rewrites its record only when one of its four rows improved).  cnt == NULL: static chunks of DC.
struct PcDyn {
    int *cnt;      // one zeroed counter per (side, pair)
    int NW;
    // Which local slices this launch covers (Dloc = their number): sel 0 all (index i = local slice i); sel 1 every
    // step-th slice (i -> i * step); sel 2 the others (i -> (i / (step-1)) * step + i % (step-1) + 1).  Two-phase
    // selection: a first launch reduces every step-th slice to the key plane, a MODE 2 launch then runs the rest
    // against that plane - its consumer lanes find a tight bound there and issue an atomic only a few times per pixel.
    // unit > 1 (sel 2 only): the same pattern in units of `unit` slices - the multiples of unit that are not multiples of
    // unit * step (a middle phase of a three-phase selection).
    int sel, step, unit;
};
__device__ __forceinline__ int pc_slice(const PcDyn &o, int i)
{
    return o.sel == 1 ? i * o.step : (o.sel == 2 ? ((i / (o.step - 1)) * o.step + i % (o.step - 1) + 1) * o.unit : i);
}

// CVC = 0: the cost slice is read from `vin`.  CVC = 1 (left volume) / 2 (right volume): the cost volume is never
// materialised - the producer waves evaluate myCostGrd (src/CVC.cpp:18-39) for their input column on the fly from the
// two g1 planes (`G1` = this side's image, `Gother` = the other one), exactly as k_cvc does.
// U8 (8-bit char mode, select mode with costs on the fly only): the producer waves build the 8-bit matching cost of
// assets/cvc.cl:279-301 (oracle: cost_u8) from the {c0,c1,c2,grad} byte planes handed over in `vin` (this side) and `vout`
// (other side) and filter cost * (1/255.0f); the consumer waves re-quantise q8 = sat_u8(rintf(q * 255)) before the
// strict-'<' selection (assets/dispsel.cl:41-62 with the initial minimum above 255) - the build-defined 8-bit contract
// of oracle/psm_oracle.h, in one pass and without an 8-bit volume in memory.
template <bool VEC4, int CVC, int MODE, bool U8 = false>
__global__ __launch_bounds__(64 * (PcLayout<MODE>::NA + PcLayout<MODE>::NB)) PSM_PC_ATTR
#if PSM_PC_OCC4   // the key form (MODE 2) capped at 128 VGPRs = four workgroups per CU
__attribute__((amdgpu_waves_per_eu((MODE == 2 || (MODE == 1 && PSM_PC_OCC4 == 2)) ? 4 : 1, (MODE == 2 || (MODE == 1 && PSM_PC_OCC4 == 2)) ? 4 : 8)))
#endif
void k_cvf_pc(
    const float *__restrict__ vin, float *__restrict__ vout, const float4 *__restrict__ G1a, const float4 *__restrict__ G2a,
    const float4 *__restrict__ G3a, const float2 *__restrict__ G4a, int W, int H, int Dloc, int ngroups, int nsegs, int seg_rows,
    int ybeg, int yend, const float4 *__restrict__ Gothera, int d_begin, int DC, float *__restrict__ kcosta, unsigned *__restrict__ kdispa, int nbmax,
    PcSide side1, PcDyn dyn)
{
    const bool right = CVC == 2 || (CVC == 3 && blockIdx.y == 1);      // buildCV_right arithmetic (uniform)
    const bool s1 = CVC == 3 && blockIdx.y == 1;
    const float4 *const G1 = s1 ? side1.G1 : G1a, *const G2 = s1 ? side1.G2 : G2a, *const G3 = s1 ? side1.G3 : G3a;
    const float2 *const G4 = s1 ? side1.G4 : G4a;
    const float4 *const Gother = s1 ? side1.Gother : Gothera;
    float *const kcost = s1 ? side1.kcost : kcosta;
    unsigned *const kdisp = s1 ? side1.kdisp : kdispa;
    using L = PcLayout<MODE>;
    constexpr int PC_NA = L::NA, PC_NB = L::NB, PC_OUT_A = L::OUT_A, PC_OUT_B = L::OUT_B, PC_COLS = L::COLS;
    constexpr int PC_MCOLS = PC_NA * PC_OUT_A;   // model columns per workgroup (>= PC_COLS + 7)
    // Select forms: the x 1/64 of the two box filters is not applied per window sum (a v_ldexp_f64 per channel and step).  The
    // guided filter is homogeneous of degree one in (box(p), box(I p)) and in (box(a), box(b)), and a power-of-two scale commutes
    // with every fp32 rounding as long as nothing under- or overflows: the producers hand over 64 x (a0, a1, a2, b), the consumers
    // form 4096 x q and one fp32 multiply by 2^-12 (exact) restores q - bit-identical to the per-sum scaling while no intermediate
    // of a voxel is subnormal or beyond 2^115 (costs are O(1); tests/test_gpu_parity.py::test_scaled_sums_domain).  The storing
    // form (MODE 0) keeps the per-sum scaling, i.e. the oracle's arithmetic op for op.
#ifndef PSM_PC_SCALED
#define PSM_PC_SCALED 1
#endif
    constexpr bool SCALED = PSM_PC_SCALED && MODE != 0;
    constexpr float QSCALE = SCALED ? 0x1p-12f : 1.0f;
#define PSM_BOX(N) (SCALED ? (float)(N) : box_out(N))
    static_assert(PC_MCOLS >= PC_COLS + 7 && PC_OUT_A <= 57 && PC_OUT_B <= 57 && (MODE != 0 || (PC_COLS % 4) == 0), "bad producer/consumer layout");
    // Model rows live in a ring of PC_RING batches of four rows; consumers run two batches behind the
    // producers, so every model row the second box filter can ask for - including the REFLECT_101 rows at
    // the top and bottom of the image, which are earlier/later rows of the same ring - is still present.
    __shared__ __attribute__((aligned(16))) float4 ring[PC_RING][4][PC_MCOLS];
    __shared__ __attribute__((aligned(16))) float qbuf[MODE == 0 ? 2 : 1][MODE == 0 ? 4 : 1][MODE == 0 ? PC_COLS : 4];   // MODE 0: output rows, two batches
    // Workgroup -> (column group, segment, slice chunk).  Blocks are observed to go round-robin over the 8 XCDs
    // (block b -> XCD b%8): every XCD owns a contiguous range of (group, segment) pairs and walks the
    // chunks of one pair back to back, so the guidance rows its resident workgroups are reading (few
    // pairs, neighbouring rows, many slices) fit its 4 MB L2 instead of coming from the MALL.  Speed only.
#ifndef PSM_PC_NODYN
#define PSM_PC_NODYN 0
#endif
    const bool dynamic = !PSM_PC_NODYN && MODE == 1 && dyn.cnt != nullptr;
    const int nchunks = dynamic ? dyn.NW : (Dloc + DC - 1) / DC;
    int id = blockIdx.x;
    const int npairs = ngroups * nsegs;               // (column group, segment) pairs
    // work items (pair, chunk), pair-major; XCD x takes the x-th eighth of them: a contiguous range of pairs whose chunks
    // run back to back, and equal work per XCD whatever the pair count
    const int nitems = npairs * nchunks;
    const int ipx = (nitems + 7) >> 3;
    const int xcd = id & 7, jj = id >> 3;
    const int item = xcd * ipx + jj;
    if (jj >= ipx || item >= nitems) return;
    const int pair = item / nchunks, ch = item % nchunks;
    const int g = pair % ngroups, seg = pair / ngroups;
    // (which hardware wave takes which role does not matter: swapping / interleaving the producer and consumer
    // waves, per workgroup or pseudo-randomly, changed nothing - the CU balances the SIMDs itself)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;   // wave index in an SGPR: everything derived from it is scalar
    const bool is_a = wave < PC_NA;
    const int xg = g * PC_COLS;                       // first output column of the workgroup
    const int xm0 = xg - 4;                           // first model column of the workgroup
    const int y0 = ybeg + seg * seg_rows, y1 = min(yend, y0 + seg_rows);   // output rows [y0, y1)
    const int mstart = max(0, y0 - 4);                // model rows produced: mstart .. mend
    const int mend = min(H - 1, max(y1 + 2, 4 - y0));   // (a one-row segment at the top still needs model row 4: REFLECT_101 of row -4)
    const int nbA = (mend - mstart + 1 + 3) >> 2;     // producer batches
    const int nf = (y1 - y0) + 7;                     // consumer feeds (model rows y0-4 .. y1+2, reflected)
    const int nbB = (nf + 3) >> 2;                    // consumer batches
    const int iters = nbB + 3;                        // barriers executed by every wave per slice
    const int i1 = ((lane + 1) & 63) << 2, i2 = ((lane + 2) & 63) << 2, i4 = ((lane + 4) & 63) << 2;
    (void)i1;
    const size_t HW = (size_t)H * W;

#if PSM_PC_TIMING
    unsigned long long q_work = 0, q_wait = 0, q_mark = __builtin_readcyclecounter();
#endif
    const int nds = MODE == 1 ? DC : 1;               // MODE 0 / 2 always run with DC == 1 (one slice per workgroup)
    __shared__ int s_next;
    bool first = true;                                // no slice processed yet: the plane holds nothing
    for (int ds = 0; ; ++ds) {                        // the slices of this chunk / this workgroup's share of the pair, ascending d
    int d;
    if (MODE == 1 && dynamic) {
        if (threadIdx.x == 0) s_next = atomicAdd(dyn.cnt + (s1 ? npairs : 0) + pair, 1);
        __syncthreads();                              // (the next write of s_next is many barriers away)
        d = __builtin_amdgcn_readfirstlane(s_next);
    } else {
        if (ds >= nds) break;
        d = ch * DC + ds;
    }
    if (MODE == 1 && d >= Dloc) break;                // uniform over the workgroup
    if (is_a) {
        // ---------------- producer: stage A ----------------
        // step s reads input row mstart-5+s; from step 8 on it yields model row mstart+(s-8)
        const int xa0 = xm0 + wave * PC_OUT_A;        // first model column of this wave
        // (plane form, several slices per workgroup: the per-lane constants of a role are recomputed per slice from an opaque
        // copy of the lane index - hoisted out of the slice loop they would stay alive through the other role's code)
        int lane_r = lane;
        if (MODE == 1 && (PSM_PC_LEAN & 128)) asm volatile("" : "+v"(lane_r));
        const int ci = r101c(xa0 - 4 + lane_r, W);    // input column of this lane
        const int xa = xa0 + lane_r;                  // model column of this lane
        const int xac = xa < 0 ? 0 : (xa > W - 1 ? W - 1 : xa);
        const bool mvalid = lane < PC_OUT_A;
        const float *vd = vin + (size_t)pc_slice(dyn, d) * HW;
        const int dg = d_begin + pc_slice(dyn, d);    // global disparity of this slice
        // buildCV_left: partner x-d while x >= d; buildCV_right: partner x+d while x < W-d (src/CVC.cpp:135-146,165-176)
        const bool inb = right ? (ci < W - dg) : (ci >= dg);
        const int cpart = right ? min(ci + dg, W - 1) : max(ci - dg, 0);
        const bool any_border = CVC != 0 && __builtin_amdgcn_ballot_w64(!inb) != 0;
        VTree t0 = {}, t1 = {}, t2 = {}, t3 = {};
        float pin[2];
        float4 oth[2], gin[2], o2[2], o3[2];
        float2 o4[2];
        constexpr bool LEANA = (PSM_PC_LEAN & 2) != 0 && (MODE == 2 || (MODE == 1 && (PSM_PC_LEAN & 32)));
        // raw buffer loads: descriptors and row offsets in scalar registers, one constant 32-bit byte offset per lane
        // (psm_create keeps W*H < 2^27, so every byte offset into a 16-byte plane fits 31 bits)
        const __amdgpu_buffer_rsrc_t rG1 = pc_rsrc(G1, (unsigned)HW * 16u), rG2 = pc_rsrc(G2, (unsigned)HW * 16u);
        const __amdgpu_buffer_rsrc_t rG3 = pc_rsrc(G3, (unsigned)HW * 16u), rG4 = pc_rsrc(G4, (unsigned)HW * 8u);
        const __amdgpu_buffer_rsrc_t rGo = pc_rsrc(CVC == 0 ? G1 : Gother, (unsigned)HW * 16u);
        const __amdgpu_buffer_rsrc_t rV = pc_rsrc(CVC == 0 ? (const void *)vd : (const void *)G1, (unsigned)HW * 4u);
        // U8: byte planes {c0,c1,c2,grad} of this side's / the other side's image (swapped for the right volume of a two-side launch)
        const __amdgpu_buffer_rsrc_t rP = pc_rsrc(U8 ? (s1 ? (const void *)vout : (const void *)vin) : (const void *)G1, (unsigned)HW * 4u);
        const __amdgpu_buffer_rsrc_t rPo = pc_rsrc(U8 ? (s1 ? (const void *)vin : (const void *)vout) : (const void *)G1, (unsigned)HW * 4u);
        unsigned pu[2], po[2];
        const int vci = ci * 16, vcp = cpart * 16, vxa = xac * 16;
#define PSM_ISSUE_PA(SLOT, STEP)                                                        \
    {                                                                                   \
        const int row_ = r101c(mstart - 5 + (STEP), H) * W;                             \
        int ya_ = mstart - 8 + (STEP);                                                  \
        ya_ = ya_ < 0 ? 0 : (ya_ > H - 1 ? H - 1 : ya_);                                \
        const int oa_ = ya_ * W;                                                        \
        if (CVC == 0) pin[SLOT] = pc_load1(rV, vci >> 2, row_ * 4);                     \
        else if (U8) {                                                                  \
            pu[SLOT] = __builtin_amdgcn_raw_buffer_load_b32(rP, vci >> 2, row_ * 4, 0); \
            po[SLOT] = __builtin_amdgcn_raw_buffer_load_b32(rPo, vcp >> 2, row_ * 4, 0); \
        } else if (!LEANA) oth[SLOT] = pc_load4(rGo, vcp, row_ * 16);                   \
        gin[SLOT] = pc_load4(rG1, vci, row_ * 16);                                      \
        if (!LEANA) {                                                                   \
            o2[SLOT] = pc_load4(rG2, vxa, oa_ * 16);                                    \
            o3[SLOT] = pc_load4(rG3, vxa, oa_ * 16);                                    \
            o4[SLOT] = pc_load2(rG4, vxa >> 1, oa_ * 8);                                \
        }                                                                               \
    }
#define PSM_ISSUE_PA2(STEP)   /* PSM_PC_LEAN: the guidance planes of step STEP, issued when the step starts */ \
    {                                                                                   \
        int ya_ = mstart - 8 + (STEP);                                                  \
        ya_ = ya_ < 0 ? 0 : (ya_ > H - 1 ? H - 1 : ya_);                                \
        const int oa_ = ya_ * W;                                                        \
        o2[0] = pc_load4(rG2, vxa, oa_ * 16);                                           \
        o3[0] = pc_load4(rG3, vxa, oa_ * 16);                                           \
        o4[0] = pc_load2(rG4, vxa >> 1, oa_ * 8);                                       \
    }
        // one step: consume the loads of step S (slot K&1), issue those of step S+1
#define PSM_STEP_PA(K, S, DST)                                                                      \
    {                                                                                               \
        PSM_ISSUE_PA((K + 1) & 1, (S) + 1)                                                          \
        if (LEANA) PSM_ISSUE_PA2(S)                                                                 \
        float p;                                                                                    \
        if (CVC == 0) p = pin[K & 1];                                                               \
        else if (U8) {                                                                              \
            const unsigned b_ = inb ? po[K & 1] : 0xffffffffu;        /* border: the other image reads as 255 */ \
            const unsigned clr_ = __builtin_amdgcn_sad_u8(pu[K & 1] & 0xffffffu, b_ & 0xffffffu, 0u); \
            const int gd_ = (int)(pu[K & 1] >> 24) - (int)(b_ >> 24);                               \
            const float f_ = __fadd_rn(__fmul_rn(0.9f, (float)(clr_ / 3u)), __fmul_rn(__fsub_rn(1.0f, 0.9f), (float)(gd_ < 0 ? -gd_ : gd_))); \
            p = __fmul_rn((float)(unsigned)(unsigned char)f_, 1 / 255.0f);                          \
        } else {                                                                                    \
            p = cost_pair(gin[K & 1], oth[LEANA ? 0 : (K & 1)]);                                    \
            if (LEANA) oth[0] = pc_load4(rGo, vcp, r101c(mstart - 5 + (S) + 1, H) * W * 16);   /* partner pixels of the next step: the current ones are dead now */ \
            if (any_border) {   /* only where x < d (left) / x >= W-d (right) occurs in this wave */   \
                asm volatile("; border cost");   /* keeps this a real branch */                     \
                const float cb_ = cost_border(gin[K & 1]);                                          \
                p = inb ? p : cb_;                                                                  \
            }                                                                                       \
        }                                                                                           \
        double h0 = hsum8(p, i1, i2, i4);                                                           \
        double h1 = hsum8(__fmul_rn(gin[K & 1].x, p), i1, i2, i4);                                  \
        double h2 = hsum8(__fmul_rn(gin[K & 1].y, p), i1, i2, i4);                                  \
        double h3 = hsum8(__fmul_rn(gin[K & 1].z, p), i1, i2, i4);                                  \
        double n0 = vstep<K>(t0, h0), n1 = vstep<K>(t1, h1), n2 = vstep<K>(t2, h2), n3 = vstep<K>(t3, h3); \
        float4 r = solve_ab(PSM_BOX(n0), PSM_BOX(n1), PSM_BOX(n2), PSM_BOX(n3), o2[LEANA ? 0 : (K & 1)], o3[LEANA ? 0 : (K & 1)], o4[LEANA ? 0 : (K & 1)]); \
        if ((DST) != nullptr && mvalid) (DST)[K * PC_MCOLS] = r;                                    \
        __builtin_amdgcn_sched_barrier(0);                                                          \
    }
        PSM_ISSUE_PA(0, 0)
        if (LEANA && CVC != 0 && !U8) oth[0] = pc_load4(rGo, vcp, r101c(mstart - 5, H) * W * 16);
        __builtin_amdgcn_sched_barrier(0);
        {   // warm-up: 8 rows fill the tree (loop bodies stay free of conditionals around the tree updates:
            // a conditional turns the trees into loop-carried phis and doubles their registers)
            float4 *const none = nullptr;
            PSM_STEP_PA(0, 0, none) PSM_STEP_PA(1, 1, none) PSM_STEP_PA(2, 2, none) PSM_STEP_PA(3, 3, none)
            PSM_STEP_PA(0, 4, none) PSM_STEP_PA(1, 5, none) PSM_STEP_PA(2, 6, none) PSM_STEP_PA(3, 7, none)
        }
#define PSM_BATCH_PA(B)                                                                            \
        {                                                                                          \
            const int s0 = 8 + (B) * 4;                                                            \
            float4 *dst = &ring[(B) & (PC_RING - 1)][0][wave * PC_OUT_A + lane];                   \
            PSM_STEP_PA(0, s0, dst) PSM_STEP_PA(1, s0 + 1, dst) PSM_STEP_PA(2, s0 + 2, dst) PSM_STEP_PA(3, s0 + 3, dst) \
            PC_SYNC();                                                                             \
        }
        // two batches per iteration: the s4 slots of the vertical trees change registers with every update, so only after eight
        // steps is every value back in the register the loop header expects (one batch per iteration: 16 v_mov_b64 at the latch)
#ifndef PSM_PC_UNROLL2
#define PSM_PC_UNROLL2 1
#endif
        for (int b = 0; b < nbA; b += PSM_PC_UNROLL2 ? 2 : 1) {
            PSM_BATCH_PA(b)
            if (PSM_PC_UNROLL2 && __builtin_expect(b + 1 < nbA, 1)) PSM_BATCH_PA(b + 1)
        }
#undef PSM_BATCH_PA
        for (int b = nbA; b < iters; ++b) PC_SYNC();
#undef PSM_STEP_PA
#undef PSM_ISSUE_PA
#undef PSM_ISSUE_PA2
    } else {
        // ---------------- consumer: stage B ----------------
        // feed j (j = 0 .. nf-1) is model row r101(y0-4+j); from feed 7 on the tree yields output row y0+j-7
        const int wb = wave - PC_NA;
        const int bwidth = wb < PC_NB - 1 ? PC_OUT_B : PC_COLS - (PC_NB - 1) * PC_OUT_B;
        const int xb0 = xg + wb * PC_OUT_B;           // first output column of this wave
        int lane_r = lane;                            // (see the producer branch)
        if (MODE == 1 && (PSM_PC_LEAN & 128)) asm volatile("" : "+v"(lane_r));
        const int xmod = xb0 - 4 + lane_r;            // model column this lane consumes
        int mc = r101(xmod, W) - xm0;                 // REFLECT_101 of the model planes, as ring column
        mc = mc < 0 ? 0 : (mc > PC_MCOLS - 1 ? PC_MCOLS - 1 : mc);
        const int xb = xb0 + lane_r;                  // output column of this lane
        const int xbc = min(xb, W - 1);
        float *od = vout + (MODE == 0 ? (size_t)pc_slice(dyn, d) * HW : 0);
        const int amax = 4 * nbA - 1;
        VTree t0 = {}, t1 = {}, t2 = {}, t3 = {};
        float o1x[4], o1y[4], o1z[4];                 // g1.xyz at (output row, output column), one batch ahead
        // MODE 1: this wave's records of the chunk plane: [batch][lane] float4 costs / uchar4 disparities of the batch's four rows
        // (compact: PC_COLS records per batch, this wave's bwidth lanes at column offset wb * PC_OUT_B; halo lanes never store)
        const size_t krec = (size_t)(ch * npairs + pair) * nbmax * PC_COLS + wb * PC_OUT_B;
        // (own descriptors: offsets stay small, and the loads can carry sc1 = served by the L2, never by this CU's L1 -
        // the record was last written by this same wave one slice earlier)
        const __amdgpu_buffer_rsrc_t rKc = pc_rsrc(MODE == 1 ? (const void *)(reinterpret_cast<float4 *>(kcost) + krec) : (const void *)G1, (unsigned)nbmax * PC_COLS * 16u);
        const __amdgpu_buffer_rsrc_t rKd = pc_rsrc(MODE == 1 ? (const void *)(kdisp + krec) : (const void *)G1, (unsigned)nbmax * PC_COLS * 4u);
#define PSM_K_LOAD(C)                                                                              \
    {                                                                                              \
        const pc_u4 v_ = __builtin_amdgcn_raw_buffer_load_b128(rKc, lane * 16, (C) * (PC_COLS * 16), PSM_K_LD_AUX); \
        kq = make_float4(__uint_as_float(v_.x), __uint_as_float(v_.y), __uint_as_float(v_.z), __uint_as_float(v_.w)); \
        kd4 = __builtin_amdgcn_raw_buffer_load_b32(rKd, lane * 4, (C) * (PC_COLS * 4), PSM_K_LD_AUX); \
    }
        float4 kq = make_float4(__builtin_inff(), __builtin_inff(), __builtin_inff(), __builtin_inff());   // running minima of the current batch
        unsigned kd4 = 0;                                                                                  // and their disparities
        // MODE 2: the volume's key plane (handed over in the kcost argument), current keys of this lane's pixel one batch ahead
        long long *const keyp = reinterpret_cast<long long *>(kcost);
        const __amdgpu_buffer_rsrc_t rKy = pc_rsrc(MODE == 2 ? (const void *)keyp : (const void *)G1, (unsigned)HW * 8u);
        long long kcur[4] = {0, 0, 0, 0};
        float acc_dbg = 0.f; (void)acc_dbg;
#define PSM_KEY_LOAD1(SLOT, J)   /* PSM_PC_LEAN: the key of feed row J alone */                     \
    {                                                                                              \
        int yk_ = y0 + (J) - 7;                                                                    \
        yk_ = yk_ < 0 ? 0 : (yk_ > H - 1 ? H - 1 : yk_);                                           \
        const pc_u2 v_ = __builtin_amdgcn_raw_buffer_load_b64(rKy, xbc * 8, yk_ * W * 8, PSM_KEY_LD_AUX); \
        kcur[SLOT] = (long long)(((unsigned long long)v_.y << 32) | v_.x);                         \
    }
#define PSM_KEY_LOAD(C)                                                                            \
    {                                                                                              \
        _Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_) {                                         \
            int yk_ = y0 + 4 * (C) + k_ - 7;                                                       \
            yk_ = yk_ < 0 ? 0 : (yk_ > H - 1 ? H - 1 : yk_);                                       \
            const pc_u2 v_ = __builtin_amdgcn_raw_buffer_load_b64(rKy, xbc * 8, yk_ * W * 8, PSM_KEY_LD_AUX); \
            kcur[k_] = (long long)(((unsigned long long)v_.y << 32) | v_.x);                       \
        }                                                                                          \
    }
        const __amdgpu_buffer_rsrc_t rG1 = pc_rsrc(G1, (unsigned)HW * 16u);
        const int vxb = xbc * 16;
        const int dg = d_begin + pc_slice(dyn, d);
        const bool lane_out = lane < bwidth && xb < W;   // this lane owns an output pixel
#define PSM_ISSUE_PB(SLOT, J)                                                           \
    {                                                                                   \
        int yb_ = y0 + (J) - 7;                                                         \
        yb_ = yb_ < 0 ? 0 : (yb_ > H - 1 ? H - 1 : yb_);                                \
        const float3 g_ = pc_load3(rG1, vxb, yb_ * W * 16);                             \
        o1x[SLOT] = g_.x; o1y[SLOT] = g_.y; o1z[SLOT] = g_.z;                           \
    }
        // ring address of the model row that feed J consumes (wave-uniform arithmetic)
        auto model_of = [&](int J) -> const float4 * {
            int a = r101(y0 - 4 + J, H) - mstart;
            a = a < 0 ? 0 : (a > amax ? amax : a);
            return &ring[(a >> 2) & (PC_RING - 1)][a & 3][mc];
        };
        // MODE 0: merged store of output batch `c` (rows parked in qbuf[c & 1] one iteration earlier)
        auto store_batch = [&](int c) {
            if constexpr (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {          // row k of the batch is stored by B wave k % PC_NB
                const int j = 4 * c + k;
                const int yo = y0 + j - 7;
                if (k % PC_NB == wb && j >= 7 && yo < y1) {
                    float *row = od + (size_t)yo * W + xg;
                    const float *src = &qbuf[c & 1][k][0];
                    if (VEC4) {
                        const int cc = lane * 4;
                        if (lane < PC_COLS / 4 && xg + cc < W) {
#if PSM_PC_NT
                            __builtin_nontemporal_store(*reinterpret_cast<const f4v *>(src + cc), reinterpret_cast<f4v *>(row + cc));
#else
                            *reinterpret_cast<float4 *>(row + cc) = *reinterpret_cast<const float4 *>(src + cc);
#endif
                        }
                    } else {
#pragma unroll
                        for (int cc = lane; cc < PC_COLS; cc += 64)
                            if (xg + cc < W) row[cc] = src[cc];
                    }
                }
            }
            }
        };
        constexpr bool LEANB = (PSM_PC_LEAN & 1) != 0 && MODE == 2;
        constexpr bool LEANG = LEANB || ((PSM_PC_LEAN & 16) != 0 && MODE == 1);   // G1 of the output rows two rows ahead instead of a batch
        constexpr bool LEANS = (PSM_PC_LEAN & 64) != 0 && MODE == 1;              // plane form: selection row by row inside the steps
        PSM_ISSUE_PB(0, 0) PSM_ISSUE_PB(1, 1)
        if (!LEANG) { PSM_ISSUE_PB(2, 2) PSM_ISSUE_PB(3, 3) }
        if constexpr (MODE == 1) {
            if (!first) {                              // records of batch 0, written by this wave one slice earlier (L1 bypassed)
                PSM_K_LOAD(0)
            }
        }
        if constexpr (MODE == 2) {
            if (LEANB) { PSM_KEY_LOAD1(0, 0) PSM_KEY_LOAD1(1, 1) }
            else PSM_KEY_LOAD(0)
        }
        PC_SYNC();                               // iteration 0
        PC_SYNC();                               // iteration 1
        // (two batches per loop iteration, as in the producer: no register moves of the s4 slots at the latch)
        auto batch_b = [&](const int c) __attribute__((always_inline)) {   // consume feed batch c
            if (c >= 1) store_batch(c - 1);
            {
                const int j0 = 4 * c;
                float4 a_cur = *model_of(j0), a_nxt;
                float qv[4];
                bool anyb = false; (void)anyb;
#define PSM_STEP_PB(K)                                                                              \
    {                                                                                               \
        if (K < 3) a_nxt = *model_of(j0 + K + 1);     /* model row of the next feed, one step ahead */ \
        double h0 = hsum8(a_cur.x, i1, i2, i4);                                                     \
        double h1 = hsum8(a_cur.y, i1, i2, i4);                                                     \
        double h2 = hsum8(a_cur.z, i1, i2, i4);                                                     \
        double h3 = hsum8(a_cur.w, i1, i2, i4);                                                     \
        double n0 = vstep<K>(t0, h0), n1 = vstep<K>(t1, h1), n2 = vstep<K>(t2, h2), n3 = vstep<K>(t3, h3); \
        qv[K] = __fadd_rn(__fadd_rn(__fadd_rn(PSM_BOX(n3), __fmul_rn(PSM_BOX(n0), o1x[LEANG ? (K & 1) : K])),        \
                                    __fmul_rn(PSM_BOX(n1), o1y[LEANG ? (K & 1) : K])), __fmul_rn(PSM_BOX(n2), o1z[LEANG ? (K & 1) : K])); \
        if (U8) {   /* q8 = sat_u8(rintf(q * 255)), NaN -> 0 (oracle: quant_u8); kept as a float: the selection is unchanged */ \
            const float r_ = rintf(__fmul_rn(qv[K], 255.0f * QSCALE));   /* (255 * 2^-12 is exact: one rounding, as q * 255) */ \
            qv[K] = !(r_ > 0.0f) ? 0.0f : (r_ > 255.0f ? 255.0f : r_);                              \
        } else if (SCALED) qv[K] = __fmul_rn(qv[K], QSCALE);                                        \
        if (LEANS) {                                                                                \
            const int j_ = j0 + K, yo_ = y0 + j_ - 7;                                               \
            float &kqK_ = K == 0 ? kq.x : (K == 1 ? kq.y : (K == 2 ? kq.z : kq.w));                 \
            const bool better_ = j_ >= 7 && yo_ < y1 && lane_out && dg != 0 && qv[K] < kqK_;        \
            kqK_ = better_ ? qv[K] : kqK_;                                                          \
            kd4 = better_ ? ((kd4 & ~(0xffu << (8 * K))) | ((unsigned)dg << (8 * K))) : kd4;        \
            anyb |= better_;                                                                        \
        }                                                                                           \
        if (LEANG && !LEANB) PSM_ISSUE_PB(K & 1, j0 + K + 2)                                        \
        else if (LEANB) {                                                                           \
            PSM_ISSUE_PB(K & 1, j0 + K + 2)                                                         \
            const int j_ = j0 + K, yo_ = y0 + j_ - 7;                                               \
            const long long key_ = pack_key_f32(qv[K], dg);                                         \
            if (j_ >= 7 && yo_ < y1 && lane_out && dg != 0 && qv[K] == qv[K] && key_ < kcur[K & 1]) \
                (void)__hip_atomic_fetch_min(keyp + (size_t)yo_ * W + xb, key_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
            PSM_KEY_LOAD1(K & 1, j0 + K + 2)                                                        \
        } else PSM_ISSUE_PB(K, j0 + K + 4)                                                          \
        a_cur = a_nxt;                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                          \
    }
                PSM_STEP_PB(0) PSM_STEP_PB(1) PSM_STEP_PB(2) PSM_STEP_PB(3)
#undef PSM_STEP_PB
                if constexpr (MODE == 0) {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (lane < bwidth) qbuf[c & 1][k][wb * PC_OUT_B + lane] = qv[k];
                } else if constexpr (LEANB) {
                    // (selection done row by row inside the steps)
                } else if constexpr (MODE == 2) {
                    // DispSel::CVSelect (src/DispSel.cpp:96-104) against the volume's shared key plane: strict '<' / lowest d on
                    // ties = signed minimum of pack_key_f32; d = 0 never a candidate; NaN never wins
#if PSM_KEY_NOATOMIC == 2   // experiment (invalid results): compute-only - the q values are summed into a register, one store per workgroup
                    acc_dbg += (qv[0] + qv[1]) + (qv[2] + qv[3]);
                    if (c == nbB - 1 && lane_out) keyp[(size_t)(y0 + lane % 4) * W + xb] = (long long)__float_as_int(acc_dbg);
#else
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int j_ = j0 + k, yo_ = y0 + j_ - 7;
                        const long long key_ = pack_key_f32(qv[k], dg);
                        if (j_ >= 7 && yo_ < y1 && lane_out && dg != 0 && qv[k] == qv[k] && key_ < kcur[k] && !PSM_KEY_NOATOMIC)
                            (void)__hip_atomic_fetch_min(keyp + (size_t)yo_ * W + xb, key_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    if (c + 1 < nbB) PSM_KEY_LOAD(c + 1)               // keys of the next batch's rows
#endif
                } else if constexpr (LEANS) {
                    // (selection done row by row inside the steps; kq / kd4 hold the updated records of this batch)
                    if (lane < bwidth && (first || anyb)) {
                        const pc_u4 kv = {__float_as_uint(kq.x), __float_as_uint(kq.y), __float_as_uint(kq.z), __float_as_uint(kq.w)};
                        __builtin_amdgcn_raw_buffer_store_b128(kv, rKc, lane * 16, c * (PC_COLS * 16), 0);
                        __builtin_amdgcn_raw_buffer_store_b32(kd4, rKd, lane * 4, c * (PC_COLS * 4), 0);
                    }
                    if (first) { kq = make_float4(__builtin_inff(), __builtin_inff(), __builtin_inff(), __builtin_inff()); kd4 = 0; }
                    else if (c + 1 < nbB) PSM_K_LOAD(c + 1)            // records of the next batch
                } else {
                    // DispSel::CVSelect (src/DispSel.cpp:96-104) over the slices of this chunk: strict '<', d = 0 never a
                    // candidate, NaN never wins.  Rows outside [y0, y1) and halo lanes keep (+inf, 0).
                    float kn[4] = {kq.x, kq.y, kq.z, kq.w};
                    unsigned dn = kd4;
                    bool any = false;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int j_ = j0 + k, yo_ = y0 + j_ - 7;
                        const bool better_ = j_ >= 7 && yo_ < y1 && lane_out && dg != 0 && qv[k] < kn[k];
                        kn[k] = better_ ? qv[k] : kn[k];
                        dn = better_ ? ((dn & ~(0xffu << (8 * k))) | ((unsigned)dg << (8 * k))) : dn;
                        any |= better_;
                    }
#ifndef PSM_PC_MASKED_STORES
#define PSM_PC_MASKED_STORES 1
#endif
                    // (per lane: only lanes with an improved row rewrite their record; 0: the whole wave whenever any lane improved)
                    if (lane < bwidth && (first || (PSM_PC_MASKED_STORES ? any : __builtin_amdgcn_ballot_w64(any) != 0))) {
                        const pc_u4 kv = {__float_as_uint(kn[0]), __float_as_uint(kn[1]), __float_as_uint(kn[2]), __float_as_uint(kn[3])};
                        __builtin_amdgcn_raw_buffer_store_b128(kv, rKc, lane * 16, c * (PC_COLS * 16), 0);
                        __builtin_amdgcn_raw_buffer_store_b32(dn, rKd, lane * 4, c * (PC_COLS * 4), 0);
                    }
                    if (!first && c + 1 < nbB) PSM_K_LOAD(c + 1)       // records of the next batch
                }
            }
            PC_SYNC();
        };
        for (int c = 0; c < nbB; c += PSM_PC_UNROLL2 ? 2 : 1) {
            batch_b(c);
            if (PSM_PC_UNROLL2 && __builtin_expect(c + 1 < nbB, 1)) batch_b(c + 1);
        }
        store_batch(nbB - 1);                          // iteration nbB+2
        PC_SYNC();
#undef PSM_ISSUE_PB
#undef PSM_KEY_LOAD
#undef PSM_KEY_LOAD1
#undef PSM_K_LOAD
    }
    if (MODE == 1) __builtin_amdgcn_s_waitcnt(0);      // the chunk planes of this slice are in the L2 before the next slice reads them
    first = false;
    }   // slices of the chunk
    if constexpr (MODE == 1) {
        if (dynamic && first && !is_a) {
            // this workgroup came too late for any slice: its plane must still read "no candidate"
            const int wb = wave - PC_NA;
            const int bw = wb < PC_NB - 1 ? PC_OUT_B : PC_COLS - (PC_NB - 1) * PC_OUT_B;
            const size_t krec = (size_t)(ch * npairs + pair) * nbmax * PC_COLS + wb * PC_OUT_B;
            float4 *kc = reinterpret_cast<float4 *>(kcost) + krec;
            unsigned *kd = kdisp + krec;
            const float inf = __builtin_inff();
            for (int c = 0; c < nbB && lane < bw; ++c) {
                kc[c * PC_COLS + lane] = make_float4(inf, inf, inf, inf);
                kd[c * PC_COLS + lane] = 0u;
            }
        }
    }
#undef PSM_BOX
#if PSM_PC_TIMING
    if (lane == 0 && MODE == 1) {
        atomicAdd(&g_pc_dbg[wave], q_work);
        atomicAdd(&g_pc_dbg[4 + wave], q_wait);
    }
#endif
}

// chunk planes -> packed WTA key and / or final map per pixel (minimum over the chunks; pack_key_f32 makes the signed
// 64-bit minimum select (min cost, then lowest d) exactly as the sequential loop of src/DispSel.cpp:96-104).
// One thread per record (pair, batch, column) = four rows of one column, as the select kernel wrote them.
__global__ __launch_bounds__(256) void k_chunk_min(const float4 *kcost, const unsigned *kdisp, int nchunks, int npairs,
                                                  int nbmax, int ngroups, int seg_rows, int W, int H, long long *keys,
                                                  uint8_t *map, const float4 *__restrict__ kcost1, const unsigned *__restrict__ kdisp1,
                                                  int ybeg, int yend)
{
    using L = PcLayout<1>;
    if (blockIdx.y == 1) {   // second volume of a two-side launch: its own planes, keys / map one image further
        kcost = kcost1;
        kdisp = kdisp1;
        if (keys) keys += (size_t)W * H;
        if (map) map += (size_t)W * H;
    }
    const size_t nrec = (size_t)npairs * nbmax * L::COLS;          // records per chunk plane
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nrec) return;
    const int col = (int)(idx % L::COLS);
    size_t t = idx / L::COLS;
    const int c = (int)(t % nbmax);
    const int pair = (int)(t / nbmax);
    const int g = pair % ngroups, seg = pair / ngroups;
    const int x = g * L::COLS + col;
    if (x >= W) return;
    const int y0 = ybeg + seg * seg_rows, y1 = min(yend, y0 + seg_rows);
    const int ya = y0 + 4 * c - 7;                                  // output row of the record's first entry
    if (ya + 3 < y0 || ya >= y1) return;
    float4 kc = kcost[idx];
    unsigned kd = kdisp[idx];
    long long best[4] = {pack_key_f32(kc.x, kd & 0xff), pack_key_f32(kc.y, (kd >> 8) & 0xff), pack_key_f32(kc.z, (kd >> 16) & 0xff),
                         pack_key_f32(kc.w, kd >> 24)};
    for (int ch = 1; ch < nchunks; ++ch) {
        kc = kcost[(size_t)ch * nrec + idx];
        kd = kdisp[(size_t)ch * nrec + idx];
        const long long k0 = pack_key_f32(kc.x, kd & 0xff), k1 = pack_key_f32(kc.y, (kd >> 8) & 0xff),
                        k2 = pack_key_f32(kc.z, (kd >> 16) & 0xff), k3 = pack_key_f32(kc.w, kd >> 24);
        best[0] = k0 < best[0] ? k0 : best[0];
        best[1] = k1 < best[1] ? k1 : best[1];
        best[2] = k2 < best[2] ? k2 : best[2];
        best[3] = k3 < best[3] ? k3 : best[3];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int yo = ya + k;
        if (yo < y0 || yo >= y1) continue;
        const size_t o = (size_t)yo * W + x;
        if (keys) keys[o] = best[k];
        if (map) map[o] = (uint8_t)((unsigned long long)best[k] & 0xffull);
    }
}

}  // namespace psm

// debug: read and clear the per-wave cycle counters of k_cvf_pc (all zero unless built with -DPSM_PC_TIMING=1)
extern "C" int psm_debug_pc_cycles(unsigned long long *out8)
{
    for (int i = 0; i < 8; ++i) out8[i] = 0;
#if PSM_PC_TIMING
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(psm::g_pc_dbg), 8 * sizeof(unsigned long long)) != hipSuccess) return 1;
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(psm::g_pc_dbg), z, sizeof z) != hipSuccess) return 1;
#endif
    return 0;
}

namespace psm {

// key plane(s) <- key(+inf, 0): the "no candidate yet" value of MODE 2
__global__ __launch_bounds__(256) void k_fill_keys(long long *__restrict__ keys, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = pack_key_f32(__builtin_inff(), 0);
}

// Segment count k (and, for MODE 1, slices per chunk DC): every segment re-walks 14 halo rows, and the launch runs in
// rounds of resident workgroups - per XCD ceil(pairs/8) (column group, segment) pairs x chunks over 32 CUs x 3
// workgroups.  Cost model, fitted to measurements at 1080p (DC = 1, 2, 4, 8, 16: 4.39, 4.41, 4.46, 4.71, 4.99 ms
// for kernel + reduction): (rounds + 1/2) x rows walked per workgroup - the last round is on average half empty, which
// is what makes long-running workgroups (large DC) expensive - plus two row-steps per chunk plane for the reduction.
PcPlan pc_plan(int W, int H, int Dloc, int seg_rows_opt, int mode)
{
    return pc_plan_cols(W, H, Dloc, seg_rows_opt, mode, mode != 0 ? PcLayout<1>::COLS : PcLayout<0>::COLS);
}

PcPlan pc_plan_cols(int W, int H, int Dloc, int seg_rows_opt, int mode_in, int cols)
{   // mode_in: 0 store; 1 select with chunk planes, 2 the same with both volumes per launch (twice the work items per launch);
    // 3 select with a shared key plane (one slice per workgroup, no reduction afterwards), 4 the same with both volumes per launch;
    // 5 select with chunk planes and dynamic slice distribution (NW resident workgroups per pair), 6 the same with both volumes
    const int mode = (mode_in == 1 || mode_in == 2) ? 1 : 0, sides = (mode_in == 2 || mode_in == 4 || mode_in == 6) ? 2 : 1;
    if (mode_in == 5 || mode_in == 6) {
        // all workgroups resident at once (256 CUs x 3): NW = slots / pairs per pair; a workgroup walks ~Dloc / NW slices of
        // rows / k + 14 rows each.  Pick the k with the smallest makespan.
        PcPlan pl;
        pl.ngroups = (W + cols - 1) / cols;
        const int slots = 768, kmax = H / 64 > 1 ? H / 64 : 1;
        long best = -1;
        int bk = 1, bnw = 1;
        for (int kk = 1; kk <= kmax && kk <= 32; ++kk) {
            if (seg_rows_opt > 0 && kk != (H + seg_rows_opt - 1) / seg_rows_opt) continue;
            const int np = sides * pl.ngroups * kk;
            int nw = slots / np;
            nw = nw < 1 ? 1 : (nw > Dloc ? Dloc : nw);
            const char *e = getenv("PSM_PC_NW");
            if (e && atoi(e) > 0) nw = atoi(e) > Dloc ? Dloc : atoi(e);
            const long rounds = ((long)np * nw + slots - 1) / slots;
            const long c = rounds * ((Dloc + nw - 1) / nw) * ((H + kk - 1) / kk + 14);
            if (best < 0 || c < best) { best = c; bk = kk; bnw = nw; }
        }
        pl.seg_rows = seg_rows_opt > 0 ? seg_rows_opt : (H + bk - 1) / bk;
        if (pl.seg_rows > H) pl.seg_rows = H;
        pl.nsegs = (H + pl.seg_rows - 1) / pl.seg_rows;
        pl.DC = 1;
        pl.NW = bnw;
        pl.nchunks = bnw;
        pl.nbmax = (pl.seg_rows + 7 + 3) / 4;
        pl.rec_per_chunk = (size_t)pl.ngroups * pl.nsegs * pl.nbmax * PcLayout<1>::COLS;
        pl.rec_bytes = 20;
        return pl;
    }
    const int rows = H;
    PcPlan pl;
    pl.ngroups = (W + cols - 1) / cols;
    const int kmax = rows / 64 > 1 ? rows / 64 : 1;
    int dcs[5] = {1, 2, 4, 8, 16};
    int ndc = mode == 1 ? 5 : 1;
    if (mode == 1) {   // tuning / experiments: PSM_PC_DC forces the slices per chunk
        const char *e = getenv("PSM_PC_DC");
        if (e && atoi(e) > 0) { dcs[0] = atoi(e); ndc = 1; }
    }
    auto cost_of = [&](int dc, int kk) -> long {
        const int nch = (Dloc + dc - 1) / dc;
        const long per_xcd = ((long)sides * pl.ngroups * kk * nch + 7) / 8;
        static const int slots_env = getenv("PSM_PC_SLOTS") ? atoi(getenv("PSM_PC_SLOTS")) : 0;
        const long slots = slots_env > 0 ? slots_env : ((mode_in == 3 || mode_in == 4) && PSM_PC_OCC4 ? 128 : 96);   // resident workgroups per XCD: 32 CUs x 3 (key form: x 4)
        static const int tail_env = getenv("PSM_PC_TAIL") ? atoi(getenv("PSM_PC_TAIL")) : -1;
        const long tail2 = tail_env >= 0 ? tail_env : 1;                             // 2 x the tail allowance in rounds
        const long rounds2 = 2 * ((per_xcd + slots - 1) / slots) + ((mode_in == 3 || mode_in == 4) ? tail2 : 1);   // 2 x (rounds + 1/2)
        return rounds2 * dc * ((rows + kk - 1) / kk + 14) / 2 + (mode == 1 ? 2L * sides * nch : 0);
    };
    auto allowed = [&](int dc, int kk) { return (dc == dcs[0] || dc <= Dloc) && (seg_rows_opt <= 0 || kk == (rows + seg_rows_opt - 1) / seg_rows_opt); };
    long best = -1;
    for (int di = 0; di < ndc; ++di)
        for (int kk = 1; kk <= kmax && kk <= 32; ++kk)
            if (allowed(dcs[di], kk)) { const long c = cost_of(dcs[di], kk); if (best < 0 || c < best) best = c; }
    int bk = seg_rows_opt > 0 ? (rows + seg_rows_opt - 1) / seg_rows_opt : 1, bdc = dcs[0];
    for (int di = 0; di < ndc && best >= 0; ++di)
        for (int kk = 1; kk <= kmax && kk <= 32; ++kk)
            if (allowed(dcs[di], kk) && cost_of(dcs[di], kk) == best) { bk = kk; bdc = dcs[di]; di = ndc; break; }
    pl.seg_rows = seg_rows_opt > 0 ? seg_rows_opt : (rows + bk - 1) / bk;
    if (pl.seg_rows > rows) pl.seg_rows = rows;
    pl.nsegs = (rows + pl.seg_rows - 1) / pl.seg_rows;
    pl.DC = bdc;
    pl.NW = 0;
    pl.nchunks = (Dloc + bdc - 1) / bdc;
    pl.nbmax = (pl.seg_rows + 7 + 3) / 4;                            // consumer batches of a full segment
    pl.rec_per_chunk = (size_t)pl.ngroups * pl.nsegs * pl.nbmax * PcLayout<1>::COLS;
    pl.rec_bytes = 20;                                               // 16 bytes of costs + 4 bytes of disparities
    return pl;
}

void launch_cvf_fused(hipStream_t s, March m, const float *vin, float *vout, Guidance gd, int W, int H, int Dloc,
                      int ybeg, int yend, const float4 *g1_other, int d_begin, int cvc_mode)
{
    if (yend <= ybeg) return;
    const int rows = yend - ybeg;
    const PcPlan pl = pc_plan(W, rows, Dloc, m.seg_rows, 0);
    const int nblocks = 8 * ((pl.ngroups * pl.nsegs * Dloc + 7) / 8);
    const dim3 blk(64 * (PcLayout<0>::NA + PcLayout<0>::NB));
#define PSM_LAUNCH_PC(V4, CV)                                                                                              \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cvf_pc<V4, CV, 0>), dim3(nblocks), blk, 0, s, vin, vout, (const float4 *)gd.g1,   \
                       (const float4 *)gd.g2, (const float4 *)gd.g3, (const float2 *)gd.g4, W, H, Dloc, pl.ngroups, pl.nsegs, \
                       pl.seg_rows, ybeg, yend, g1_other, d_begin, 1, (float *)nullptr, (unsigned *)nullptr, 0, PcSide{}, PcDyn{nullptr, 0, 0, 1, 1})
    const bool v4 = (W & 3) == 0;
    if (cvc_mode == 1) { if (v4) PSM_LAUNCH_PC(true, 1); else PSM_LAUNCH_PC(false, 1); }
    else if (cvc_mode == 2) { if (v4) PSM_LAUNCH_PC(true, 2); else PSM_LAUNCH_PC(false, 2); }
    else { if (v4) PSM_LAUNCH_PC(true, 0); else PSM_LAUNCH_PC(false, 0); }
#undef PSM_LAUNCH_PC
}

void launch_cvf_select(hipStream_t s, March m, const float *vin, Guidance gd, int W, int H, int Dloc, const float4 *g1_other,
                       int d_begin, int cvc_mode, void *scratch, int *cnt, const uint8_t *p4_own, const uint8_t *p4_other, int sel, int step)
{   // p4_own != NULL (cvc_mode 1 / 2 only): 8-bit char mode; cnt != NULL: dynamic slice distribution (npairs ints, zeroed here)
    // Dloc = number of slices of this launch, (sel, step) = which ones (PcDyn)
    const PcPlan pl = pc_plan(W, m.rows(H), Dloc, m.seg_rows, cnt ? 5 : 1);
    const PcDyn dyn = {cnt, pl.NW, sel, step, 1};
    if (cnt) (void)hipMemsetAsync(cnt, 0, sizeof(int) * pl.ngroups * pl.nsegs, s);
    float *kcost = (float *)scratch;                                           // nchunks * rec_per_chunk float4
    unsigned *kdisp = (unsigned *)(kcost + 4 * pl.rec_per_chunk * pl.nchunks);  // nchunks * rec_per_chunk uchar4
    const int nblocks = 8 * ((pl.ngroups * pl.nsegs * pl.nchunks + 7) / 8);
    const dim3 blk(64 * (PcLayout<1>::NA + PcLayout<1>::NB));
#define PSM_LAUNCH_PC(CV)                                                                                                   \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cvf_pc<false, CV, 1>), dim3(nblocks), blk, 0, s, vin, (float *)nullptr, (const float4 *)gd.g1, \
                       (const float4 *)gd.g2, (const float4 *)gd.g3, (const float2 *)gd.g4, W, H, Dloc, pl.ngroups, pl.nsegs, \
                       pl.seg_rows, m.y0(H), m.y1(H), g1_other, d_begin, pl.DC, kcost, kdisp, pl.nbmax, PcSide{}, dyn)
    if (p4_own && cvc_mode != 0) {
#define PSM_LAUNCH_PC8(CV)                                                                                                  \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cvf_pc<false, CV, 1, true>), dim3(nblocks), blk, 0, s, (const float *)p4_own, (float *)const_cast<uint8_t *>(p4_other), \
                       (const float4 *)gd.g1, (const float4 *)gd.g2, (const float4 *)gd.g3, (const float2 *)gd.g4, W, H, Dloc, pl.ngroups, pl.nsegs, \
                       pl.seg_rows, m.y0(H), m.y1(H), g1_other, d_begin, pl.DC, kcost, kdisp, pl.nbmax, PcSide{}, dyn)
        if (cvc_mode == 1) PSM_LAUNCH_PC8(1); else PSM_LAUNCH_PC8(2);
#undef PSM_LAUNCH_PC8
    } else if (cvc_mode == 1) PSM_LAUNCH_PC(1); else if (cvc_mode == 2) PSM_LAUNCH_PC(2); else PSM_LAUNCH_PC(0);
#undef PSM_LAUNCH_PC
}

// Select mode with a shared key plane (MODE 2): keys[H*W] of this volume receives the packed minima over the local slices.
void launch_cvf_select_keys(hipStream_t s, March m, const float *vin, Guidance gd, int W, int H, int Dloc, const float4 *g1_other,
                            int d_begin, int cvc_mode, long long *keys, const uint8_t *p4_own, const uint8_t *p4_other, int init, int sel, int step)
{   // init: start from key(+inf, 0); otherwise continue from what `keys` holds (second phase of the two-phase selection)
    const PcPlan pl = pc_plan(W, m.rows(H), Dloc, m.seg_rows, 3);
    const size_t HW = (size_t)W * H;
    if (init) hipLaunchKernelGGL(k_fill_keys, dim3((unsigned)((HW + 255) / 256)), dim3(256), 0, s, keys, HW);
    const int nblocks = 8 * ((pl.ngroups * pl.nsegs * Dloc + 7) / 8);
    const dim3 blk(64 * (PcLayout<2>::NA + PcLayout<2>::NB));
#define PSM_LAUNCH_K(CV, U8V, A0, A1)                                                                                       \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cvf_pc<false, CV, 2, U8V>), dim3(nblocks), blk, 0, s, A0, A1, (const float4 *)gd.g1, \
                       (const float4 *)gd.g2, (const float4 *)gd.g3, (const float2 *)gd.g4, W, H, Dloc, pl.ngroups, pl.nsegs, \
                       pl.seg_rows, m.y0(H), m.y1(H), g1_other, d_begin, 1, (float *)keys, (unsigned *)nullptr, 0, PcSide{}, PcDyn{nullptr, 0, sel, step, 1})
    if (p4_own && cvc_mode != 0) {
        if (cvc_mode == 1) PSM_LAUNCH_K(1, true, (const float *)p4_own, (float *)const_cast<uint8_t *>(p4_other));
        else PSM_LAUNCH_K(2, true, (const float *)p4_own, (float *)const_cast<uint8_t *>(p4_other));
    } else if (cvc_mode == 1) PSM_LAUNCH_K(1, false, (const float *)nullptr, (float *)nullptr);
    else if (cvc_mode == 2) PSM_LAUNCH_K(2, false, (const float *)nullptr, (float *)nullptr);
    else PSM_LAUNCH_K(0, false, vin, (float *)nullptr);
#undef PSM_LAUNCH_K
}

// ... both volumes in one launch: keys[2][H][W]
void launch_cvf_select_keys2(hipStream_t s, March m, const Guidance *g, int W, int H, int Dloc, int d_begin, long long *keys,
                             const uint8_t *const *p4, int init, int sel, int step, int unit)
{
    const PcPlan pl = pc_plan(W, m.rows(H), Dloc, m.seg_rows, 4);
    const size_t HW = (size_t)W * H;
    if (init) hipLaunchKernelGGL(k_fill_keys, dim3((unsigned)((2 * HW + 255) / 256)), dim3(256), 0, s, keys, 2 * HW);
    const int nblocks = 8 * ((pl.ngroups * pl.nsegs * Dloc + 7) / 8);
    const dim3 blk(64 * (PcLayout<2>::NA + PcLayout<2>::NB));
    const PcSide s1 = {(const float4 *)g[1].g1, (const float4 *)g[1].g2, (const float4 *)g[1].g3, (const float2 *)g[1].g4, (const float4 *)g[0].g1,
                       (float *)(keys + HW), nullptr};
    if (p4)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cvf_pc<false, 3, 2, true>), dim3(nblocks, 2), blk, 0, s, (const float *)p4[0], (float *)const_cast<uint8_t *>(p4[1]),
                           (const float4 *)g[0].g1, (const float4 *)g[0].g2, (const float4 *)g[0].g3, (const float2 *)g[0].g4, W, H, Dloc, pl.ngroups,
                           pl.nsegs, pl.seg_rows, m.y0(H), m.y1(H), (const float4 *)g[1].g1, d_begin, 1, (float *)keys, (unsigned *)nullptr, 0, s1, PcDyn{nullptr, 0, sel, step, unit});
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cvf_pc<false, 3, 2, false>), dim3(nblocks, 2), blk, 0, s, (const float *)nullptr, (float *)nullptr,
                           (const float4 *)g[0].g1, (const float4 *)g[0].g2, (const float4 *)g[0].g3, (const float2 *)g[0].g4, W, H, Dloc, pl.ngroups,
                           pl.nsegs, pl.seg_rows, m.y0(H), m.y1(H), (const float4 *)g[1].g1, d_begin, 1, (float *)keys, (unsigned *)nullptr, 0, s1, PcDyn{nullptr, 0, sel, step, unit});
}

void launch_chunk_min(hipStream_t s, March m, int W, int H, int Dloc, void *scratch, long long *keys, uint8_t *map, int dynamic)
{
    const PcPlan pl = pc_plan(W, m.rows(H), Dloc, m.seg_rows, dynamic ? 5 : 1);
    const float *kcost = (const float *)scratch;
    const unsigned *kdisp = (const unsigned *)(kcost + 4 * pl.rec_per_chunk * pl.nchunks);
    hipLaunchKernelGGL(k_chunk_min, dim3((unsigned)((pl.rec_per_chunk + 255) / 256)), dim3(256), 0, s, (const float4 *)kcost, (const unsigned *)kdisp,
                       pl.nchunks, pl.ngroups * pl.nsegs, pl.nbmax, pl.ngroups, pl.seg_rows, W, H, keys, map, (const float4 *)nullptr,
                       (const unsigned *)nullptr, m.y0(H), m.y1(H));
}

// Both volumes in one launch each (costs built on the fly): left volume = (g[0], other g[1].g1), right = (g[1], other g[0].g1);
// scratch: 2 x pc_plan(...).scratch_bytes(); keys / map: [2][H][W].
void launch_cvf_select2(hipStream_t s, March m, const Guidance *g, int W, int H, int Dloc, int d_begin, void *scratch, int *cnt,
                        const uint8_t *const *p4, int sel, int step)
{   // p4 != NULL: 8-bit char mode, p4[0] / p4[1] = byte planes {c0,c1,c2,grad} of the left / right image
    // cnt != NULL: dynamic slice distribution (2 * npairs ints, zeroed here)
    const PcPlan pl = pc_plan(W, m.rows(H), Dloc, m.seg_rows, cnt ? 6 : 2);
    const PcDyn dyn = {cnt, pl.NW, sel, step, 1};
    if (cnt) (void)hipMemsetAsync(cnt, 0, sizeof(int) * 2 * pl.ngroups * pl.nsegs, s);
    float *kcost0 = (float *)scratch;
    unsigned *kdisp0 = (unsigned *)(kcost0 + 4 * pl.rec_per_chunk * pl.nchunks);
    float *kcost1 = (float *)((char *)scratch + pl.scratch_bytes());
    unsigned *kdisp1 = (unsigned *)(kcost1 + 4 * pl.rec_per_chunk * pl.nchunks);
    const int nblocks = 8 * ((pl.ngroups * pl.nsegs * pl.nchunks + 7) / 8);
    const dim3 blk(64 * (PcLayout<1>::NA + PcLayout<1>::NB));
    const PcSide s1 = {(const float4 *)g[1].g1, (const float4 *)g[1].g2, (const float4 *)g[1].g3, (const float2 *)g[1].g4, (const float4 *)g[0].g1, kcost1, kdisp1};
    if (p4)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cvf_pc<false, 3, 1, true>), dim3(nblocks, 2), blk, 0, s, (const float *)p4[0], (float *)const_cast<uint8_t *>(p4[1]),
                           (const float4 *)g[0].g1, (const float4 *)g[0].g2, (const float4 *)g[0].g3, (const float2 *)g[0].g4, W, H, Dloc, pl.ngroups,
                           pl.nsegs, pl.seg_rows, m.y0(H), m.y1(H), (const float4 *)g[1].g1, d_begin, pl.DC, kcost0, kdisp0, pl.nbmax, s1, dyn);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cvf_pc<false, 3, 1>), dim3(nblocks, 2), blk, 0, s, (const float *)nullptr, (float *)nullptr, (const float4 *)g[0].g1,
                           (const float4 *)g[0].g2, (const float4 *)g[0].g3, (const float2 *)g[0].g4, W, H, Dloc, pl.ngroups, pl.nsegs, pl.seg_rows, m.y0(H), m.y1(H),
                           (const float4 *)g[1].g1, d_begin, pl.DC, kcost0, kdisp0, pl.nbmax, s1, dyn);
}

void launch_chunk_min2sides(hipStream_t s, March m, int W, int H, int Dloc, void *scratch, long long *keys, uint8_t *map, int dynamic)
{
    const PcPlan pl = pc_plan(W, m.rows(H), Dloc, m.seg_rows, dynamic ? 6 : 2);
    const float *kcost0 = (const float *)scratch;
    const unsigned *kdisp0 = (const unsigned *)(kcost0 + 4 * pl.rec_per_chunk * pl.nchunks);
    const float *kcost1 = (const float *)((const char *)scratch + pl.scratch_bytes());
    const unsigned *kdisp1 = (const unsigned *)(kcost1 + 4 * pl.rec_per_chunk * pl.nchunks);
    hipLaunchKernelGGL(k_chunk_min, dim3((unsigned)((pl.rec_per_chunk + 255) / 256), 2), dim3(256), 0, s, (const float4 *)kcost0, kdisp0,
                       pl.nchunks, pl.ngroups * pl.nsegs, pl.nbmax, pl.ngroups, pl.seg_rows, W, H, keys, map, (const float4 *)kcost1, kdisp1, m.y0(H), m.y1(H));
}

}  // namespace psm
