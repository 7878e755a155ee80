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
#include "../../include/primesm_hip.h"

#include <mutex>

#include "psm_pc_debug.inc"   // PSM_PC_TIMING per-workgroup traces and the PSM_EXPERIMENTS knobs: all compiled out of the product

namespace psm {

typedef float f4v __attribute__((ext_vector_type(4)));

// Occupancy of the key form (MODE 2, 4/5 of the slices of a 256-slice volume): four workgroups per CU instead of three.
// The kernel is latency-sensitive at 3 waves per SIMD (barrier waits, LDS round trips); 128 VGPRs need a shorter load
// look-ahead - consumer: G1 / keys two rows ahead instead of a batch, selection row by row inside the steps; producer: guidance
// planes issued at the start of their own step, partner pixels right after the cost is formed - and cost 4 spilled registers
// per producer batch.  Measured at 1080p x 256: key phase 6.36 -> 5.86 ms (the same cap without the diet: 65 spills, 12.7 ms
// per frame).  The plane form (MODE 1) stays at three workgroups per CU: its own diet got it from 159 to 139 registers / 9
// spills under the cap, but there the shorter look-ahead costs what the fourth workgroup gains.
constexpr int PC_RING = 4;   // batches of four model rows kept in LDS

// NARROW (select forms only): one producer + one consumer wave, 57 model columns -> 50 output columns.  Per wave a little less
// useful (25 output columns against 26.75) but half the granularity: an image whose last 107-column group would be mostly empty
// (450 columns = 4 groups + 22 columns) is cut into 9 groups of 50 instead - 18 waves' worth of columns instead of 20.  pc_plan picks
// the layout that needs fewer waves per row of the image (ties: the wide one).
template <int MODE, bool NARROW = false> struct PcLayout;
template <> struct PcLayout<0, false> { static constexpr int NA = 2, NB = 2, OUT_A = 52, OUT_B = 48, COLS = 96; };
template <> struct PcLayout<1, false> { static constexpr int NA = 2, NB = 2, OUT_A = 57, OUT_B = 54, COLS = 107; };
template <> struct PcLayout<2, false> : PcLayout<1, false> {};
template <> struct PcLayout<1, true> { static constexpr int NA = 1, NB = 1, OUT_A = 57, OUT_B = 50, COLS = 50; };
template <> struct PcLayout<2, true> : PcLayout<1, true> {};
constexpr int PC_K_LD_AUX = 0;     // cache policy of MODE 1's record loads.  A record is only ever read by the lane that wrote it
                                   // (one slice earlier), and a thread always observes its own stores: plain cached loads are
                                   // coherent here.  (16 = sc1 bypasses the L2 as well: measured 25 % slower kernel.)
// cache policy of MODE 2's key loads.  A stale key only costs a redundant atomic (the atomicMin decides), so the loads need not be
// device-coherent: 1 = sc0 - served by the XCD's L2, never by this CU's L1.  Round 6, same box, 16 (sc1: always from the memory
// side, rounds 2-5) / 0 / 1: 1280 x 720 x 128 1.785 / 1.743 / 1.745 ms, 1920 x 1080 x 256 6.751 / 6.758 / 6.743, 3840 x 2160 x 256
// 27.38 / 27.37 / 27.25 (profiles/r06/exp_key_load_policy.txt) - and the key plane's share of the fabric traffic at 4K goes away.
constexpr int PC_KEY_LD_AUX = 1;

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

// Which local slices a launch covers (Dloc = their number): sel 0 all (index i = local slice i); sel 1 every step-th slice
// (i -> i * step); sel 2 the others (i -> (i / (step-1)) * step + i % (step-1) + 1).  Two-phase selection: a first launch
// reduces every step-th slice to the key plane, a MODE 2 launch then runs the rest against that plane - its consumer lanes
// find a tight bound there and issue an atomic only a few times per pixel.
struct PcSel {
    int sel, step;
    int nxcd;      // XCDs of the device (blocks are dispatched round-robin over them)
    int spread;    // key form: dispatch the slices of a pair in `spread` interleaved passes (0 / 1: ascending)
    int n;         // ... over n slices
    unsigned long long rec_total;   // BATCH launches: float4 cost records per side in a pair's scratch (the disparities follow them)
    int dstep;     // global disparity of local slice i: d_begin + i * dstep (1: a contiguous range; psm_create_shard_strided: the rank count)
};
__device__ __forceinline__ int pc_slice(const PcSel &o, int i)
{
    // Key form: the workgroups of one (column group, segment) pair that run at the same time would be CONSECUTIVE slices - the
    // ones most likely to tie or nearly tie for a pixel's minimum, each deciding on a snapshot of the key that predates the
    // others' updates (redundant atomics; with quantised 8-bit costs exact ties are the rule).  Dispatch position i -> slice
    // index in `spread` passes (0, s, 2s, .. | 1, s+1, .. | ..): concurrent slices are s apart, a slice's lower neighbour has
    // long finished when it starts - and a later tie loses against the lower d without an atomic.
    if (o.spread > 1) {
        int r = 0, base = 0;
        for (; r < o.spread; ++r) {
            const int cnt = (o.n - r + o.spread - 1) / o.spread;     // slices with index % spread == r
            if (i < base + cnt) break;
            base += cnt;
        }
        i = (i - base) * o.spread + r;
    }
    return o.sel == 1 ? i * o.step : (o.sel == 2 ? (i / (o.step - 1)) * o.step + i % (o.step - 1) + 1 : i);
}

// CVC = 0: the cost slice is read from `vin`.  CVC = 1 (left volume) / 2 (right volume): the cost volume is never
// materialised - the producer waves evaluate myCostGrd (src/CVC.cpp:18-39) for their input column on the fly from the
// two g1 planes (`G1` = this side's image, `Gother` = the other one), exactly as k_cvc does.
// U8 (8-bit char mode, select mode with costs on the fly only): the producer waves build the 8-bit matching cost of
// assets/cvc.cl:279-301 (oracle: cost_u8) from the {c0,c1,c2,grad} byte planes handed over in `vin` (this side) and `vout`
// (other side) and filter cost * (1/255.0f); the consumer waves re-quantise q8 = sat_u8(rintf(q * 255)) before the
// strict-'<' selection (assets/dispsel.cl:41-62 with the initial minimum above 255) - the build-defined 8-bit contract
// of oracle/psm_oracle.h, in one pass and without an 8-bit volume in memory.
// BATCH (psm_compute_batch): blockIdx.z = stereo pair; every plane pointer of the pair comes from the device table `batch`
// (uniform: scalar loads), so B Middlebury-size pairs fill the chip for many rounds of workgroups instead of 1.7.
// VAR - the two opt-in arithmetic variants of float mode (0: the canon, bit-exact against the oracle):
//   1 (PSM_FLAG_F32_TOL): level 1 of the horizontal trees of both roles in fp32 (psm_dev.h: hsum8<true>) - within the 1e-4
//     BASELINE.json states for float mode, not the oracle's bits;
//   2 (PSM_FLAG_FMA_SOLVE): the 3x3 solve as an FMA target compiles it (psm_dev.h: solve_ab<true>; minors / DET from
//     k_guide_march in their fused forms) - the oracle's reading PSMO_VAR_FMA_SOLVE, bit for bit.
template <bool VEC4, int CVC, int MODE, bool U8 = false, bool BATCH = false, int VAR = 0, bool NARROW = false>
__global__ __launch_bounds__(64 * (PcLayout<MODE, NARROW>::NA + PcLayout<MODE, NARROW>::NB))
__attribute__((amdgpu_waves_per_eu(MODE == 2 ? 4 : 1, MODE == 2 ? 4 : 8)))   // the key form capped at 128 VGPRs = four workgroups per CU
void k_cvf_pc(
    const float *__restrict__ vin, float *__restrict__ vout, const float4 *__restrict__ G1a, const float4 *__restrict__ G2a,
    const float4 *__restrict__ G3a, const float2 *__restrict__ G4a, int W, int H, int Dloc, int ngroups, int nsegs, int seg_rows,
    int ybeg, int yend, const float4 *__restrict__ Gothera, int d_begin, int DC, float *__restrict__ kcosta, unsigned *__restrict__ kdispa, int nbmax,
    PcSide side1, PcSel dyn, unsigned long long *__restrict__ ts, const PcPair *__restrict__ batch)
{
    constexpr bool TOL = VAR == 1, FMA = VAR == 2;
    static_assert(VAR >= 0 && VAR <= 2 && !(TOL && (U8 || MODE == 0)) && !(FMA && (U8 || BATCH)), "tolerance form: float select forms; FMA solve: float mode, single pair");
    if constexpr (BATCH) {
        static_assert(CVC == 3 && MODE != 0, "batched launches: both volumes per launch, select forms");
        const PcPair pp = batch[blockIdx.z];
        G1a = pp.g[0].g1; G2a = pp.g[0].g2; G3a = pp.g[0].g3; G4a = pp.g[0].g4; Gothera = pp.g[1].g1;
        side1.G1 = pp.g[1].g1; side1.G2 = pp.g[1].g2; side1.G3 = pp.g[1].g3; side1.G4 = pp.g[1].g4; side1.Gother = pp.g[0].g1;
        if (U8) { vin = (const float *)pp.p4[0]; vout = (float *)pp.p4[1]; }
        if (MODE == 1) {     // minima planes: [side][costs | disparities] in the pair's scratch
            kcosta = (float *)pp.scratch;
            kdispa = (unsigned *)(kcosta + 4 * dyn.rec_total);
            side1.kcost = (float *)((char *)pp.scratch + dyn.rec_total * 20);
            side1.kdisp = (unsigned *)(side1.kcost + 4 * dyn.rec_total);
        } else {             // key planes
            kcosta = (float *)pp.keys;
            side1.kcost = (float *)(pp.keys + (size_t)W * H);
        }
    }
    const bool right = CVC == 2 || (CVC == 3 && blockIdx.y == 1);      // buildCV_right arithmetic (uniform)
    const bool s1 = CVC == 3 && blockIdx.y == 1;
    const float4 *const G1 = s1 ? side1.G1 : G1a, *const G2 = s1 ? side1.G2 : G2a, *const G3 = s1 ? side1.G3 : G3a;
    const float2 *const G4 = s1 ? side1.G4 : G4a;
    const float4 *const Gother = s1 ? side1.Gother : Gothera;
    float *const kcost = s1 ? side1.kcost : kcosta;
    unsigned *const kdisp = s1 ? side1.kdisp : kdispa;
    static_assert(!NARROW || (MODE != 0 && CVC == 3), "the narrow layout exists for the two-volume select forms");
    using L = PcLayout<MODE, NARROW>;
    constexpr int PC_NA = L::NA, PC_NB = L::NB, PC_OUT_A = L::OUT_A, PC_OUT_B = L::OUT_B, PC_COLS = L::COLS;
    constexpr int PC_MCOLS = PC_NA * PC_OUT_A;   // model columns per workgroup (>= PC_COLS + 7)
    // Select forms: the x 1/64 of the two box filters is not applied per window sum (a v_ldexp_f64 per channel and step).  The
    // guided filter is homogeneous of degree one in (box(p), box(I p)) and in (box(a), box(b)), and a power-of-two scale commutes
    // with every fp32 rounding as long as nothing under- or overflows: the producers hand over 64 x (a0, a1, a2, b), the consumers
    // form 4096 x q and one fp32 multiply by 2^-12 (exact) restores q - bit-identical to the per-sum scaling while no intermediate
    // of a voxel is subnormal or beyond 2^115 (costs are O(1); tests/test_gpu_parity.py::test_scaled_sums_domain).  The storing
    // form (MODE 0) keeps the per-sum scaling, i.e. the oracle's arithmetic op for op.
    constexpr bool SCALED = MODE != 0;
    constexpr float QSCALE = SCALED ? 0x1p-12f : 1.0f;
#define PSM_BOX(N) (SCALED ? (float)(N) : box_out(N))
    static_assert(PC_MCOLS >= PC_COLS + 7 && PC_OUT_A <= 57 && PC_OUT_B <= 57 && (MODE != 0 || (PC_COLS % 4) == 0), "bad producer/consumer layout");
    // Model rows live in a ring of PC_RING batches of four rows; consumers run two batches behind the
    // producers, so every model row the second box filter can ask for - including the REFLECT_101 rows at
    // the top and bottom of the image, which are earlier/later rows of the same ring - is still present.
    __shared__ __attribute__((aligned(16))) float4 ring[PC_RING][4][PC_MCOLS];
    __shared__ __attribute__((aligned(16))) float qbuf[MODE == 0 ? 2 : 1][MODE == 0 ? 4 : 1][MODE == 0 ? PC_COLS : 4];   // MODE 0: output rows, two batches
    // Workgroup -> (column group, segment, slice chunk).  Blocks are observed to go round-robin over the XCDs
    // (block b -> XCD b % nxcd): every XCD owns a contiguous range of (group, segment) pairs and walks the
    // chunks of one pair back to back, so the guidance rows its resident workgroups are reading (few
    // pairs, neighbouring rows, many slices) fit its 4 MB L2 instead of coming from the MALL.  Speed only.
    const int nchunks = (Dloc + DC - 1) / DC;
    int id = blockIdx.x;
    const int npairs = ngroups * nsegs;               // (column group, segment) pairs
    // work items (pair, chunk), pair-major; XCD x takes the x-th share of them: a contiguous range of pairs whose chunks
    // run back to back, and equal work per XCD whatever the pair count
    const int nitems = npairs * nchunks;
    const int ipx = (nitems + dyn.nxcd - 1) / dyn.nxcd;
    const int xcd = id % dyn.nxcd, jj = id / dyn.nxcd;
    const int item = xcd * ipx + jj;
    if (jj >= ipx || item >= nitems) return;
    if (ts != nullptr && threadIdx.x == 0) {          // PSM_OPT_PROFILE 2: when did the first workgroup of this launch start
        (void)__hip_atomic_fetch_min(ts, (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (blockIdx.x == 0 && blockIdx.y == 0) ts[2 * PC_TS_SLOTS] = (unsigned long long)MODE;
    }
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

    PC_TRACE_BEGIN()
    // Slices of this chunk (MODE 0 / 2 always run with DC == 1: one slice per workgroup).  Round 6: consecutive slices of a chunk
    // OVERLAP - the producers start slice s + 1 (its eight warm-up steps and first two batches) while the consumers finish the last
    // two batches of slice s, instead of idling through them and making the consumers idle through the next warm-up.  Barriers per
    // slice: nbB (was nbB + 3); the consumers' two leading barriers are executed once per chunk, the producers' three trailing ones
    // after the last slice.  The ring keeps counting batches across slices (rb): the producers are never more than two batches ahead
    // of what the consumers still read, within a slice as before and across the seam (slots rb+nbA, rb+nbA+1 are written while
    // rb+nbA-2, rb+nbA-1 are read).
    const int nds = MODE == 1 ? min(DC, Dloc - ch * DC) : 1;
    bool first = true;                                // no slice processed yet: the plane holds nothing
    int rb = 0;                                       // ring slot of this slice's model batch 0
    for (int ds = 0; ds < nds; ++ds) {                // ascending d
    const int d = ch * DC + ds;
    const bool last_slice = ds == nds - 1;
    if (is_a) {
        // ---------------- producer: stage A ----------------
        // step s reads input row mstart-5+s; from step 8 on it yields model row mstart+(s-8)
        const int xa0 = xm0 + wave * PC_OUT_A;        // first model column of this wave
        const int ci = r101c(xa0 - 4 + lane, W);      // input column of this lane
        const int xa = xa0 + lane;                    // model column of this lane
        const int xac = xa < 0 ? 0 : (xa > W - 1 ? W - 1 : xa);
        const bool mvalid = lane < PC_OUT_A;
        const float *vd = vin + (size_t)pc_slice(dyn, d) * HW;
        const int dg = d_begin + dyn.dstep * pc_slice(dyn, d);    // global disparity of this slice
        // buildCV_left: partner x-d while x >= d; buildCV_right: partner x+d while x < W-d (src/CVC.cpp:135-146,165-176)
        const bool inb = right ? (ci < W - dg) : (ci >= dg);
        const int cpart = right ? min(ci + dg, W - 1) : max(ci - dg, 0);
        const bool any_border = CVC != 0 && __builtin_amdgcn_ballot_w64(!inb) != 0;
        VTree t0 = {}, t1 = {}, t2 = {}, t3 = {};
        float pin[2];
        float4 oth[2], gin[2], o2[2], o3[2];
        float2 o4[2];
        constexpr bool LEANA = MODE == 2;             // short look-ahead (128-VGPR diet of the key form)
        // raw buffer loads: descriptors and row offsets in scalar registers, one constant 32-bit byte offset per lane
        // (psm_create keeps W*H < 2^27, so every byte offset into a 16-byte plane fits 31 bits)
        const __amdgpu_buffer_rsrc_t rG1 = pc_rsrc(G1, (unsigned)HW * 16u), rG2 = pc_rsrc(G2, (unsigned)HW * 16u);
        const __amdgpu_buffer_rsrc_t rG3 = pc_rsrc(G3, (unsigned)HW * 16u), rG4 = pc_rsrc(G4, (unsigned)HW * 8u);
        const __amdgpu_buffer_rsrc_t rGo = pc_rsrc(CVC == 0 ? G1 : Gother, (unsigned)HW * 16u);
        const __amdgpu_buffer_rsrc_t rV = pc_rsrc(CVC == 0 ? (const void *)vd : (const void *)G1, (unsigned)HW * 4u);
        // U8: byte planes {c0,c1,c2,grad} of this side's / the other side's image (swapped for the right volume of a two-side launch)
        const __amdgpu_buffer_rsrc_t rPo = pc_rsrc(U8 ? (s1 ? (const void *)vin : (const void *)vout) : (const void *)G1, (unsigned)HW * 4u);
        unsigned po[2];
        const int vci = ci * 16, vcp = cpart * 16, vxa = xac * 16;
        // The loads of a step come in two groups with different life times: the PIXELS (this column's g1 entry and its partner's /
        // the stored cost / the partner's bytes) are consumed at the START of their step (cost, three products, conversions), the
        // d-invariant GUIDANCE (means, 1/DET, adjugate: 40 B) at its END (the solve).  Guidance is issued at the start of the step
        // before (two step lengths in flight).  The pixels were issued at the same point - one step length in flight - and are what
        // a producer wave waits for when a line comes from the MALL / HBM instead of the L2 (short launches: several (column group,
        // segment) pairs share an XCD and every line is a compulsory miss for the first of the workgroups that march in step).
        // DEEP: a slot's pixels are dead once the products are formed, so the loads of step S + 2 are issued right there, into the
        // slot step S just read - nearly two step lengths in flight at no register cost.
        constexpr bool DEEP = !LEANA;
#define PSM_ISSUE_PIX(SLOT, STEP)                                                       \
    {                                                                                   \
        const int row_ = r101c(mstart - 5 + (STEP), H) * W;                             \
        if (CVC == 0) pin[SLOT] = pc_load1(rV, vci >> 2, row_ * 4);                     \
        else if (U8) {   /* (this pixel's own bytes ride in gin.w: k_prep_u8) */         \
            po[SLOT] = __builtin_amdgcn_raw_buffer_load_b32(rPo, vcp >> 2, row_ * 4, 0); \
        } else if (!LEANA) oth[SLOT] = pc_load4(rGo, vcp, row_ * 16);                   \
        gin[SLOT] = pc_load4(rG1, vci, row_ * 16);                                      \
    }
#define PSM_ISSUE_GUI(SLOT, STEP)                                                       \
    {                                                                                   \
        int ya_ = mstart - 8 + (STEP);                                                  \
        ya_ = ya_ < 0 ? 0 : (ya_ > H - 1 ? H - 1 : ya_);                                \
        const int oa_ = ya_ * W;                                                        \
        o2[SLOT] = pc_load4(rG2, vxa, oa_ * 16);                                        \
        o3[SLOT] = pc_load4(rG3, vxa, oa_ * 16);                                        \
        o4[SLOT] = pc_load2(rG4, vxa >> 1, oa_ * 8);                                    \
    }
#define PSM_ISSUE_PA(SLOT, STEP)                                                        \
    {                                                                                   \
        if (!DEEP) PSM_ISSUE_PIX(SLOT, STEP)                                            \
        if (!LEANA) PSM_ISSUE_GUI(SLOT, STEP)                                           \
    }
#define PSM_ISSUE_PA2(STEP)   /* key form: the guidance planes of step STEP, issued when the step starts */ \
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
            const unsigned pu_ = __float_as_uint(gin[K & 1].w);       /* this pixel's {c0,c1,c2,grad} bytes */ \
            /* |gradient difference| and colour sum from two SADs (all four bytes, then the gradient byte alone); clr / 3 as a   \
               24-bit multiply (clr <= 765: floor(clr * 21846 / 65536) == clr / 3); the truncating cast to uchar of a value in    \
               [0, 256) as v_trunc_f32 - integer arithmetic and exact conversions only, same bits as oracle cost_u8 */           \
            const unsigned grd_ = __builtin_amdgcn_sad_u8(pu_ & 0xff000000u, b_ & 0xff000000u, 0u); \
            const unsigned clr_ = __builtin_amdgcn_sad_u8(pu_, b_, 0u) - grd_;                      \
            const float f_ = __fadd_rn(__fmul_rn(0.9f, (float)(__umul24(clr_, 21846u) >> 16)), __fmul_rn(__fsub_rn(1.0f, 0.9f), (float)grd_)); \
            p = __fmul_rn(__builtin_truncf(f_), 1 / 255.0f);                                        \
        } else {                                                                                    \
            p = cost_pair(gin[K & 1], oth[LEANA ? 0 : (K & 1)]);                                    \
            if (LEANA) oth[0] = pc_load4(rGo, vcp, r101c(mstart - 5 + (S) + 1, H) * W * 16);   /* partner pixels of the next step: the current ones are dead now */ \
            if (any_border) {   /* only where x < d (left) / x >= W-d (right) occurs in this wave */   \
                asm volatile("; border cost");   /* keeps this a real branch */                     \
                const float cb_ = cost_border(gin[K & 1]);                                          \
                p = inb ? p : cb_;                                                                  \
            }                                                                                       \
        }                                                                                           \
        const float m1_ = __fmul_rn(gin[K & 1].x, p), m2_ = __fmul_rn(gin[K & 1].y, p), m3_ = __fmul_rn(gin[K & 1].z, p); \
        if (DEEP) PSM_ISSUE_PIX(K & 1, (S) + 2)       /* this slot's pixels are dead: those of step S + 2 take their place */ \
        double h0 = hsum8<TOL>(p, i1, i2, i4);                                                      \
        double h1 = hsum8<TOL>(m1_, i1, i2, i4);                                                    \
        double h2 = hsum8<TOL>(m2_, i1, i2, i4);                                                    \
        double h3 = hsum8<TOL>(m3_, i1, i2, i4);                                                    \
        double n0 = vstep<K>(t0, h0), n1 = vstep<K>(t1, h1), n2 = vstep<K>(t2, h2), n3 = vstep<K>(t3, h3); \
        float4 r = solve_ab<FMA>(PSM_BOX(n0), PSM_BOX(n1), PSM_BOX(n2), PSM_BOX(n3), o2[LEANA ? 0 : (K & 1)], o3[LEANA ? 0 : (K & 1)], o4[LEANA ? 0 : (K & 1)]); \
        if ((DST) != nullptr && mvalid) (DST)[K * PC_MCOLS] = r;                                    \
        __builtin_amdgcn_sched_barrier(0);                                                          \
    }
        PSM_ISSUE_PA(0, 0)
        if (DEEP) { PSM_ISSUE_PIX(0, 0) PSM_ISSUE_PIX(1, 1) }
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
            float4 *dst = &ring[((B) + rb) & (PC_RING - 1)][0][wave * PC_OUT_A + lane];            \
            PSM_STEP_PA(0, s0, dst) PSM_STEP_PA(1, s0 + 1, dst) PSM_STEP_PA(2, s0 + 2, dst) PSM_STEP_PA(3, s0 + 3, dst) \
            PC_SYNC();                                                                             \
        }
        // two batches per iteration: the s4 slots of the vertical trees change registers with every update, so only after eight
        // steps is every value back in the register the loop header expects (one batch per iteration: 16 v_mov_b64 at the latch)
        for (int b = 0; b < nbA; b += 2) {
            PSM_BATCH_PA(b)
            if (__builtin_expect(b + 1 < nbA, 1)) PSM_BATCH_PA(b + 1)
        }
#undef PSM_BATCH_PA
        for (int b = nbA; b < (last_slice ? iters : nbB); ++b) PC_SYNC();      // (the three trailing barriers: after the chunk's last slice only)
#undef PSM_STEP_PA
#undef PSM_ISSUE_PA
#undef PSM_ISSUE_PIX
#undef PSM_ISSUE_GUI
#undef PSM_ISSUE_PA2
    } else {
        // ---------------- consumer: stage B ----------------
        // feed j (j = 0 .. nf-1) is model row r101(y0-4+j); from feed 7 on the tree yields output row y0+j-7
        const int wb = wave - PC_NA;
        const int bwidth = wb < PC_NB - 1 ? PC_OUT_B : PC_COLS - (PC_NB - 1) * PC_OUT_B;
        const int xb0 = xg + wb * PC_OUT_B;           // first output column of this wave
        const int xmod = xb0 - 4 + lane;              // model column this lane consumes
        int mc = r101(xmod, W) - xm0;                 // REFLECT_101 of the model planes, as ring column
        mc = mc < 0 ? 0 : (mc > PC_MCOLS - 1 ? PC_MCOLS - 1 : mc);
        const int xb = xb0 + lane;                    // output column of this lane
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
        const pc_u4 v_ = __builtin_amdgcn_raw_buffer_load_b128(rKc, lane * 16, (C) * (PC_COLS * 16), PC_K_LD_AUX); \
        kq = make_float4(__uint_as_float(v_.x), __uint_as_float(v_.y), __uint_as_float(v_.z), __uint_as_float(v_.w)); \
        kd4 = __builtin_amdgcn_raw_buffer_load_b32(rKd, lane * 4, (C) * (PC_COLS * 4), PC_K_LD_AUX); \
    }
        float4 kq = make_float4(__builtin_inff(), __builtin_inff(), __builtin_inff(), __builtin_inff());   // running minima of the current batch
        unsigned kd4 = 0;                                                                                  // and their disparities
        // MODE 2: the volume's key plane (handed over in the kcost argument), current keys of this lane's pixel one batch ahead
        long long *const keyp = reinterpret_cast<long long *>(kcost);
        const __amdgpu_buffer_rsrc_t rKy = pc_rsrc(MODE == 2 ? (const void *)keyp : (const void *)G1, (unsigned)HW * 8u);
        long long kcur[4] = {0, 0, 0, 0};
#define PSM_KEY_LOAD1(SLOT, J)   /* the key of feed row J */                                         \
    {                                                                                              \
        int yk_ = y0 + (J) - 7;                                                                    \
        yk_ = yk_ < 0 ? 0 : (yk_ > H - 1 ? H - 1 : yk_);                                           \
        const pc_u2 v_ = __builtin_amdgcn_raw_buffer_load_b64(rKy, xbc * 8, yk_ * W * 8, PC_KEY_LD_AUX); \
        kcur[SLOT] = (long long)(((unsigned long long)v_.y << 32) | v_.x);                         \
    }
        const __amdgpu_buffer_rsrc_t rG1 = pc_rsrc(G1, (unsigned)HW * 16u);
        const int vxb = xbc * 16;
        const int dg = d_begin + dyn.dstep * pc_slice(dyn, d);
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
            return &ring[((a >> 2) + rb) & (PC_RING - 1)][a & 3][mc];
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
                            // (nontemporal: the volume is next read long after it left the L2)
                            __builtin_nontemporal_store(*reinterpret_cast<const f4v *>(src + cc), reinterpret_cast<f4v *>(row + cc));
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
        constexpr bool LEANB = MODE == 2;   // key form: G1 of the output rows and the keys two rows ahead instead of a batch, selection inside the steps
        PSM_ISSUE_PB(0, 0) PSM_ISSUE_PB(1, 1)
        if (!LEANB) { PSM_ISSUE_PB(2, 2) PSM_ISSUE_PB(3, 3) }
        if constexpr (MODE == 1) {
            if (!first) {                              // records of batch 0, written by this wave one slice earlier (L1 bypassed)
                PSM_K_LOAD(0)
            }
        }
        if constexpr (MODE == 2) { PSM_KEY_LOAD1(0, 0) PSM_KEY_LOAD1(1, 1) }
        if (ds == 0) {                           // (two batches behind the producers: once per chunk - later slices stay in step)
            PC_SYNC();                           // iteration 0
            PC_SYNC();                           // iteration 1
        }
        // (two batches per loop iteration, as in the producer: no register moves of the s4 slots at the latch)
        auto batch_b = [&](const int c) __attribute__((always_inline)) {   // consume feed batch c
            if (c >= 1) store_batch(c - 1);
            {
                const int j0 = 4 * c;
                float4 a_cur = *model_of(j0), a_nxt;
                float qv[4];
#define PSM_STEP_PB(K)                                                                              \
    {                                                                                               \
        if (K < 3) a_nxt = *model_of(j0 + K + 1);     /* model row of the next feed, one step ahead */ \
        double h0 = hsum8<TOL>(a_cur.x, i1, i2, i4);                                                \
        double h1 = hsum8<TOL>(a_cur.y, i1, i2, i4);                                                \
        double h2 = hsum8<TOL>(a_cur.z, i1, i2, i4);                                                \
        double h3 = hsum8<TOL>(a_cur.w, i1, i2, i4);                                                \
        double n0 = vstep<K>(t0, h0), n1 = vstep<K>(t1, h1), n2 = vstep<K>(t2, h2), n3 = vstep<K>(t3, h3); \
        qv[K] = __fadd_rn(__fadd_rn(__fadd_rn(PSM_BOX(n3), __fmul_rn(PSM_BOX(n0), o1x[LEANB ? (K & 1) : K])),        \
                                    __fmul_rn(PSM_BOX(n1), o1y[LEANB ? (K & 1) : K])), __fmul_rn(PSM_BOX(n2), o1z[LEANB ? (K & 1) : K])); \
        if (U8) {   /* q8 = sat_u8(rintf(q * 255)), NaN -> 0 (oracle: quant_u8); kept as a float: the selection is unchanged */ \
            const float r_ = rintf(__fmul_rn(qv[K], 255.0f * QSCALE));   /* (255 * 2^-12 is exact: one rounding, as q * 255) */ \
            qv[K] = __builtin_amdgcn_fmed3f(r_, 0.0f, 255.0f);   /* saturate; a NaN input makes v_med3_f32 return min3 = 0 */ \
        } else if (SCALED) qv[K] = __fmul_rn(qv[K], QSCALE);                                        \
        if (LEANB) {                                                                                \
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
                } else if constexpr (MODE == 2) {
                    // (DispSel::CVSelect against the volume's shared key plane: done row by row inside the steps)
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
                    // (per lane: only lanes with an improved row rewrite their record)
                    if (lane < bwidth && (first || any)) {
                        const pc_u4 kv = {__float_as_uint(kn[0]), __float_as_uint(kn[1]), __float_as_uint(kn[2]), __float_as_uint(kn[3])};
                        __builtin_amdgcn_raw_buffer_store_b128(kv, rKc, lane * 16, c * (PC_COLS * 16), 0);
                        __builtin_amdgcn_raw_buffer_store_b32(dn, rKd, lane * 4, c * (PC_COLS * 4), 0);
                    }
                    if (!first && c + 1 < nbB) PSM_K_LOAD(c + 1)       // records of the next batch
                }
            }
            PC_SYNC();
        };
        for (int c = 0; c < nbB; c += 2) {
            batch_b(c);
            if (__builtin_expect(c + 1 < nbB, 1)) batch_b(c + 1);
        }
        store_batch(nbB - 1);                          // iteration nbB+2
        if (last_slice) PC_SYNC();
#undef PSM_ISSUE_PB
#undef PSM_KEY_LOAD1
#undef PSM_K_LOAD
    }
    if (MODE == 1) __builtin_amdgcn_s_waitcnt(0);      // the chunk planes of this slice are in the L2 before the next slice reads them
    first = false;
    rb = (rb + nbA) & (PC_RING - 1);
    }   // slices of the chunk
#undef PSM_BOX
    if (ts != nullptr && threadIdx.x == 0)            // ... and when did the last one end (all waves have passed the last barrier)
        (void)__hip_atomic_fetch_max(ts + PC_TS_SLOTS, (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    PC_TRACE_END()
}

// chunk planes -> packed WTA key and / or final map per pixel (minimum over the chunks; pack_key_f32 makes the signed
// 64-bit minimum select (min cost, then lowest d) exactly as the sequential loop of src/DispSel.cpp:96-104).
// One thread per record (pair, batch, column) = four rows of one column, as the select kernel wrote them.
__global__ __launch_bounds__(256) void k_chunk_min(const float4 *kcost, const unsigned *kdisp, int nchunks, int npairs,
                                                  int nbmax, int ngroups, int seg_rows, int W, int H, long long *keys,
                                                  uint8_t *map, const float4 *__restrict__ kcost1, const unsigned *__restrict__ kdisp1,
                                                  int ybeg, int yend, const PcPair *__restrict__ batch, int to_maps, int cols)
{   // cols: output columns per (column group) of the layout the select kernel ran with (107 / 50)
    if (batch) {             // batched launch: blockIdx.z = pair, planes / keys / maps from the table
        const PcPair pp = batch[blockIdx.z];
        const size_t rec_total = (size_t)npairs * nbmax * cols * nchunks;
        kcost = (const float4 *)pp.scratch;
        kdisp = (const unsigned *)(kcost + rec_total);
        kcost1 = (const float4 *)((const char *)pp.scratch + rec_total * 20);
        kdisp1 = (const unsigned *)(kcost1 + rec_total);
        keys = pp.keys;
        map = to_maps ? pp.maps : nullptr;
    }
    if (blockIdx.y == 1) {   // second volume of a two-side launch: its own planes, keys / map one image further
        kcost = kcost1;
        kdisp = kdisp1;
        if (keys) keys += (size_t)W * H;
        if (map) map += (size_t)W * H;
    }
    const size_t nrec = (size_t)npairs * nbmax * cols;          // records per chunk plane
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nrec) return;
    const int col = (int)(idx % cols);
    size_t t = idx / cols;
    const int c = (int)(t % nbmax);
    const int pair = (int)(t / nbmax);
    const int g = pair % ngroups, seg = pair / ngroups;
    const int x = g * cols + col;
    if (x >= W) return;
    const int y0 = ybeg + seg * seg_rows, y1 = min(yend, y0 + seg_rows);
    const int ya = y0 + 4 * c - 7;                                  // output row of the record's first entry
    if (ya + 3 < y0 || ya >= y1) return;
    // (One chunk's record per loop iteration, on purpose.  Round 6 kept 2 / 4 / 8 / 16 chunks' records in flight per thread: alone the
    // kernel got faster - 17.8 -> 13.7 us at 450 x 375 x 64 -, but with two frames in flight the frame rose from 0.211 to 0.229 ms:
    // beside three resident workgroups of the fused kernel a SIMD has 32 vector registers left, and only this small form still fits
    // there; the wider ones wait for a fused workgroup to leave.  profiles/r06/exp_chunk_min_unroll.txt)
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

namespace psm {

// key plane(s) <- key(+inf, 0): the "no candidate yet" value of MODE 2
__global__ __launch_bounds__(256) void k_fill_keys(long long *__restrict__ keys, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = pack_key_f32(__builtin_inff(), 0);
}

// What the planner needs to know about the device the calling thread is bound to: XCDs (blocks go round-robin over them, each
// has its own L2) and CUs per XCD - from the runtime, not assumed (MI355X: 8 x 32).
PcDev pc_dev()
{
    static PcDev cache[64];
    static std::mutex mu;                        // hosts drive one context per thread / GPU: first use may be concurrent
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    std::lock_guard<std::mutex> lock(mu);
    PcDev &d = cache[dev];
    if (d.nxcd == 0) {
        int nx = 0, cus = 0;
        if (hipDeviceGetAttribute(&nx, hipDeviceAttributeNumberOfXccs, dev) != hipSuccess || nx < 1) nx = 1;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        (void)hipGetLastError();
        d.cus_per_xcd = cus / nx > 0 ? cus / nx : 1;
        d.nxcd = nx;
    }
    return d;
}


// Segment count k (and, for the plane form, slices per chunk DC): every segment re-walks 14 halo rows, and the launch runs in
// rounds of resident workgroups - per XCD ceil(pairs / nxcd) (column group, segment) pairs x chunks over cus_per_xcd CUs x 3
// workgroups (key form: x 4).  Cost model, fitted to measurements at 1080p (DC = 1, 2, 4, 8, 16: 4.39, 4.41, 4.46, 4.71,
// 4.99 ms for kernel + reduction): (rounds + 1/2) x rows walked per workgroup - the last round is on average half empty,
// which is what makes long-running workgroups (large DC) expensive - plus two row-steps per chunk plane for the reduction.
// Round 5 added the choice of the column-group layout (wide / narrow) and, for several small pairs at a time, a flow model
// in place of the rounds (both below).
PcPlan pc_plan(int W, int rows, int Dloc, int seg_rows_opt, int form, int batch, int inflight)
{   // form: PC_STORE; PC_PLANES (select with chunk planes) / PC_KEYS (select against a shared key plane, one slice per
    // workgroup, no reduction afterwards), each + PC_BOTH when one launch covers both volumes (twice the work items)
    const bool planes = (form & 3) == PC_PLANES, keys = (form & 3) == PC_KEYS;
    if (PSM_KNOB(planes ? "PSM_PC_SEGP" : "PSM_PC_SEGK", 0) > 0 && (planes || keys))      // (experiment builds: segment rows per form)
        seg_rows_opt = PSM_KNOB(planes ? "PSM_PC_SEGP" : "PSM_PC_SEGK", 0);
    const int sides = ((form & PC_BOTH) ? 2 : 1) * (batch > 1 ? batch : 1);   // volumes per launch
    const PcDev dev = pc_dev();
    PcPlan pl;
    pl.nxcd = dev.nxcd;
    // layout: the narrow one (2 waves per 50 columns) where it needs fewer waves per image row than the wide one (4 per 107);
    // both-volume select launches only (the product path); ties go to the wide layout
    constexpr int WIDE = PcLayout<1>::COLS, NARROW = PcLayout<1, true>::COLS;
    const bool select2 = (form & 3) != PC_STORE && (form & PC_BOTH);
    const int narrow_knob = PSM_KNOB("PSM_PC_NARROW", 0);        // (experiment builds: 1 forces the narrow layout, 2 the wide one)
    pl.narrow = select2 && (narrow_knob ? narrow_knob == 1 : 2 * ((W + NARROW - 1) / NARROW) < 4 * ((W + WIDE - 1) / WIDE));
    const int cols = (form & 3) == PC_STORE ? PcLayout<0>::COLS : (pl.narrow ? NARROW : WIDE);
    pl.cols = cols;
    pl.ngroups = (W + cols - 1) / cols;
    const int kdiv = PSM_KNOB("PSM_PC_KDIV", rows < 128 ? 32 : 64);     // (under 128 rows: segments of 32+ rows - 150 x 120: 0.082 -> 0.057 ms)
    const int kmax = rows / kdiv > 1 ? rows / kdiv : 1;
    int dcs[5] = {1, 2, 4, 8, 16};
    int ndc = planes ? 5 : 1;
    if (planes && PSM_KNOB("PSM_PC_DC", 0) > 0) { dcs[0] = PSM_KNOB("PSM_PC_DC", 0); ndc = 1; }
    const long slots = PSM_KNOB("PSM_PC_SLOTS", (long)dev.cus_per_xcd * (keys ? 4 : 3) * (pl.narrow ? 2 : 1));   // resident workgroups per XCD (two-wave workgroups: twice as many)
    // Many small items (a batch of pairs, or frames in flight on other streams: `inflight`) do not run in lockstep rounds: the
    // launch behaves like a flow - work / slots plus half an item of tail - and the reduction hides under the next pair's
    // filter.  Measured in round 5 on batches of 8 (profiles/r05/exp_plan_model.txt): against the rounds model 450 x 375
    // 0.197 -> 0.178 ms per pair, 340 x 256 0.104 -> 0.097, 150 x 120 0.022 -> 0.016; two frames in flight 450 x 375 0.203 ->
    // 0.188.  Single pairs and everything from 1280 x 720 up keep the rounds model (the flow model loses 1 - 7 % there: long
    // uniform items do run in rounds).
    const int conc = inflight > 1 ? inflight : 1;
    const int model_knob = PSM_KNOB("PSM_PC_MODEL", 0);    // (experiment builds: 1 forces the flow model, 2 the rounds model)
    const bool flow = model_knob ? model_knob == 1 : (sides * conc > 2 && (long)W * rows < 524288);
    auto cost_of = [&](int dc, int kk) -> long {
        const int nch = (Dloc + dc - 1) / dc;
        if (flow) {
            const double items = (double)sides * conc * pl.ngroups * kk * nch, len = dc * ((rows + kk - 1) / kk + 14) + 10.0;
            double c = items * len / ((double)slots * dev.nxcd) + 0.5 * len;
            if (planes && dc == 1) c *= 1.06;
            return (long)(c * 16.0);
        }
        const long per_xcd = ((long)sides * pl.ngroups * kk * nch + dev.nxcd - 1) / dev.nxcd;
        if (keys) {
            // Key form (round 6, profiles/r06/exp_plan_two_phase.txt): one slice per item, an item walks its rows + 14 halo rows + ~16
            // rows of pipeline fill, and a launch ends with a round and a half of stragglers - (rounds + 3/2) x (rows / k + 30).
            // 1080p x 256 with 224 key slices: 3 segments (6.79 ms) where the older (rounds + 1/2) x (rows / k + 14) cut 4 (6.89) or
            // 2 (6.93); 720p x 128: 3; 4K x 256: 4.
            return (2 * ((per_xcd + slots - 1) / slots) + 3) * ((rows + kk - 1) / kk + 30) / 2;
        }
        const long rounds2 = 2 * ((per_xcd + slots - 1) / slots) + 1;   // 2 x (rounds + 1/2)
        long c = rounds2 * dc * ((rows + kk - 1) / kk + 14) / 2 + (planes ? 2L * sides * nch : 0);
        // one slice per plane rewrites every record (a chunk's later slices only the improved ones): +6 % once the planes of
        // an image no longer sit in the L2s (measured at 1080p, 32 local slices: DC 1 / 2 = 1.28 / 1.17 ms; at 450 x 375 DC 1 wins)
        if (planes && dc == 1) c += (long)(0.06 * c * ((double)W * rows >= 2097152.0 ? 1.0 : (double)W * rows / 2097152.0));
        return c;
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
    pl.nchunks = (Dloc + bdc - 1) / bdc;
    pl.nbmax = (pl.seg_rows + 7 + 3) / 4;                            // consumer batches of a full segment
    pl.rec_per_chunk = (size_t)pl.ngroups * pl.nsegs * pl.nbmax * pl.cols;
    pl.rec_bytes = 20;                                               // 16 bytes of costs + 4 bytes of disparities
    return pl;
}

constexpr int PC_KEY_SPREAD = 4;    // passes of the key form's slice order (pc_slice; measured 1 / 2 / 4 / 8 / 16: f32 5.12 / 5.07 / 5.06 / 5.08 / 5.10 ms, 8-bit 5.94 / 5.82 / 5.80 / 5.81 / 5.83)

int pc_seed_stride(int W, int rows, bool u8)
{   // every S-th slice goes through the minima planes and seeds the key plane.  Rounds 3-5: 5, and 4 from 4 Mpixel up (flat from 5 to
    // 12 then).  Round 6, with the key loads served by the L2 (PC_KEY_LD_AUX) a pixel's key costs less to read and fewer seeds pay:
    // S = 8 - same-box A/B against 5 (4 at 4K): 4K x 256 27.63 -> 26.43 ms, 1080p x 256 6.81 -> 6.72 (with the key launch's
    // three-segment cut, pc_plan), 720p x 128 1.706 -> 1.661, 8-bit 720p 1.949 -> 1.817; 10 and 12 lose again.  Two places keep 5,
    // where how the two launches fill their rounds of workgroups weighs more: stripes under 200 rows (1/8 of 1080p: 0.951 vs 0.963)
    // and 8-bit mode from 1.5 Mpixel up, whose key form is the dearer one (1080p x 256: 7.38 vs 7.54).  profiles/r06/exp_plan_two_phase.txt
    const int e = PSM_KNOB("PSM_PC_S", 0);
    if (e > 1) return e;
    if (rows < 200 || (u8 && (size_t)W * rows >= 1500000)) return 5;      // (8-bit: 1280 x 720 takes 8, 1920 x 1080 takes 5)
    return 8;
}

static int pc_blocks(const PcPlan &pl, int chunks) { return pl.nxcd * ((pl.ngroups * pl.nsegs * chunks + pl.nxcd - 1) / pl.nxcd); }

// Storing form (MODE 0): vin (or, cvc_mode 1 / 2, the costs built on the fly) -> vout, one slice per workgroup.
void launch_cvf_fused(hipStream_t s, March m, const float *vin, float *vout, Guidance gd, int W, int H, int Dloc,
                      int ybeg, int yend, const float4 *g1_other, int d_begin, int cvc_mode, unsigned long long *ts)
{
    if (yend <= ybeg) return;
    const PcPlan pl = pc_plan(W, yend - ybeg, Dloc, m.seg_rows, PC_STORE);
    const dim3 grid(pc_blocks(pl, Dloc)), blk(64 * (PcLayout<0>::NA + PcLayout<0>::NB));
#define PSM_LAUNCH_PC(V4, CV, VR)                                                                                          \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cvf_pc<V4, CV, 0, false, false, VR>), grid, blk, 0, s, vin, vout, (const float4 *)gd.g1, \
                       (const float4 *)gd.g2, (const float4 *)gd.g3, (const float2 *)gd.g4, W, H, Dloc, pl.ngroups, pl.nsegs, \
                       pl.seg_rows, ybeg, yend, g1_other, d_begin, 1, (float *)nullptr, (unsigned *)nullptr, 0, PcSide{}, PcSel{0, 1, pl.nxcd, 0, Dloc, 0, 1}, ts, (const PcPair *)nullptr)
#define PSM_LAUNCH_PCV(V4, CV) { if (m.flags & PSM_FLAG_FMA_SOLVE) PSM_LAUNCH_PC(V4, CV, 2); else PSM_LAUNCH_PC(V4, CV, 0); }
    const bool v4 = (W & 3) == 0;
    if (cvc_mode == 1) { if (v4) PSM_LAUNCH_PCV(true, 1) else PSM_LAUNCH_PCV(false, 1) }
    else if (cvc_mode == 2) { if (v4) PSM_LAUNCH_PCV(true, 2) else PSM_LAUNCH_PCV(false, 2) }
    else { if (v4) PSM_LAUNCH_PCV(true, 0) else PSM_LAUNCH_PCV(false, 0) }
#undef PSM_LAUNCH_PCV
#undef PSM_LAUNCH_PC
}

// Select form with chunk planes (MODE 1), one volume: costs read from vin (cvc_mode 0) or built on the fly (1 / 2; p4_own !=
// NULL: 8-bit char mode).  scratch: pc_plan(..., PC_PLANES).scratch_bytes().
void launch_cvf_select(hipStream_t s, March m, const float *vin, Guidance gd, int W, int H, int Dloc, const float4 *g1_other,
                       int d_begin, int cvc_mode, void *scratch, unsigned long long *ts, const uint8_t *p4_own, const uint8_t *p4_other)
{
    const PcPlan pl = pc_plan(W, m.rows(H), Dloc, m.seg_rows, PC_PLANES);
    const PcSel sel = {0, 1, pl.nxcd, 0, Dloc, 0, 1};
    float *kcost = (float *)scratch;                                           // nchunks * rec_per_chunk float4
    unsigned *kdisp = (unsigned *)(kcost + 4 * pl.rec_per_chunk * pl.nchunks);  // nchunks * rec_per_chunk uchar4
    const dim3 grid(pc_blocks(pl, pl.nchunks)), blk(64 * (PcLayout<1>::NA + PcLayout<1>::NB));
#define PSM_LAUNCH_PC(CV, U8V, VR, A0, A1)                                                                                  \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cvf_pc<false, CV, 1, U8V, false, VR>), grid, blk, 0, s, A0, A1, (const float4 *)gd.g1, \
                       (const float4 *)gd.g2, (const float4 *)gd.g3, (const float2 *)gd.g4, W, H, Dloc, pl.ngroups, pl.nsegs, \
                       pl.seg_rows, m.y0(H), m.y1(H), g1_other, d_begin, pl.DC, kcost, kdisp, pl.nbmax, PcSide{}, sel, ts, (const PcPair *)nullptr)
    const bool fma = (m.flags & PSM_FLAG_FMA_SOLVE) != 0;      // (one volume per launch: the canon and the FMA reading; no tolerance form)
    if (p4_own && cvc_mode != 0) {
        if (cvc_mode == 1) PSM_LAUNCH_PC(1, true, 0, (const float *)p4_own, (float *)const_cast<uint8_t *>(p4_other));
        else PSM_LAUNCH_PC(2, true, 0, (const float *)p4_own, (float *)const_cast<uint8_t *>(p4_other));
    } else if (cvc_mode == 1) { if (fma) PSM_LAUNCH_PC(1, false, 2, vin, (float *)nullptr); else PSM_LAUNCH_PC(1, false, 0, vin, (float *)nullptr); }
    else if (cvc_mode == 2) { if (fma) PSM_LAUNCH_PC(2, false, 2, vin, (float *)nullptr); else PSM_LAUNCH_PC(2, false, 0, vin, (float *)nullptr); }
    else { if (fma) PSM_LAUNCH_PC(0, false, 2, vin, (float *)nullptr); else PSM_LAUNCH_PC(0, false, 0, vin, (float *)nullptr); }
#undef PSM_LAUNCH_PC
}

void launch_chunk_min(hipStream_t s, March m, int W, int H, int Dloc, void *scratch, long long *keys, uint8_t *map)
{
    const PcPlan pl = pc_plan(W, m.rows(H), Dloc, m.seg_rows, PC_PLANES);
    const float *kcost = (const float *)scratch;
    const unsigned *kdisp = (const unsigned *)(kcost + 4 * pl.rec_per_chunk * pl.nchunks);
    hipLaunchKernelGGL(k_chunk_min, dim3((unsigned)((pl.rec_per_chunk + 255) / 256)), dim3(256), 0, s, (const float4 *)kcost, (const unsigned *)kdisp,
                       pl.nchunks, pl.ngroups * pl.nsegs, pl.nbmax, pl.ngroups, pl.seg_rows, W, H, keys, map, (const float4 *)nullptr,
                       (const unsigned *)nullptr, m.y0(H), m.y1(H), (const PcPair *)nullptr, 0, pl.cols);
}

// Both volumes in one launch each (costs built on the fly): left volume = (g[0], other g[1].g1), right = (g[1], other g[0].g1).
// Dloc = number of slices of this launch, (sel, step) = which ones (PcSel).  p4 != NULL: 8-bit char mode, p4[0] / p4[1] = byte
// planes {c0,c1,c2,grad} of the left / right image.
// ... plane form: scratch = 2 x pc_plan(..., PC_PLANES | PC_BOTH).scratch_bytes()
void launch_cvf_select2(hipStream_t s, March m, const Guidance *g, int W, int H, int Dloc, int d_begin, void *scratch,
                        unsigned long long *ts, const uint8_t *const *p4, int sel, int step)
{
    const bool tol = !p4 && (m.flags & PSM_FLAG_F32_TOL), fma = !p4 && (m.flags & PSM_FLAG_FMA_SOLVE);
    const PcPlan pl = pc_plan(W, m.rows(H), Dloc, m.seg_rows, PC_PLANES | PC_BOTH, 1, m.inflight);
    const PcSel ps = {sel, step, pl.nxcd, 0, Dloc, 0, m.dstep};
    float *kcost0 = (float *)scratch;
    unsigned *kdisp0 = (unsigned *)(kcost0 + 4 * pl.rec_per_chunk * pl.nchunks);
    float *kcost1 = (float *)((char *)scratch + pl.scratch_bytes());
    unsigned *kdisp1 = (unsigned *)(kcost1 + 4 * pl.rec_per_chunk * pl.nchunks);
    const dim3 grid(pc_blocks(pl, pl.nchunks), 2), blk(pl.narrow ? 64 * (PcLayout<1, true>::NA + PcLayout<1, true>::NB) : 64 * (PcLayout<1>::NA + PcLayout<1>::NB));
    const PcSide s1 = {(const float4 *)g[1].g1, (const float4 *)g[1].g2, (const float4 *)g[1].g3, (const float2 *)g[1].g4, (const float4 *)g[0].g1, kcost1, kdisp1};
#define PSM_LAUNCH_PC(U8V, VR, NW, A0, A1)                                                                                   \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cvf_pc<false, 3, 1, U8V, false, VR, NW>), grid, blk, 0, s, A0, A1, (const float4 *)g[0].g1, \
                       (const float4 *)g[0].g2, (const float4 *)g[0].g3, (const float2 *)g[0].g4, W, H, Dloc, pl.ngroups, pl.nsegs, pl.seg_rows, m.y0(H), m.y1(H), \
                       (const float4 *)g[1].g1, d_begin, pl.DC, kcost0, kdisp0, pl.nbmax, s1, ps, ts, (const PcPair *)nullptr)
#define PSM_LAUNCH_PCN(U8V, VR, A0, A1) { if (pl.narrow) PSM_LAUNCH_PC(U8V, VR, true, A0, A1); else PSM_LAUNCH_PC(U8V, VR, false, A0, A1); }
    if (p4) PSM_LAUNCH_PCN(true, 0, (const float *)p4[0], (float *)const_cast<uint8_t *>(p4[1]))
    else if (fma) PSM_LAUNCH_PCN(false, 2, (const float *)nullptr, (float *)nullptr)
    else if (tol) PSM_LAUNCH_PCN(false, 1, (const float *)nullptr, (float *)nullptr)
    else PSM_LAUNCH_PCN(false, 0, (const float *)nullptr, (float *)nullptr)
#undef PSM_LAUNCH_PCN
#undef PSM_LAUNCH_PC
}

void launch_chunk_min2sides(hipStream_t s, March m, int W, int H, int Dloc, void *scratch, long long *keys, uint8_t *map)
{
    const PcPlan pl = pc_plan(W, m.rows(H), Dloc, m.seg_rows, PC_PLANES | PC_BOTH, 1, m.inflight);
    const float *kcost0 = (const float *)scratch;
    const unsigned *kdisp0 = (const unsigned *)(kcost0 + 4 * pl.rec_per_chunk * pl.nchunks);
    const float *kcost1 = (const float *)((const char *)scratch + pl.scratch_bytes());
    const unsigned *kdisp1 = (const unsigned *)(kcost1 + 4 * pl.rec_per_chunk * pl.nchunks);
    hipLaunchKernelGGL(k_chunk_min, dim3((unsigned)((pl.rec_per_chunk + 255) / 256), 2), dim3(256), 0, s, (const float4 *)kcost0, kdisp0,
                       pl.nchunks, pl.ngroups * pl.nsegs, pl.nbmax, pl.ngroups, pl.seg_rows, W, H, keys, map, (const float4 *)kcost1, kdisp1, m.y0(H), m.y1(H),
                       (const PcPair *)nullptr, 0, pl.cols);
}

// ... key form (MODE 2): keys[2][H][W] receives the packed minima (init: start from key(+inf, 0); otherwise continue from what
// `keys` holds - the second phase of the two-phase selection)
void launch_cvf_select_keys2(hipStream_t s, March m, const Guidance *g, int W, int H, int Dloc, int d_begin, long long *keys,
                             unsigned long long *ts, const uint8_t *const *p4, int init, int sel, int step)
{
    const PcPlan pl = pc_plan(W, m.rows(H), Dloc, m.seg_rows, PC_KEYS | PC_BOTH, 1, m.inflight);
    const PcSel ps = {sel, step, pl.nxcd, PSM_KNOB("PSM_PC_SPREAD", PC_KEY_SPREAD), Dloc, 0, m.dstep};
    const size_t HW = (size_t)W * H;
    if (init) hipLaunchKernelGGL(k_fill_keys, dim3((unsigned)((2 * HW + 255) / 256)), dim3(256), 0, s, keys, 2 * HW);
    const dim3 grid(pc_blocks(pl, Dloc), 2), blk(pl.narrow ? 64 * (PcLayout<2, true>::NA + PcLayout<2, true>::NB) : 64 * (PcLayout<2>::NA + PcLayout<2>::NB));
    const PcSide s1 = {(const float4 *)g[1].g1, (const float4 *)g[1].g2, (const float4 *)g[1].g3, (const float2 *)g[1].g4, (const float4 *)g[0].g1,
                       (float *)(keys + HW), nullptr};
#define PSM_LAUNCH_PC(U8V, VR, NW, A0, A1)                                                                                   \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cvf_pc<false, 3, 2, U8V, false, VR, NW>), grid, blk, 0, s, A0, A1,                     \
                       (const float4 *)g[0].g1, (const float4 *)g[0].g2, (const float4 *)g[0].g3, (const float2 *)g[0].g4, W, H, Dloc, pl.ngroups, \
                       pl.nsegs, pl.seg_rows, m.y0(H), m.y1(H), (const float4 *)g[1].g1, d_begin, 1, (float *)keys, (unsigned *)nullptr, 0, s1, ps, ts, (const PcPair *)nullptr)
#define PSM_LAUNCH_PCN(U8V, VR, A0, A1) { if (pl.narrow) PSM_LAUNCH_PC(U8V, VR, true, A0, A1); else PSM_LAUNCH_PC(U8V, VR, false, A0, A1); }
    if (p4) PSM_LAUNCH_PCN(true, 0, (const float *)p4[0], (float *)const_cast<uint8_t *>(p4[1]))
    else if (m.flags & PSM_FLAG_FMA_SOLVE) PSM_LAUNCH_PCN(false, 2, (const float *)nullptr, (float *)nullptr)
    else if (m.flags & PSM_FLAG_F32_TOL) PSM_LAUNCH_PCN(false, 1, (const float *)nullptr, (float *)nullptr)
    else PSM_LAUNCH_PCN(false, 0, (const float *)nullptr, (float *)nullptr)
#undef PSM_LAUNCH_PCN
#undef PSM_LAUNCH_PC
}

// ---- the same launches for `npairs` stereo pairs at once (psm_compute_batch): blockIdx.z = pair, pointers from the table ----
void launch_cvf_select2_batch(hipStream_t s, March m, const PcPair *tab, int npairs, int W, int H, int Dloc, int d_begin,
                              unsigned long long *ts, bool u8, int sel, int step)
{
    const PcPlan pl = pc_plan(W, m.rows(H), Dloc, m.seg_rows, PC_PLANES | PC_BOTH, npairs);
    const PcSel ps = {sel, step, pl.nxcd, 0, Dloc, (unsigned long long)pl.rec_per_chunk * pl.nchunks, m.dstep};
    const dim3 grid(pc_blocks(pl, pl.nchunks), 2, npairs), blk(pl.narrow ? 64 * (PcLayout<1, true>::NA + PcLayout<1, true>::NB) : 64 * (PcLayout<1>::NA + PcLayout<1>::NB));
#define PSM_LAUNCH_PCB(U8V) { if (pl.narrow) PSM_LAUNCH_PCBN(U8V, true); else PSM_LAUNCH_PCBN(U8V, false); }
#define PSM_LAUNCH_PCBN(U8V, NW)                                                                                              \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cvf_pc<false, 3, 1, U8V, true, 0, NW>), grid, blk, 0, s, (const float *)nullptr, (float *)nullptr,  \
                       (const float4 *)nullptr, (const float4 *)nullptr, (const float4 *)nullptr, (const float2 *)nullptr, W, H, Dloc, \
                       pl.ngroups, pl.nsegs, pl.seg_rows, m.y0(H), m.y1(H), (const float4 *)nullptr, d_begin, pl.DC, (float *)nullptr, \
                       (unsigned *)nullptr, pl.nbmax, PcSide{}, ps, ts, tab)
    if (u8) PSM_LAUNCH_PCB(true) else PSM_LAUNCH_PCB(false)
#undef PSM_LAUNCH_PCB
#undef PSM_LAUNCH_PCBN
}

void launch_chunk_min2sides_batch(hipStream_t s, March m, const PcPair *tab, int npairs, int W, int H, int Dloc, bool to_maps)
{
    const PcPlan pl = pc_plan(W, m.rows(H), Dloc, m.seg_rows, PC_PLANES | PC_BOTH, npairs);
    hipLaunchKernelGGL(k_chunk_min, dim3((unsigned)((pl.rec_per_chunk + 255) / 256), 2, npairs), dim3(256), 0, s, (const float4 *)nullptr,
                       (const unsigned *)nullptr, pl.nchunks, pl.ngroups * pl.nsegs, pl.nbmax, pl.ngroups, pl.seg_rows, W, H, (long long *)nullptr,
                       (uint8_t *)nullptr, (const float4 *)nullptr, (const unsigned *)nullptr, m.y0(H), m.y1(H), tab, to_maps ? 1 : 0, pl.cols);
}

void launch_cvf_select_keys2_batch(hipStream_t s, March m, const PcPair *tab, int npairs, int W, int H, int Dloc, int d_begin,
                                   unsigned long long *ts, bool u8, int sel, int step)
{
    const PcPlan pl = pc_plan(W, m.rows(H), Dloc, m.seg_rows, PC_KEYS | PC_BOTH, npairs);
    const PcSel ps = {sel, step, pl.nxcd, PC_KEY_SPREAD, Dloc, 0, m.dstep};
    const dim3 grid(pc_blocks(pl, Dloc), 2, npairs), blk(pl.narrow ? 64 * (PcLayout<2, true>::NA + PcLayout<2, true>::NB) : 64 * (PcLayout<2>::NA + PcLayout<2>::NB));
#define PSM_LAUNCH_PCB(U8V) { if (pl.narrow) PSM_LAUNCH_PCBN(U8V, true); else PSM_LAUNCH_PCBN(U8V, false); }
#define PSM_LAUNCH_PCBN(U8V, NW)                                                                                              \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cvf_pc<false, 3, 2, U8V, true, 0, NW>), grid, blk, 0, s, (const float *)nullptr, (float *)nullptr,  \
                       (const float4 *)nullptr, (const float4 *)nullptr, (const float4 *)nullptr, (const float2 *)nullptr, W, H, Dloc, \
                       pl.ngroups, pl.nsegs, pl.seg_rows, m.y0(H), m.y1(H), (const float4 *)nullptr, d_begin, 1, (float *)nullptr,      \
                       (unsigned *)nullptr, 0, PcSide{}, ps, ts, tab)
    if (u8) PSM_LAUNCH_PCB(true) else PSM_LAUNCH_PCB(false)
#undef PSM_LAUNCH_PCB
#undef PSM_LAUNCH_PCBN
}

}  // namespace psm
