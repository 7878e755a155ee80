// psm_pp.hip - post-processing "next" row: the plain weighted-median filter of src/PP.cpp:145-247 (wgtMedian) on gfx950.
//
// The reference filters the map IN PLACE in raster order: every invalid pixel is replaced by the weighted median of the
// disparities in its 19 x 19 (modulo-wrapped) window, and that window already contains the filtered values of the
// invalid pixels before it in raster order.  A data-parallel "all pixels from the input map" evaluation is a different
// filter.  What is parallel: the dependence is local - a pixel only depends on the earlier INVALID pixels inside its
// window - so the rows run as a dataflow pipeline:
//   * one wave per image row, rows dispatched in order; the wave walks its invalid pixels left to right;
//   * prog[y] = "every invalid pixel of row y left of this column is final".  Before a pixel is evaluated, lanes 0..18
//     (one per window row) wait until no unfinished invalid pixel of an EARLIER row lies inside the window
//     (nxt[y][x] = next invalid column >= x makes that test O(1)); rows later in raster order are read as they are: the
//     window relation is symmetric on the torus, so a later pixel inside the window cannot have been filtered yet - it
//     is itself waiting for this one;
//   * no fences: the map is read with agent-scope atomic dword loads and updated with one returning agent-scope
//     atomic XOR of the byte's changed bits (only this wave ever writes that byte; rows may share a dword when W % 4
//     != 0), and prog[y] is stored after the XOR has returned.  Release/acquire fences (L2 write-back + invalidate per
//     pixel) made the first version ~100x slower: 1.9 s instead of tens of ms for a 450 x 375 pair;
//   * the 361 weights of a pixel are evaluated by the 64 lanes in parallel (fp32 colour distance op for op, double exp
//     narrowed to float as the reference does); the histogram bins and the total are then accumulated in window raster
//     order (lane l owns bins l, l+64, ..: every lane scans the 361 (disparity, weight) pairs from LDS and adds the ones
//     that fall into its bins, so every float sum is formed in exactly the reference's order);
//   * the threshold scan runs over the non-empty bins in ascending order (adding an empty bin is the identity).
// A waiting wave only ever waits for rows with a smaller block index, so in-order dispatch guarantees progress; a
// watchdog turns a (never observed) stall into an error flag instead of a hang.
#include "psm_kernels.h"
#include "psm_dev.h"

#define PSM_EXP_FN __device__ __forceinline__
#define PSM_EXP_FMA(a, b, c) __fma_rn(a, b, c)
#include "psm_exp.h"
#include "psm_wm_tab.h"

namespace psm {

constexpr int WM_R = 9;                 // MED_SZ / 2, include/PP.h:12
constexpr int WM_K = 2 * WM_R + 1;      // 19
constexpr int WM_TAPS = WM_K * WM_K;    // 361

// nxt[y][x] (x = 0..W) = smallest x' >= x with valid[y][x'] == 0, W if there is none; prog[y] = nxt[y][0]
__global__ __launch_bounds__(64) void k_wm_next(const uint8_t *__restrict__ valid, int W, int *__restrict__ nxt, int *__restrict__ prog)
{
    const int y = blockIdx.x, lane = threadIdx.x;
    const uint8_t *v = valid + (size_t)y * W;
    int *n = nxt + (size_t)y * (W + 1);
    int carry = W;
    if (lane == 0) n[W] = W;
    for (int x0 = ((W - 1) / 64) * 64; x0 >= 0; x0 -= 64) {
        const int x = x0 + lane;
        const bool inv = x < W && v[x] == 0;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(inv);
        const unsigned long long up = m >> lane;            // invalid flags of columns x, x+1, ...
        const int nx = up ? x + __builtin_ctzll(up) : carry;
        if (x < W) n[x] = nx;
        if (m) carry = x0 + __builtin_ctzll(m);
    }
    if (lane == 0) prog[y] = carry;
}

__device__ const unsigned long long wm_exp_tab[256] = PSM_EXP_TAB_INIT;

// The spatial term -disWgt / (SIG_DIS * SIG_DIS) depends on the tap only: a table of its 361 float values per map (psm_wm_tab.h,
// generated with the reference's float arithmetic) instead of a division - and for the right map a root - per weight.
__device__ const unsigned wm_sp_left[WM_TAPS] = PSM_WM_SP_TAB_LEFT;
__device__ const unsigned wm_sp_right[WM_TAPS] = PSM_WM_SP_TAB_RIGHT;

// clrWgt / (SIG_CLR * SIG_CLR) (float / double -> double, src/PP.cpp:175,224) without the division: one Newton step on
// x * RN(1 / c) with fused multiply-adds.  Bit-identical to __ddiv_rn((double)x, 0.1 * 0.1) for EVERY non-negative finite float x -
// checked exhaustively, 2^31 operands, by scripts/exp/wm_arith.hip on the device (a double division is ~35 instructions);
// +inf and NaN take the division itself.
__device__ __forceinline__ double wm_div_sig_clr(float xf)
{
    const double c = 0.1 * 0.1, r = 1.0 / (0.1 * 0.1);
    const double x = (double)xf;
    if (!(xf < __builtin_inff())) return __ddiv_rn(x, c);   // +inf (squared colour differences of float images overflow) / NaN: the fused
    const double q0 = __dmul_rn(x, r);                      // form would turn inf into NaN (fma(-inf, c, inf)); the division gives the reference's inf
    return __fma_rn(__fma_rn(-q0, c, x), r, q0);
}
// sqrt(float) of <cmath>, correctly rounded.  __fsqrt_rn of this ROCm is NOT (one ulp low for sqrt(162.0f) and for 15 % of random
// operands, scripts/exp/sq.hip - which made one pixel of a 230 x 110 map come out 46 where the oracle says 48 in round 3); the
// double-precision root narrowed to float is (53 >= 2 x 24 + 2 bits) and was round 3's form.  From 2^-100 up the classic
// refinement of the hardware's reciprocal root - s = x r; s' = fma(fma(-s, s, x), r / 2, s) - gives the same float for EVERY
// operand (exhaustive, scripts/exp/wm_arith.hip); below (and for 0) the double root stays.
__device__ __forceinline__ float wm_sqrtf(float x)
{
    if (!(x >= 0x1p-100f) || !(x < __builtin_inff())) return (float)sqrt((double)x);   // (+inf: rsq gives 0 and inf * 0 a NaN; sqrt gives inf)
    const float r = __builtin_amdgcn_rsqf(x);
    const float s = __fmul_rn(x, r);
    return __fmaf_rn(__fmaf_rn(-s, s, x), __fmul_rn(0.5f, r), s);
}

template <bool RIGHT>
__device__ __forceinline__ float wm_weight(float4 p, float4 q, int wx, int wy)
{
    // src/PP.cpp:169-175 (left) / 216-224 (right)
    const float sp = __uint_as_float((RIGHT ? wm_sp_right : wm_sp_left)[(wy + WM_R) * WM_K + (wx + WM_R)]);   // -disWgt / 81 (float / int -> float)
    const float d0 = __fsub_rn(p.x, q.x), d1 = __fsub_rn(p.y, q.y), d2 = __fsub_rn(p.z, q.z);
    float clrWgt = __fadd_rn(__fadd_rn(__fmul_rn(d0, d0), __fmul_rn(d1, d1)), __fmul_rn(d2, d2));
    if (RIGHT) clrWgt = wm_sqrtf(clrWgt);
    // clrWgt / (SIG_CLR*SIG_CLR): float / double -> double
    const double arg = __dsub_rn((double)sp, wm_div_sig_clr(clrWgt));
    return (float)psm_exp_nonpos(arg, wm_exp_tab);     // the host libm's exp, bit for bit (psm_exp.h): arg <= 0 or NaN
}

// one map byte through an agent-scope atomic load of the dword that holds it (coherent across the XCDs' L2s)
__device__ __forceinline__ int wm_load_byte(const uint8_t *p)
{
    const uintptr_t a = (uintptr_t)p;
    const unsigned w = __hip_atomic_load((const unsigned *)(a & ~(uintptr_t)3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (int)((w >> ((a & 3) * 8)) & 0xffu);
}

template <bool RIGHT, int NB>   // NB = ceil(maxDis / 64) histogram bins per lane
__global__ __launch_bounds__(64) void k_wgt_median(uint8_t *dis, const float4 *__restrict__ g1, const int *__restrict__ nxt,
                                                  int *prog, int W, int H, int maxDis, int *err)
{
    __shared__ float2 taps[WM_TAPS + 3];     // (disparity as float bits, weight) in window raster order
    __shared__ float hist[64 * NB];
    const int y = blockIdx.x, lane = threadIdx.x;
    const int *myn = nxt + (size_t)y * (W + 1);
    uint8_t *drow = dis + (size_t)y * W;
    // window row of this lane (lanes 0..18) for the dependency test
    const int wy_l = lane - WM_R;
    const int qy_l = lane < WM_K ? (y + wy_l + H) % H : y;
    const bool earlier = lane < WM_K && qy_l < y;
    const int *qn = nxt + (size_t)qy_l * (W + 1);
    // tap t of this lane's k-th round: t = lane + 64 k  (k = 0..5, 361 taps)
    constexpr int WM_ROUNDS = (WM_TAPS + 63) / 64;
    int prc = 0;                             // last value of prog[qy_l] this lane has seen
    int x = myn[0];
    while (x < W) {
        // ---- 361 weights, 64 at a time: they depend on the colours only, so they are evaluated BEFORE the wait - on a
        // map whose invalid pixels chain (every pixel waiting for its left neighbour) the wait is the critical path ----
        const float4 p = g1[(size_t)y * W + x];
        float wgt[WM_ROUNDS];
        int tap_off[WM_ROUNDS];
#pragma unroll
        for (int k = 0; k < WM_ROUNDS; ++k) {
            const int t = min(lane + 64 * k, WM_TAPS - 1);
            const int wy = t / WM_K - WM_R, wx = t % WM_K - WM_R;
            const int qy = (y + wy + H) % H, qx = (x + wx + W) % W;
            tap_off[k] = qy * W + qx;
            wgt[k] = wm_weight<RIGHT>(p, g1[tap_off[k]], wx, wy);
        }
        // ---- wait until every earlier invalid pixel inside the window is final ----
        {
            const int lo = x - WM_R, hi = x + WM_R;
            int a0, b0, a1 = 1, b1 = 0;   // up to two column intervals (second one empty by default)
            if (W < WM_K) { a0 = 0; b0 = W - 1; }
            else if (lo < 0) { a0 = lo + W; b0 = W - 1; a1 = 0; b1 = hi; }
            else if (hi >= W) { a0 = lo; b0 = W - 1; a1 = 0; b1 = hi - W; }
            else { a0 = lo; b0 = hi; }
            unsigned spins = 0;
            auto clear = [&](int pr) {     // no unfinished invalid pixel of this lane's window row inside the column intervals
                const int s0 = max(a0, pr), s1 = max(a1, pr);
                return !(s0 <= b0 && qn[s0] <= b0) && !(s1 <= b1 && qn[s1] <= b1);
            };
            for (;;) {
                bool ok = true;
                if (earlier) {
                    ok = clear(prc);           // prog only grows: the value seen last time often already suffices (no round trip)
                    if (!ok) {
                        prc = __hip_atomic_load(&prog[qy_l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok = clear(prc);
                    }
                }
                if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 24)) {          // watchdog: never hang the device
                    if (lane == 0) atomicExch(err, 1);
                    break;
                }
            }
            // (a prog value, cached or fresh, was loaded after the map bytes it vouches for were stored, and the loop exit
            // depends on it: the map loads below are issued after it)
        }
        // ---- the window's current disparities; a pixel of disparity 0 does not vote (src/PP.cpp:167) ----
#pragma unroll
        for (int k = 0; k < WM_ROUNDS; ++k) {
            const int t = lane + 64 * k;
            const int dep = wm_load_byte(dis + tap_off[k]);
            if (t < WM_TAPS) taps[t] = make_float2(__int_as_float(dep), dep != 0 ? wgt[k] : 0.0f);
        }
        __syncthreads();
        // ---- histogram + total in window raster order (adding 0.0f for "does not vote" is the identity) ----
        float acc[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[j] = 0.0f;
        float tot = 0.0f;
#pragma unroll 19
        for (int t = 0; t < WM_TAPS; ++t) {
            const float2 tw = taps[t];
            const int dep = __float_as_int(tw.x);
            tot = __fadd_rn(tot, tw.y);
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[j] = __fadd_rn(acc[j], dep == lane + 64 * j ? tw.y : 0.0f);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) hist[lane + 64 * j] = acc[j];
        __syncthreads();
        // ---- threshold scan over the non-empty bins, ascending d (src/PP.cpp:184-192) ----
        const float half = __fdiv_rn(tot, 2.0f);
        float run = 0.0f;
        int filterDep = 0;
        bool found = false;
        if (run >= half) { filterDep = 0; found = true; }        // d = 0: sumWgt(0) >= halfWgt only when nobody voted
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            unsigned long long m = __builtin_amdgcn_ballot_w64(acc[j] != 0.0f && lane + 64 * j < maxDis);
            while (m && !found) {
                const int b = __builtin_ctzll(m);
                m &= m - 1;
                run = __fadd_rn(run, hist[b + 64 * j]);
                if (run >= half) { filterDep = b + 64 * j; found = true; }
            }
        }
        // (not found: the running sum never reaches half - only possible with NaN weights - filterDep stays 0 as in the reference)
        const int xn = myn[x + 1];
        if (lane == 0) {
            const int old = __float_as_int(taps[WM_TAPS / 2].x);      // the centre tap is this pixel's current value
            int pv = xn;
            if (old != filterDep) {
                const uintptr_t a = (uintptr_t)(drow + x);
                const unsigned r = __hip_atomic_fetch_xor((unsigned *)(a & ~(uintptr_t)3), (unsigned)(old ^ filterDep) << ((a & 3) * 8),
                                                          __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("" : "+v"(pv) : "v"(r));                 // prog is stored only after the XOR has returned
            }
            __hip_atomic_store(&prog[y], pv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        x = xn;
    }
}

// ------------------------------------------------------------------------------------------
// Parallel form of the same filter: sweeps to the fixed point.
// The in-place raster-order result s is the unique solution of  s[p] = f(s[q] for invalid q earlier than p in raster
// order, orig[q] otherwise)  (induction over the raster order).  Iterating  new[p] = f(cur[earlier], orig[later])  over all
// invalid pixels at once, and afterwards only over the pixels that have a pixel changed by the previous sweep among the
// earlier taps of their window, stops when a sweep changes nothing - at that fixed point, i.e. at the reference's map,
// bit for bit.  Measured on the Middlebury pairs: 10-17 sweeps, ~4 evaluations per invalid pixel, every sweep fully
// parallel (one wave per pixel) - instead of a dependency chain through every invalid pixel of the map.
// f is evaluated exactly as in k_wgt_median (same weights, same raster-order float sums, same threshold scan).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int wm_append(int *cnt, bool want)
{   // wave-aggregated list append: returns this lane's slot (only meaningful where want)
    const unsigned long long m = __builtin_amdgcn_ballot_w64(want);
    if (!m) return 0;
    const int before = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
    int base = 0;
    if (threadIdx.x % 64 == __builtin_ctzll(m)) base = atomicAdd(cnt, __builtin_popcountll(m));
    base = __builtin_amdgcn_readlane(base, __builtin_ctzll(m));
    return base + before;
}

// Start of a call, both maps per launch: the invalid pixels of a map as a list (any order; 16 pixels per thread, one atomic per
// wave), the input copy `orig` the later taps of a window are read from, and the per-pixel sweep state (stamp, the gather form's
// byte maps) zeroed - round 6: one launch where seeds, copies and fills were eight.
constexpr int WM_SEED_PER = 16;
__global__ __launch_bounds__(256) void k_wm_seed(WmPair pr, int HW, int nb)
{   // nb: length of the scratch planes (HW rounded up to whole blocks of 256 pixels)
    const WmSide &a = pr.s[blockIdx.y];
    const int lane = threadIdx.x & 63;
    const int base = (blockIdx.x * blockDim.x + threadIdx.x) * WM_SEED_PER;
    typedef unsigned wm_u32 __attribute__((aligned(1)));
    unsigned m = 0;                                   // bit k: pixel base + k is invalid
    uint4 dv = make_uint4(0u, 0u, 0u, 0u);
    if (base + WM_SEED_PER <= HW) {
        const wm_u32 *pv = reinterpret_cast<const wm_u32 *>(a.valid + base), *pd = reinterpret_cast<const wm_u32 *>(a.cur + base);
        unsigned v[4] = {pv[0], pv[1], pv[2], pv[3]};
        dv = make_uint4(pd[0], pd[1], pd[2], pd[3]);
#pragma unroll
        for (int k = 0; k < WM_SEED_PER; ++k)
            if (((v[k >> 2] >> (8 * (k & 3))) & 0xffu) == 0u) m |= 1u << k;
    } else {
        unsigned d[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int k = 0; k < WM_SEED_PER; ++k)
            if (base + k < HW) {
                if (a.valid[base + k] == 0) m |= 1u << k;
                d[k >> 2] |= (unsigned)a.cur[base + k] << (8 * (k & 3));
            }
        dv = make_uint4(d[0], d[1], d[2], d[3]);
    }
    if (base < nb) {
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
        *reinterpret_cast<uint4 *>(a.orig + base) = dv;
        *reinterpret_cast<uint4 *>(a.chgb + base) = z;
        *reinterpret_cast<uint4 *>(a.rowany + base) = z;
    }
    {   // the wave's 1 024 stamps, a contiguous kilobyte per store (16 bytes a lane at a 64-byte stride reach the L2 as partial lines)
        const int wbase = base - lane * WM_SEED_PER;
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int k = 0; k < WM_SEED_PER / 4; ++k) {
            const int i0 = wbase + 4 * (k * 64 + lane);
            if (i0 < nb) *reinterpret_cast<uint4 *>(a.stamp + i0) = z;
        }
    }
    const int c = __builtin_popcount(m);
    int incl = c;                                     // inclusive prefix sum over the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    const int total = __shfl(incl, 63);
    if (total == 0) return;
    int wbase = 0;
    if (lane == 63) wbase = atomicAdd(a.cnt, total);
    wbase = __shfl(wbase, 63);
    int slot = wbase + incl - c;
    while (m) {
        const int k = __builtin_ctz(m);
        m &= m - 1;
        a.inv[slot++] = base + k;
    }
}

// One evaluation of f per LANE (round 3; round 2 spent a whole wave per pixel: every lane scanned all 361 taps for "its"
// bins - 361 x 64 lane-steps for 722 useful additions).  A lane walks the 361 taps of its own pixel in window raster order:
// weight (fp32 colour distance op for op, double exp narrowed to float - or read from the cache below), tot += w,
// hist[dep] += w in a private column of an LDS histogram laid out [bin][lane] (bank = lane: conflict free) - every float sum is
// formed in exactly the reference's order (src/PP.cpp:164-192), a bin sees its taps in raster order.  Then the threshold scan
// over the bins some lane touched, in ascending d (adding an empty bin is the identity), which also zeroes them again.
// (Round 4 tried a compact histogram - WM_RB = 32 bins from the window's smallest voting disparity, found by a first pass over the
// disparities, 8 KB of LDS and 16 waves per CU instead of 2, the 13 % of pixels with a wider window handed to this full-size form:
// bit-exact, and only 1.5 x faster on the 87 % it takes (265 vs 400 us for the first sweep of a 1080p map) - the extra pass, the
// hand-over launches and maps with random disparities (every window wide: 9.4 -> 12.6 ms) made it a net loss.  Not occupancy.)
// LDS: maxDis x 256 bytes per wave (64 KB at D = 256: two waves per CU) - the kernel runs at the pace ONE wave issues
// instructions, so what counts is the length of its instruction stream: no data-dependent branches per tap, window rows read
// as five dwords, the scan four bins at a time (1080p, 20 % invalid, random disparities: 24.9 -> 9.6 ms for both maps).
// The weights of a pixel depend on the image only, and the sweeps evaluate an invalid pixel ~5 times: k_wm_weights forms the
// 19 x 19 weights of every invalid pixel once, at full occupancy (this kernel runs two waves per CU at D = 256: its LDS
// histograms) - rows of 20 floats, pixel-major, slot = position in the list of invalid pixels, slot_of[pix] remembers it -
// and every evaluation (CACHED) loads them: five float4 per window row instead of 19 g1 loads, 19 colour distances,
// divisions and double-precision exps.  !CACHED: weights formed here (few invalid pixels, or the cache would not fit).
// Cache layout (round 4): blocks of 64 consecutive invalid pixels, structure of arrays - float4 index
//   wm_widx(slot, row, j) = ((slot / 64) * 19 + row) * 5 * 64 + j * 64 + slot % 64        (j-th float4 of the 20-float window row)
// so that 64 lanes working on 64 consecutive pixels of the list (the weights kernel, and the first sweep's lane-per-pixel
// evaluation, whose list IS the invalid list) read and write whole contiguous kilobytes.  The list is in raster order within
// blocks of 1 024 pixels (k_wm_seed), and invalid pixels come in runs along x: consecutive lanes are mostly consecutive x of one
// image row, so the 19 g1 loads of a window row coalesce as well.  (Round 3: one thread per (pixel, window row), 80-byte rows
// pixel-major - every load and store instruction walked 64 cache lines: 0.47 ms per 1080p map at 17 % invalid, bound by the
// address path, not by the double-precision exp.)
__device__ __forceinline__ size_t wm_widx(int slot, int row, int j)
{
    return ((size_t)(slot >> 6) * WM_K + row) * (WM_WROW / 4) * 64 + (size_t)j * 64 + (slot & 63);
}

template <bool RIGHT>
__global__ __launch_bounds__(256) void k_wm_weights(const float4 *__restrict__ g1, const int *__restrict__ inv, const int *n_inv,
                                                   float4 *__restrict__ wts, int *__restrict__ slot_of, int W, int H)
{   // one thread per (window row, invalid pixel): a wave = 64 consecutive pixels of the list, one window row
    const int n = *n_inv;
    const long long nblk = (n + 63) / 64, total = nblk * WM_K * 64;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(t & 63);
        const long long q = t >> 6;
        const int r = (int)(q % WM_K), wy = r - WM_R;
        const int slot = (int)(q / WM_K) * 64 + lane;
        if (slot >= n) continue;
        const int pix = inv[slot];
        const int y = pix / W, x = pix - y * W;
        if (r == 0) slot_of[pix] = slot;
        const float4 p = g1[pix];
        int qy = y + wy;
        qy = qy < 0 ? qy + H : (qy >= H ? qy - H : qy);
        float wk[WM_WROW];
#pragma unroll
        for (int k = 0; k < WM_K; ++k) {
            int qx = x + k - WM_R;
            qx = qx < 0 ? qx + W : (qx >= W ? qx - W : qx);
            wk[k] = wm_weight<RIGHT>(p, g1[qy * W + qx], k - WM_R, wy);
        }
        wk[WM_WROW - 1] = 0.0f;
#pragma unroll
        for (int k = 0; k < WM_WROW / 4; ++k) wts[wm_widx(slot, r, k)] = make_float4(wk[4 * k], wk[4 * k + 1], wk[4 * k + 2], wk[4 * k + 3]);
    }
}

template <bool RIGHT, bool CACHED>
__device__ __forceinline__ void wm_eval_lanes(float *hist, const uint8_t *cur, const uint8_t *__restrict__ orig,
                                              const float4 *__restrict__ g1, const int *__restrict__ list, const int *n_act,
                                              uint8_t *__restrict__ newv, int *__restrict__ chg, int *n_chg, int W, int H, int maxDis,
                                              const float4 *__restrict__ wts, const int *__restrict__ slot_of,
                                              uint8_t *curw, uint8_t *__restrict__ chgb, uint8_t *__restrict__ rowany, int mark)
{   // hist: [maxDis][64] floats of LDS, all zero between two evaluations
    const int lane = threadIdx.x;
    const int n = *n_act;
    if (n < WM_LANE_MIN) return;               // short lists: k_wm_eval_w (latency of one evaluation instead of a batch's)
    for (int d = 0; d < maxDis; ++d) hist[d * 64 + lane] = 0.0f;
    for (int i0 = blockIdx.x * 64; i0 < n; i0 += gridDim.x * 64) {
        const bool live = i0 + lane < n;
        const int pix = list[live ? i0 + lane : i0];
        const int y = pix / W, x = pix - y * W;
        const float4 p = g1[pix];
        // this pixel's row of the weight cache (WM_WROW / 4 float4 per window row)
        int wslot = 0;
        if constexpr (CACHED) wslot = slot_of[pix];
        unsigned dlo = (unsigned)maxDis, dhi = 0u;   // range of the bins this evaluation touches: dlo + 1 .. dhi (zeroed again after the scan)
        float tot = 0.0f;
        // One window row at a time: its loads (five dwords of disparities and five float4 of cached weights, or the 19 g1 values
        // the weights are formed from; every lane its own pixel: uncoalesced) are all issued before the row is accumulated, and
        // the next row's are in flight meanwhile.
        // P - 1 rows' loads are in flight ahead of the row being accumulated.  (Round 6 tried P = 4 with the cache - 25 registers a
        // row - on the theory that two waves per CU cannot hide a load: 6 % SLOWER at 1080p, 19 % at 450 x 375.  The kernel is
        // bound by the ~30 instructions a tap costs at one wave per SIMD, not by the loads' latency.)
        constexpr int P = 2;
        float4 gq[P][CACHED ? WM_WROW / 4 : WM_K];     // CACHED: the row's weights instead of its g1 values
        // the 19 disparities of a window row, packed four to a dword.  A row is 19 consecutive bytes of one plane (the current
        // iterate for an earlier row, the input for a later one; the pixel's own row: left of it / from it on) unless the window
        // wraps around the image's left or right edge: five unaligned dword loads instead of 19 byte loads - every lane reads
        // its own pixel's window, so each load instruction walks 64 cache lines whatever its width.
        typedef unsigned wm_u32 __attribute__((aligned(1)));
        unsigned dq[P][5];
        const bool inner = x - WM_R >= 0 && x + WM_R + 1 < W;     // (the fifth dword's last byte is still inside the row)
        auto issue_row = [&](int slot, int wy) __attribute__((always_inline)) {
            int qy = y + wy;
            qy = qy < 0 ? qy + H : (qy >= H ? qy - H : qy);
            const int rowo = qy * W;
            // raster order is index order: an earlier pixel shows its current iterate (= the input where it is valid),
            // a later one - and the pixel itself - the input (src/PP.cpp:164-166 reads the map in place)
            if (inner) {
                const int b0 = rowo + x - WM_R;
                const wm_u32 *pa = reinterpret_cast<const wm_u32 *>((qy <= y ? cur : orig) + b0);    // taps 0 .. 8 of the own row: current
                const wm_u32 *pb = reinterpret_cast<const wm_u32 *>((qy < y ? cur : orig) + b0);     // taps 9 .. 18 of the own row: input
                dq[slot][0] = pa[0]; dq[slot][1] = pa[1];
                dq[slot][2] = (pa[2] & 0xffu) | (pb[2] & 0xffffff00u);
                dq[slot][3] = pb[3]; dq[slot][4] = pb[4];
            } else {
#pragma unroll
                for (int j = 0; j < 5; ++j) dq[slot][j] = 0;
#pragma unroll
                for (int k = 0; k < WM_K; ++k) {
                    int qx = x + k - WM_R;
                    qx = qx < 0 ? qx + W : (qx >= W ? qx - W : qx);
                    const int off = rowo + qx;
                    dq[slot][k >> 2] |= (unsigned)(off < pix ? cur[off] : orig[off]) << (8 * (k & 3));
                }
            }
            if constexpr (!CACHED) {
#pragma unroll
                for (int k = 0; k < WM_K; ++k) {
                    int qx = x + k - WM_R;
                    qx = qx < 0 ? qx + W : (qx >= W ? qx - W : qx);
                    gq[slot][k] = g1[rowo + qx];
                }
            }
            if constexpr (CACHED) {
#pragma unroll
                for (int k = 0; k < WM_WROW / 4; ++k) gq[slot][k] = wts[wm_widx(wslot, wy + WM_R, k)];
            }
        };
        auto take_row = [&](int slot, int wy) __attribute__((always_inline)) {
            // (!CACHED: the 19 weights first - independent double-precision chains, division and exp, the scheduler can interleave)
            float wk[WM_WROW];
            if constexpr (CACHED) {
#pragma unroll
                for (int k = 0; k < WM_WROW / 4; ++k) { wk[4 * k] = gq[slot][k].x; wk[4 * k + 1] = gq[slot][k].y; wk[4 * k + 2] = gq[slot][k].z; wk[4 * k + 3] = gq[slot][k].w; }
            } else {
#pragma unroll
                for (int k = 0; k < WM_K; ++k) wk[k] = wm_weight<RIGHT>(p, gq[slot][k], k - WM_R, wy);
            }
            // Branch free: one wave per SIMD at best (the LDS histograms), so the kernel runs at the pace one wave issues
            // instructions, and the exec-mask bookkeeping of three data-dependent branches per tap (a run-length shortcut for
            // equal neighbours among them) was 2/3 of its instruction stream (PMC: 39 VALU + 39 scalar + 10 branch instructions per
            // tap).  A tap that does not vote (dep == 0, or dep >= maxDis: cannot come out of a WTA over maxDis slices) adds its
            // weight to bin 0, which nothing reads; adding +0.0f to the non-negative total is the identity.
            // Round 6: the taps of a dword of disparities (four; the row's last: three) read their bins TOGETHER and are then
            // added in tap order in registers - a tap whose bin an earlier tap of the group also hit continues from that tap's
            // result instead of what it read, and the results are stored in tap order (the last store to a bin is its sum).  Every
            // bin still sees its taps in raster order, one addition each: the same bits.  95 LDS round trips per pixel instead of
            // 361: -10 % at D = 64 (450 x 375, 43 % invalid: 1.51 -> 1.35 ms), nothing at D = 256, where a SIMD holds one wave and
            // the instruction count per tap binds (the compares cost what the waits did).
#pragma unroll
            for (int g = 0; g < (WM_K + 3) / 4; ++g) {
                constexpr int G = 4;
                float *hp[G];
                unsigned bin[G];
                float h[G], r[G];
#pragma unroll
                for (int j = 0; j < G; ++j) {
                    const int k = 4 * g + j;
                    if (k >= WM_K) continue;
                    const unsigned dep = (dq[slot][g] >> (8 * j)) & 0xffu;
                    tot = __fadd_rn(tot, dep != 0u ? wk[k] : 0.0f);
                    bin[j] = dep < (unsigned)maxDis ? dep : 0u;
                    // (ds_add_f32 gives the same bits and needs no read, but measured 40 % slower: the LDS processes a 64-lane float
                    // atomic far below the rate of a read and a write)
                    hp[j] = &hist[bin[j] * 64u + lane];
                    h[j] = *hp[j];
                    dlo = min(dlo, bin[j] - 1u);           // (bin 0: 0xffffffff, no effect)
                    dhi = max(dhi, bin[j]);
                }
#pragma unroll
                for (int j = 0; j < G; ++j) {
                    if (4 * g + j >= WM_K) continue;
                    float v = h[j];
#pragma unroll
                    for (int i = 0; i < j; ++i) v = bin[i] == bin[j] ? r[i] : v;      // (ascending: the latest earlier tap of the bin wins)
                    r[j] = __fadd_rn(v, wk[4 * g + j]);
                }
#pragma unroll
                for (int j = 0; j < G; ++j)
                    if (4 * g + j < WM_K) *hp[j] = r[j];
            }
        };
        // (19 rows = 9 P + P - 1)
#pragma unroll
        for (int j = 0; j < P - 1; ++j) issue_row(j, -WM_R + j);
        for (int wy = -WM_R; wy + 2 * P - 2 <= WM_R; wy += P) {
#pragma unroll
            for (int j = 0; j < P; ++j) {
                issue_row((j + P - 1) % P, wy + j + P - 1);
                take_row(j, wy + j);
            }
        }
#pragma unroll
        for (int j = 0; j < P - 1; ++j) take_row(j, WM_R - (P - 2) + j);
        // ---- threshold scan over the non-empty bins, ascending d (src/PP.cpp:184-192) ----
        const float half = __fdiv_rn(tot, 2.0f);
        float run = 0.0f;
        int filterDep = 0;
        bool found = run >= half;             // d = 0: sumWgt(0) >= halfWgt only when nobody voted
        // (only the bins some lane of the wave touched: the others are zero, and adding an empty bin is the identity)
        int wlo = (int)dlo + 1, whi = (int)dhi;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { wlo = min(wlo, __shfl_xor(wlo, o)); whi = max(whi, __shfl_xor(whi, o)); }
        // (branch free, four bins per iteration with their loads issued together: while !found, run < half, so an empty bin -
        // adding +0.0f - can neither change run nor end the scan)
        for (int d0 = wlo; d0 <= whi; d0 += 4) {
            float h[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) h[j] = d0 + j <= whi ? hist[(d0 + j) * 64 + lane] : 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (d0 + j <= whi) hist[(d0 + j) * 64 + lane] = 0.0f;
                run = __fadd_rn(run, h[j]);
                const bool hit = !found && run >= half;
                filterDep = hit ? d0 + j : filterDep;
                found = found || hit;
            }
        }
        // (not found: the running sum never reaches half - only possible with NaN weights - filterDep stays 0 as in the reference)
        const bool changed = live && filterDep != (int)cur[pix];
        const int slot = wm_append(n_chg, changed);
        if (changed) {
            // (round 6) the pixel takes its value at once - as in the wave form's tail sweeps, a lane still evaluating may read
            // either value, and whoever reads it is evaluated again next sweep - and leaves the marks of the gather form (itself
            // and the 19 columns around it in its row) here, where rounds 3 - 5 had k_wm_apply walk the list of changes for it
            // (66 us for the first sweep's 38 000 changes of the 1080p bench pair).  The scatter form still finds its list in chg.
            newv[pix] = (uint8_t)filterDep;
            chg[slot] = pix;
            curw[pix] = (uint8_t)filterDep;
            chgb[pix] = (uint8_t)mark;
#pragma unroll
            for (int wx = -WM_R; wx <= WM_R; ++wx) rowany[y * W + ((x + wx) % W + W) % W] = (uint8_t)mark;
        }
    }
}

// sweep `sw` of both maps (blockIdx.y); the cached form does not depend on the map (the weights kernel did)
template <bool CACHED>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2))) void k_wm_eval(WmPair pr, int sw, int W, int H, int maxDis)
{
    extern __shared__ float hist[];
    const WmSide &a = pr.s[blockIdx.y];
    const int *act = sw ? a.list[(sw + 1) & 1] : a.inv;
    if (CACHED || blockIdx.y == 0)
        wm_eval_lanes<false, CACHED>(hist, a.cur, a.orig, a.g1, act, a.cnt + 2 * sw, a.newv, a.chg, a.cnt + 2 * sw + 1, W, H, maxDis, (const float4 *)a.wts, a.slot_of,
                                     a.cur, a.chgb, a.rowany, sw + 1);
    else
        wm_eval_lanes<true, CACHED>(hist, a.cur, a.orig, a.g1, act, a.cnt + 2 * sw, a.newv, a.chg, a.cnt + 2 * sw + 1, W, H, maxDis, (const float4 *)a.wts, a.slot_of,
                                    a.cur, a.chgb, a.rowany, sw + 1);
}

// Short active lists (the tail sweeps: a few hundred pixels whose evaluation latency is what the sweep costs): one WAVE per
// pixel - the 361 weights of a pixel are evaluated by the 64 lanes in parallel, the histogram bins and the total are then
// accumulated in window raster order (lane l owns bins l, l+64, ..: every lane scans the (disparity, weight) pairs from LDS
// and adds the ones that fall into its bins).  ~10 us per evaluation instead of ~80 us for a batch of 64, at 30x the
// instructions per evaluation.
template <bool RIGHT, int NB, bool CACHED>
__device__ __forceinline__ void wm_eval_wave(const uint8_t *cur, const uint8_t *__restrict__ orig,
                                             const float4 *__restrict__ g1, const int *__restrict__ list, const int *n_act,
                                             uint8_t *__restrict__ newv, int *__restrict__ chg, int *n_chg, int W, int H, int maxDis,
                                             const float *__restrict__ wts, const int *__restrict__ slot_of, int force, int wg, int nwg,
                                             uint8_t *curw, const uint8_t *__restrict__ valid, int *__restrict__ stamp, int mark,
                                             int *__restrict__ next, int *n_next)
{   // wg of nwg: this wave's place among the waves that share the list
    // curw != null (the tail sweeps, round 6): a pixel that changes takes its new value AT ONCE and its wave queues the later
    // invalid pixels of its window for the next sweep itself (what k_wm_apply does after the evaluations otherwise) - one launch a
    // sweep.  Waves still evaluating may see the old value or the new one: either way the pixels that read it are evaluated again
    // in the next sweep (they are exactly the ones queued here), and a sweep that changes nothing has read a map nobody wrote -
    // the fixed point, which is unique (psm_api_pp.cpp).  Only the NUMBER of evaluations can differ from run to run.
    // NB == 1: taps in window raster order.  NB > 1: the voting taps stably partitioned by dep / 64 (bucket j holds, in raster
    // order, the taps of the bins lane + 64 j, buckets back to back), so a lane walks every tap once instead of NB times;
    // wsum[] keeps the raster order for the total.
    __shared__ float2 taps[WM_TAPS + 3];
    __shared__ float wsum[NB == 1 ? 1 : WM_TAPS + 3];
    __shared__ float hist[64 * NB];
    const int lane = threadIdx.x;
    const int n = *n_act;
    if (!force && n >= WM_LANE_MIN) return;    // long lists: k_wm_eval (force: a tail sweep, launched without it)
    constexpr int WM_ROUNDS = (WM_TAPS + 63) / 64;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int i = wg; i < n; i += nwg) {
        const int pix = list[i];
        const int y = pix / W, x = pix - y * W;
        const float4 p = g1[pix];
        const int before = cur[pix];        // (read with the taps: nothing changes the map during an evaluation pass)
        int cntj[NB], basej[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) cntj[j] = 0;
        int dep[WM_ROUNDS], off[WM_ROUNDS];
#pragma unroll
        for (int k = 0; k < WM_ROUNDS; ++k) {
            const int t = min(lane + 64 * k, WM_TAPS - 1);
            const int wy = t / WM_K - WM_R, wx = t % WM_K - WM_R;
            const int qy = (y + wy + H) % H, qx = (x + wx + W) % W;
            off[k] = qy * W + qx;
            // raster order is index order: an earlier pixel shows its current iterate (= the input where it is valid),
            // a later one - and the pixel itself - the input (src/PP.cpp:164-166 reads the map in place)
            dep[k] = off[k] < pix ? cur[off[k]] : orig[off[k]];
            if (NB > 1) {
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    cntj[j] += __builtin_popcountll(__builtin_amdgcn_ballot_w64(lane + 64 * k < WM_TAPS && dep[k] != 0 && (dep[k] >> 6) == j));
            }
        }
        if (NB > 1) {
            int b = 0;
#pragma unroll
            for (int j = 0; j < NB; ++j) { basej[j] = b; b += cntj[j]; }
        }
        int fill[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) fill[j] = 0;
#pragma unroll
        for (int k = 0; k < WM_ROUNDS; ++k) {
            const int t = min(lane + 64 * k, WM_TAPS - 1);
            const int wy = t / WM_K - WM_R, wx = t % WM_K - WM_R;
            float w;
            if constexpr (CACHED) {             // (the weights kernel stored it; float index inside the float4 array)
                const int k_ = wx + WM_R;
                w = wts[4 * wm_widx(slot_of[pix], wy + WM_R, k_ >> 2) + (k_ & 3)];
            }
            else w = wm_weight<RIGHT>(p, g1[off[k]], wx, wy);
            const bool live = lane + 64 * k < WM_TAPS;
            if (NB == 1) {
                if (live) taps[t] = make_float2(__int_as_float(dep[k]), dep[k] != 0 ? w : 0.0f);
            } else {
                if (live) wsum[t] = dep[k] != 0 ? w : 0.0f;
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const bool mine = live && dep[k] != 0 && (dep[k] >> 6) == j;
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(mine);
                    if (mine) taps[basej[j] + fill[j] + __builtin_popcountll(m & below)] = make_float2(__int_as_float(dep[k] & 63), w);
                    fill[j] += __builtin_popcountll(m);
                }
            }
        }
        __syncthreads();
        float acc[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[j] = 0.0f;
        float tot = 0.0f;
        if (NB == 1) {
#pragma unroll 19
            for (int t = 0; t < WM_TAPS; ++t) {
                const float2 tw = taps[t];
                tot = __fadd_rn(tot, tw.y);
                acc[0] = __fadd_rn(acc[0], __float_as_int(tw.x) == lane ? tw.y : 0.0f);
            }
        } else {
#pragma unroll 19
            for (int t = 0; t < WM_TAPS; ++t) tot = __fadd_rn(tot, wsum[t]);     // adding 0.0f for "does not vote" is the identity
#pragma unroll
            for (int j = 0; j < NB; ++j)
                for (int t = 0; t < cntj[j]; ++t) {
                    const float2 tw = taps[basej[j] + t];
                    acc[j] = __fadd_rn(acc[j], __float_as_int(tw.x) == lane ? tw.y : 0.0f);
                }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) hist[lane + 64 * j] = acc[j];
        __syncthreads();
        const float half = __fdiv_rn(tot, 2.0f);
        float run = 0.0f;
        int filterDep = 0;
        bool found = false;
        if (run >= half) { filterDep = 0; found = true; }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            unsigned long long m = __builtin_amdgcn_ballot_w64(acc[j] != 0.0f && lane + 64 * j < maxDis);
            while (m && !found) {
                const int b = __builtin_ctzll(m);
                m &= m - 1;
                run = __fadd_rn(run, hist[b + 64 * j]);
                if (run >= half) { filterDep = b + 64 * j; found = true; }
            }
        }
        if (curw) {
            if (filterDep != before) {                 // (wave-uniform: every lane ran the same scan)
                if (lane == 0) {
                    curw[pix] = (uint8_t)filterDep;
                    atomicAdd(n_chg, 1);
                }
#pragma unroll
                for (int k = 0; k < WM_ROUNDS; ++k) {
                    const int t = lane + 64 * k;
                    bool want = false;
                    int pp = 0;
                    if (t < WM_TAPS) {
                        const int wy = t / WM_K - WM_R, wx = t % WM_K - WM_R;
                        // the pixel whose tap (wy, wx) is this one: (py + wy + H) % H == y, (px + wx + W) % W == x
                        const int py = ((y - wy) % H + H) % H, px = ((x - wx) % W + W) % W;
                        pp = py * W + px;
                        want = pp > pix && valid[pp] == 0 && atomicExch(&stamp[pp], mark) != mark;
                    }
                    const int slot = wm_append(n_next, want);
                    if (want) next[slot] = pp;
                }
            }
        } else if (lane == 0 && filterDep != before) {
            newv[pix] = (uint8_t)filterDep;
            chg[atomicAdd(n_chg, 1)] = pix;
        }
        __syncthreads();
    }
}

template <int NB, bool CACHED>
__global__ __launch_bounds__(64) void k_wm_eval_w(WmPair pr, int sw, int W, int H, int maxDis, int force)
{
    const WmSide &a = pr.s[blockIdx.y];
    const int *act = sw ? a.list[(sw + 1) & 1] : a.inv;
    uint8_t *const curw = force ? a.cur : nullptr;       // (force = the tail sweeps: evaluation and dependents in this one launch)
    if (CACHED || blockIdx.y == 0)
        wm_eval_wave<false, NB, CACHED>(a.cur, a.orig, a.g1, act, a.cnt + 2 * sw, a.newv, a.chg, a.cnt + 2 * sw + 1, W, H, maxDis, a.wts, a.slot_of, force, blockIdx.x, gridDim.x,
                                        curw, a.valid, a.stamp, sw + 1, a.list[sw & 1], a.cnt + 2 * (sw + 1));
    else
        wm_eval_wave<true, NB, CACHED>(a.cur, a.orig, a.g1, act, a.cnt + 2 * sw, a.newv, a.chg, a.cnt + 2 * sw + 1, W, H, maxDis, a.wts, a.slot_of, force, blockIdx.x, gridDim.x,
                                       curw, a.valid, a.stamp, sw + 1, a.list[sw & 1], a.cnt + 2 * (sw + 1));
}

// Which pixels does the next sweep evaluate?  Every invalid pixel that has a pixel changed by this sweep among the EARLIER taps
// of its window.  Two forms, chosen on the device from the counts:
//   scatter (k_wm_apply, short change lists - the tail sweeps): one wave per changed pixel stamps its later invalid window
//     neighbours (an atomic exchange each: once per pixel and sweep);
//   gather (k_wm_apply marks, k_wm_gather collects; long change lists - the first sweeps, where scatter spent 36 atomics per changed pixel:
//     7.7 of 19 ms at 1080p / 20 % invalid): the changed pixels mark themselves and the 19 columns around them in their own
//     row (rowany[y][x] = "row y changed within x +- 9", plain byte stores of the sweep's mark), then every invalid pixel looks
//     at rowany of the earlier rows of its window at its own column and at the earlier taps of its own row: <= 36 byte loads.
// (measured at 1080p, both maps per launch: marks + gather pass ~0.07 ms whatever changed, the scatter pass ~12 ns per changed pixel)
__device__ __forceinline__ bool wm_gather_form(int n_chg, int n_inv) { return n_inv >= 4096 && (long long)n_chg * 80 >= n_inv; }

__global__ __launch_bounds__(256) void k_wm_gather(WmPair pr, int sw, int W, int H)
{
    const WmSide &a = pr.s[blockIdx.y];
    const int *__restrict__ inv = a.inv, *n_inv = a.cnt, *n_chg = a.cnt + 2 * sw + 1;
    const uint8_t *__restrict__ chgb = a.chgb, *__restrict__ rowany = a.rowany;
    const int mark = sw + 1;
    int *__restrict__ next = a.list[sw & 1], *n_next = a.cnt + 2 * (sw + 1);
    const int ninv = *n_inv;
    if (!wm_gather_form(*n_chg, ninv)) return;
    const int lane = threadIdx.x & 63;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i - lane < ninv; i += gridDim.x * blockDim.x) {   // (whole waves: wm_append)
        const bool live = i < ninv;
        const int pix = inv[live ? i : 0];
        const int y = pix / W, x = pix - y * W;
        bool want = false;
#pragma unroll
        for (int w = -WM_R; w <= WM_R; ++w) {
            if (w == 0) continue;
            const int qy = ((y + w) % H + H) % H, qx = ((x + w) % W + W) % W;
            if (qy < y) want |= rowany[qy * W + x] == (uint8_t)mark;      // an earlier row of the window (also through the wrap)
            if (qx < x) want |= chgb[y * W + qx] == (uint8_t)mark;        // an earlier tap of the pixel's own row
        }
        want = want && live;
        const int slot = wm_append(n_next, want);
        if (want) next[slot] = pix;
    }
}

// the changed pixels take their new value; every invalid pixel LATER in raster order that has one of them in its window
// is evaluated again in the next sweep (stamp: once)
__device__ __forceinline__ void wm_apply(uint8_t *__restrict__ cur, const uint8_t *__restrict__ newv, const uint8_t *__restrict__ valid,
                                         const int *__restrict__ chg, const int *n_chg, const int *n_inv, int *__restrict__ stamp, int mark,
                                         int *__restrict__ next, int *n_next, int W, int H, uint8_t *__restrict__ chgb, uint8_t *__restrict__ rowany, int force,
                                         int wg, int nwg, const int *n_act)
{
    const int lane = threadIdx.x;
    const int n = *n_chg;
    if (!force && wm_gather_form(n, *n_inv)) {   // gather form: k_wm_gather builds the next list from the marks the evaluations left
        if (*n_act >= WM_LANE_MIN) return;       // (k_wm_eval applied and marked its changes itself)
        for (int i = wg * 64 + lane; i < n; i += nwg * 64) {      // changes found by the wave form (a short list of a long map)
            const int pix = chg[i];
            const int y = pix / W, x = pix - y * W;
            cur[pix] = newv[pix];
            chgb[pix] = (uint8_t)mark;
#pragma unroll
            for (int wx = -WM_R; wx <= WM_R; ++wx) rowany[y * W + ((x + wx) % W + W) % W] = (uint8_t)mark;
        }
        return;
    }
    constexpr int WM_ROUNDS = (WM_TAPS + 63) / 64;
    for (int i = wg; i < n; i += nwg) {
        const int pix = chg[i];
        const int y = pix / W, x = pix - y * W;
        if (lane == 0) cur[pix] = newv[pix];
#pragma unroll
        for (int k = 0; k < WM_ROUNDS; ++k) {
            const int t = lane + 64 * k;
            bool want = false;
            int pp = 0;
            if (t < WM_TAPS) {
                const int wy = t / WM_K - WM_R, wx = t % WM_K - WM_R;
                // the pixel whose tap (wy, wx) is this one: (py + wy + H) % H == y, (px + wx + W) % W == x
                const int py = ((y - wy) % H + H) % H, px = ((x - wx) % W + W) % W;
                pp = py * W + px;
                want = pp > pix && valid[pp] == 0 && atomicExch(&stamp[pp], mark) != mark;
            }
            const int slot = wm_append(n_next, want);
            if (want) next[slot] = pp;
        }
    }
}

__global__ __launch_bounds__(64) void k_wm_apply(WmPair pr, int sw, int W, int H, int force)
{
    const WmSide &a = pr.s[blockIdx.y];
    wm_apply(a.cur, a.newv, a.valid, a.chg, a.cnt + 2 * sw + 1, a.cnt, a.stamp, sw + 1, a.list[sw & 1], a.cnt + 2 * (sw + 1), W, H, a.chgb, a.rowany, force,
             blockIdx.x, gridDim.x, a.cnt + 2 * sw);
}

// the 19 x 19 weights of the n_inv invalid pixels of `inv` (n = an upper bound of *n_inv, for the grid) -> wts, slot_of
void launch_wm_weights(hipStream_t s, const float4 *g1, int W, int H, int right, const int *inv, const int *n_inv, int n, float *wts, int *slot_of)
{
    const long long threads = (long long)((n + 63) / 64) * WM_K * 64;
    const int blocks = (int)((threads + 255) / 256 < 65536 ? (threads + 255) / 256 : 65536);
    if (right) hipLaunchKernelGGL(k_wm_weights<true>, dim3(blocks), dim3(256), 0, s, g1, inv, n_inv, (float4 *)wts, slot_of, W, H);
    else hipLaunchKernelGGL(k_wm_weights<false>, dim3(blocks), dim3(256), 0, s, g1, inv, n_inv, (float4 *)wts, slot_of, W, H);
}

void launch_wm_seed(hipStream_t s, const WmPair &p, int W, int H)
{   // (the counters of both maps are zero: the caller's memset)
    const int HW = W * H, per_block = 256 * WM_SEED_PER;
    hipLaunchKernelGGL(k_wm_seed, dim3((HW + per_block - 1) / per_block, 2), dim3(256), 0, s, p, HW, (HW + 255) / 256 * 256);
}

// sweep `sw` of both maps: evaluate the active lists -> changed pixels -> applied, dependents -> the next lists (WmSide)
void launch_wm_sweep(hipStream_t s, const WmPair &p, int W, int H, int maxDis, int sw, bool cached, bool tail)
{   // tail: the host has seen short lists going into this sweep on both maps - only the one-wave-per-pixel evaluation is launched
    // (made to take whatever the lists turn out to be), and it applies its changes and queues their dependents itself: one launch
    // instead of four.  cached: both maps have their weight cache (launch_wm_weights).
    const dim3 ga(2048, 2);
    // two waves per CU at D = 256 (64 KB of LDS each), up to ten at D <= 64; the grid strides over batches of 64 pixels
    const size_t lds = (size_t)maxDis * 64 * sizeof(float);
    const PcDev dev = pc_dev();
    const int per_cu = (int)(160 * 1024 / (lds > 16384 ? lds : 16384));
    const dim3 ge(dev.nxcd * dev.cus_per_xcd * (per_cu < 1 ? 1 : per_cu), 2);
    const int force = tail ? 1 : 0;
    if (!tail) {
        if (cached) hipLaunchKernelGGL(k_wm_eval<true>, ge, dim3(64), lds, s, p, sw, W, H, maxDis);
        else hipLaunchKernelGGL(k_wm_eval<false>, ge, dim3(64), lds, s, p, sw, W, H, maxDis);
    }
    // ... and the one-wave-per-pixel form for short lists (either kernel returns at once when a list is not its size)
    const int nb = (maxDis + 63) / 64;
    const dim3 gw(8192, 2);
#define PSM_LAUNCH_WE(NBV, CA) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wm_eval_w<NBV, CA>), gw, dim3(64), 0, s, p, sw, W, H, maxDis, force)
#define PSM_LAUNCH_WE2(NBV) { if (cached) PSM_LAUNCH_WE(NBV, true); else PSM_LAUNCH_WE(NBV, false); }
    if (nb <= 1) PSM_LAUNCH_WE2(1) else if (nb == 2) PSM_LAUNCH_WE2(2) else if (nb == 3) PSM_LAUNCH_WE2(3) else PSM_LAUNCH_WE2(4)
#undef PSM_LAUNCH_WE2
#undef PSM_LAUNCH_WE
    if (tail) return;               // (the forced wave form applied its changes and queued their dependents itself)
    hipLaunchKernelGGL(k_wm_apply, ga, dim3(64), 0, s, p, sw, W, H, force);
    hipLaunchKernelGGL(k_wm_gather, dim3(2048, 2), dim3(256), 0, s, p, sw, W, H);
}

void launch_wgt_median(hipStream_t s, uint8_t *dis, const uint8_t *valid, const float4 *g1, int W, int H, int maxDis, int right,
                       int *nxt, int *prog, int *err)
{
    hipLaunchKernelGGL(k_wm_next, dim3(H), dim3(64), 0, s, valid, W, nxt, prog);
    const int nb = (maxDis + 63) / 64;
#define PSM_LAUNCH_WM(R, NBV) \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wgt_median<R, NBV>), dim3(H), dim3(64), 0, s, dis, g1, (const int *)nxt, prog, W, H, maxDis, err)
    if (right) {
        if (nb <= 1) PSM_LAUNCH_WM(true, 1); else if (nb == 2) PSM_LAUNCH_WM(true, 2); else if (nb == 3) PSM_LAUNCH_WM(true, 3); else PSM_LAUNCH_WM(true, 4);
    } else {
        if (nb <= 1) PSM_LAUNCH_WM(false, 1); else if (nb == 2) PSM_LAUNCH_WM(false, 2); else if (nb == 3) PSM_LAUNCH_WM(false, 3); else PSM_LAUNCH_WM(false, 4);
    }
#undef PSM_LAUNCH_WM
}

}  // namespace psm

// debug / test hook (not in include/primesm_hip.h): wgtMedian's bilateral weight for n operand tuples on the device - pq: n x 8
// floats {p.xyz, -, q.xyz, -}, wxy: n x 2 ints, out: n floats (host pointers).  tests/test_gpu_pp_ocv.py compares them bit for bit
// with the host's (roots through double, exp as glibc forms it: the two places where device and host could part).
namespace psm {
template <bool RIGHT>
__global__ __launch_bounds__(256) void k_wm_weight_probe(const float4 *pq, const int2 *wxy, int n, float *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = wm_weight<RIGHT>(pq[2 * i], pq[2 * i + 1], wxy[i].x, wxy[i].y);
}
}  // namespace psm
extern "C" int psm_debug_wm_weights(const float *pq, const int *wxy, int n, int right, float *out)
{
    float4 *dpq = nullptr; int2 *dw = nullptr; float *dout = nullptr;
    int rc = 1;
    if (hipMalloc((void **)&dpq, (size_t)n * 32) == hipSuccess && hipMalloc((void **)&dw, (size_t)n * 8) == hipSuccess &&
        hipMalloc((void **)&dout, (size_t)n * 4) == hipSuccess &&
        hipMemcpy(dpq, pq, (size_t)n * 32, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(dw, wxy, (size_t)n * 8, hipMemcpyHostToDevice) == hipSuccess) {
        if (right) hipLaunchKernelGGL(psm::k_wm_weight_probe<true>, dim3((n + 255) / 256), dim3(256), 0, 0, dpq, dw, n, dout);
        else hipLaunchKernelGGL(psm::k_wm_weight_probe<false>, dim3((n + 255) / 256), dim3(256), 0, 0, dpq, dw, n, dout);
        rc = hipMemcpy(out, dout, (size_t)n * 4, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
    }
    (void)hipFree(dpq); (void)hipFree(dw); (void)hipFree(dout);
    return rc;
}
