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

namespace psm {

constexpr int WM_R = 9;                 // MED_SZ / 2, include/PP.h:12
constexpr int WM_K = 2 * WM_R + 1;      // 19
constexpr int WM_TAPS = WM_K * WM_K;    // 361
constexpr int WM_LANE_MIN = 8192;       // active pixels from which a sweep evaluates one pixel per LANE (k_wm_eval) instead of per wave

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

template <bool RIGHT>
__device__ __forceinline__ float wm_weight(float4 p, float4 q, int wx, int wy)
{
    // src/PP.cpp:169-175 (left) / 216-224 (right)
    float disWgt = (float)(wx * wx + wy * wy);
    const float d0 = __fsub_rn(p.x, q.x), d1 = __fsub_rn(p.y, q.y), d2 = __fsub_rn(p.z, q.z);
    float clrWgt = __fadd_rn(__fadd_rn(__fmul_rn(d0, d0), __fmul_rn(d1, d1)), __fmul_rn(d2, d2));
    if (RIGHT) {
        disWgt = __fsqrt_rn(disWgt);
        clrWgt = __fsqrt_rn(clrWgt);
    }
    // -disWgt / (SIG_DIS*SIG_DIS): float / int -> float;  clrWgt / (SIG_CLR*SIG_CLR): float / double -> double
    const double arg = __dsub_rn((double)__fdiv_rn(-disWgt, 81.0f), __ddiv_rn((double)clrWgt, 0.1 * 0.1));
    return (float)exp(arg);
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

__global__ __launch_bounds__(256) void k_wm_seed(const uint8_t *__restrict__ valid, int HW, int *__restrict__ list, int *cnt)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool inv = i < HW && valid[i] == 0;
    const int slot = wm_append(cnt, inv);
    if (inv) list[slot] = i;
}

// One evaluation of f per LANE (round 3; round 2 spent a whole wave per pixel: every lane scanned all 361 taps for "its"
// bins - 361 x 64 lane-steps for 722 useful additions).  A lane walks the 361 taps of its own pixel in window raster order:
// weight (fp32 colour distance op for op, double exp narrowed to float), tot += w, hist[dep] += w in a private column of an
// LDS histogram laid out [bin][lane] (bank = lane: conflict free) - every float sum is formed in exactly the reference's order
// (src/PP.cpp:164-192), a bin sees its taps in raster order.  Runs of equal disparities (the common case in a disparity map)
// stay in a register; the LDS column is touched only when the disparity changes.  Then the threshold scan over the bins in
// ascending d (adding an empty bin is the identity).  LDS: maxDis x 256 bytes per wave (64 KB at D = 256: two waves per CU).
template <bool RIGHT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2))) void k_wm_eval(const uint8_t *__restrict__ cur, const uint8_t *__restrict__ orig,
                                               const float4 *__restrict__ g1, const int *__restrict__ list, const int *n_act,
                                               uint8_t *__restrict__ newv, int *__restrict__ chg, int *n_chg, int W, int H, int maxDis)
{
    extern __shared__ float hist[];          // [maxDis][64]
    const int lane = threadIdx.x;
    const int n = *n_act;
    if (n < WM_LANE_MIN) return;               // short lists: k_wm_eval_w (latency of one evaluation instead of a batch's)
    for (int i0 = blockIdx.x * 64; i0 < n; i0 += gridDim.x * 64) {
        const bool live = i0 + lane < n;
        const int pix = list[live ? i0 + lane : i0];
        const int y = pix / W, x = pix - y * W;
        const float4 p = g1[pix];
        for (int d = 0; d < maxDis; ++d) hist[d * 64 + lane] = 0.0f;
        float tot = 0.0f;
        int run_d = 0;                        // disparity of the current run (0: none) and its bin's sum so far
        float run_s = 0.0f;
        // One window row at a time: its 19 + 19 loads (every lane its own pixel: uncoalesced, latency bound) are all issued
        // before the first weight is formed, and the next row's are in flight while this row is accumulated.
        float4 gq[2][WM_K];
        int dq[2][WM_K];
        auto issue_row = [&](int slot, int wy) __attribute__((always_inline)) {
            int qy = y + wy;
            qy = qy < 0 ? qy + H : (qy >= H ? qy - H : qy);
            const int rowo = qy * W;
#pragma unroll
            for (int k = 0; k < WM_K; ++k) {
                int qx = x + k - WM_R;
                qx = qx < 0 ? qx + W : (qx >= W ? qx - W : qx);
                const int off = rowo + qx;
                // raster order is index order: an earlier pixel shows its current iterate (= the input where it is valid),
                // a later one - and the pixel itself - the input (src/PP.cpp:164-166 reads the map in place)
                dq[slot][k] = off < pix ? cur[off] : orig[off];
                gq[slot][k] = g1[off];
            }
        };
        auto take_row = [&](int slot, int wy) __attribute__((always_inline)) {
            // (the 19 weights first: independent double-precision chains - division, exp - the scheduler can interleave; with
            // one or two waves per SIMD a single chain would run at its own latency)
            float wk[WM_K];
#pragma unroll
            for (int k = 0; k < WM_K; ++k) wk[k] = wm_weight<RIGHT>(p, gq[slot][k], k - WM_R, wy);
#pragma unroll
            for (int k = 0; k < WM_K; ++k) {
                const int dep = dq[slot][k];
                const float w = wk[k];
                if (dep != 0) tot = __fadd_rn(tot, w);
                if (dep != 0 && dep < maxDis) {            // (dep >= maxDis cannot come out of a WTA over maxDis slices: no bin)
                    if (dep != run_d) {
                        if (run_d != 0) hist[run_d * 64 + lane] = run_s;
                        run_s = hist[dep * 64 + lane];
                        run_d = dep;
                    }
                    run_s = __fadd_rn(run_s, w);
                }
            }
        };
        issue_row(0, -WM_R);
        for (int wy = -WM_R; wy < WM_R; wy += 2) {     // rows wy (slot 0) and wy + 1 (slot 1); the last row after the loop
            issue_row(1, wy + 1);
            take_row(0, wy);
            issue_row(0, wy + 2);
            take_row(1, wy + 1);
        }
        take_row(0, WM_R);
        if (run_d != 0) hist[run_d * 64 + lane] = run_s;
        // ---- threshold scan over the non-empty bins, ascending d (src/PP.cpp:184-192) ----
        const float half = __fdiv_rn(tot, 2.0f);
        float run = 0.0f;
        int filterDep = 0;
        bool found = run >= half;             // d = 0: sumWgt(0) >= halfWgt only when nobody voted
        for (int d = 1; d < maxDis; ++d) {
            const float h = hist[d * 64 + lane];
            if (!found && h != 0.0f) {
                run = __fadd_rn(run, h);
                if (run >= half) { filterDep = d; found = true; }
            }
        }
        // (not found: the running sum never reaches half - only possible with NaN weights - filterDep stays 0 as in the reference)
        const bool changed = live && filterDep != (int)cur[pix];
        const int slot = wm_append(n_chg, changed);
        if (changed) {
            newv[pix] = (uint8_t)filterDep;
            chg[slot] = pix;
        }
    }
}

// Short active lists (the tail sweeps: a few hundred pixels whose evaluation latency is what the sweep costs): one WAVE per
// pixel - the 361 weights of a pixel are evaluated by the 64 lanes in parallel, the histogram bins and the total are then
// accumulated in window raster order (lane l owns bins l, l+64, ..: every lane scans the (disparity, weight) pairs from LDS
// and adds the ones that fall into its bins).  ~10 us per evaluation instead of ~80 us for a batch of 64, at 30x the
// instructions per evaluation.
template <bool RIGHT, int NB>
__global__ __launch_bounds__(64) void k_wm_eval_w(const uint8_t *__restrict__ cur, const uint8_t *__restrict__ orig,
                                               const float4 *__restrict__ g1, const int *__restrict__ list, const int *n_act,
                                               uint8_t *__restrict__ newv, int *__restrict__ chg, int *n_chg, int W, int H, int maxDis)
{
    // NB == 1: taps in window raster order.  NB > 1: the voting taps stably partitioned by dep / 64 (bucket j holds, in raster
    // order, the taps of the bins lane + 64 j, buckets back to back), so a lane walks every tap once instead of NB times;
    // wsum[] keeps the raster order for the total.
    __shared__ float2 taps[WM_TAPS + 3];
    __shared__ float wsum[NB == 1 ? 1 : WM_TAPS + 3];
    __shared__ float hist[64 * NB];
    const int lane = threadIdx.x;
    const int n = *n_act;
    if (n >= WM_LANE_MIN) return;              // long lists: k_wm_eval
    constexpr int WM_ROUNDS = (WM_TAPS + 63) / 64;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const int pix = list[i];
        const int y = pix / W, x = pix - y * W;
        const float4 p = g1[pix];
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
            const float w = wm_weight<RIGHT>(p, g1[off[k]], wx, wy);
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
        if (lane == 0 && filterDep != (int)cur[pix]) {
            newv[pix] = (uint8_t)filterDep;
            chg[atomicAdd(n_chg, 1)] = pix;
        }
        __syncthreads();
    }
}

// the changed pixels take their new value; every invalid pixel LATER in raster order that has one of them in its window
// is evaluated again in the next sweep (stamp: once)
__global__ __launch_bounds__(64) void k_wm_apply(uint8_t *__restrict__ cur, const uint8_t *__restrict__ newv, const uint8_t *__restrict__ valid,
                                                const int *__restrict__ chg, const int *n_chg, int *__restrict__ stamp, int mark,
                                                int *__restrict__ next, int *n_next, int W, int H)
{
    const int lane = threadIdx.x;
    const int n = *n_chg;
    constexpr int WM_ROUNDS = (WM_TAPS + 63) / 64;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
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

void launch_wm_seed(hipStream_t s, const uint8_t *valid, int W, int H, int *list, int *cnt)
{
    const int HW = W * H;
    hipLaunchKernelGGL(k_wm_seed, dim3((HW + 255) / 256), dim3(256), 0, s, valid, HW, list, cnt);
}

// one sweep: evaluate list `act` (count *n_act) -> changed pixels (chg, *n_chg) -> applied, dependents -> list `next` (*n_next)
void launch_wm_sweep(hipStream_t s, uint8_t *cur, const uint8_t *orig, const uint8_t *valid, const float4 *g1, int W, int H, int maxDis,
                     int right, const int *act, const int *n_act, uint8_t *newv, int *chg, int *n_chg, int *stamp, int mark,
                     int *next, int *n_next)
{
    const dim3 ga(2048);
    // two waves per CU at D = 256 (64 KB of LDS each), up to ten at D <= 64; the grid strides over batches of 64 pixels
    const size_t lds = (size_t)maxDis * 64 * sizeof(float);
    const PcDev dev = pc_dev();
    const int per_cu = (int)(160 * 1024 / (lds > 16384 ? lds : 16384));
    const dim3 ge(dev.nxcd * dev.cus_per_xcd * (per_cu < 1 ? 1 : per_cu));
    if (right) hipLaunchKernelGGL(k_wm_eval<true>, ge, dim3(64), lds, s, (const uint8_t *)cur, orig, g1, act, n_act, newv, chg, n_chg, W, H, maxDis);
    else hipLaunchKernelGGL(k_wm_eval<false>, ge, dim3(64), lds, s, (const uint8_t *)cur, orig, g1, act, n_act, newv, chg, n_chg, W, H, maxDis);
    // ... and the one-wave-per-pixel form for short lists (either kernel returns at once when the list is not its size)
    const int nb = (maxDis + 63) / 64;
    const dim3 gw(8192);
#define PSM_LAUNCH_WE(R, NBV) \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wm_eval_w<R, NBV>), gw, dim3(64), 0, s, (const uint8_t *)cur, orig, g1, act, n_act, newv, chg, n_chg, W, H, maxDis)
    if (right) {
        if (nb <= 1) PSM_LAUNCH_WE(true, 1); else if (nb == 2) PSM_LAUNCH_WE(true, 2); else if (nb == 3) PSM_LAUNCH_WE(true, 3); else PSM_LAUNCH_WE(true, 4);
    } else {
        if (nb <= 1) PSM_LAUNCH_WE(false, 1); else if (nb == 2) PSM_LAUNCH_WE(false, 2); else if (nb == 3) PSM_LAUNCH_WE(false, 3); else PSM_LAUNCH_WE(false, 4);
    }
#undef PSM_LAUNCH_WE
    hipLaunchKernelGGL(k_wm_apply, ga, dim3(64), 0, s, cur, (const uint8_t *)newv, valid, (const int *)chg, (const int *)n_chg, stamp, mark, next, n_next, W, H);
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
