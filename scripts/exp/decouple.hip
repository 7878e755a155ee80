// Scratch experiment: does moving the stage-A stores to separate waves of the same workgroup restore
// overlap between the load/compute chain and the 16 B/voxel store stream?
#include "psm_kernels.hip"
#include <cstdio>
#include <vector>
using namespace psm;
#define KEEP(x) asm volatile("" ::"v"(x))

// MODE 0: normal (each wave loads, computes, stores); MODE 1: compute waves (0-3) never store, store waves (4-7)
// write the same rows (garbage data) with no loads; MODE 2: compute waves only (no stores at all); MODE 3: store waves only
template <int MODE>
__global__ __launch_bounds__(512) void k_dec(const float *__restrict__ vol, float4 *__restrict__ ab,
                                            const float4 *__restrict__ G1, const float4 *__restrict__ G2,
                                            const float4 *__restrict__ G3, const float2 *__restrict__ G4,
                                            int W, int H, int Dloc, int nstrips, int nsegs, int seg_rows, int nzg)
{
    const int wave8 = threadIdx.x >> 6;
    const bool storer = wave8 >= 4;
    if (MODE == 0 && storer) return;
    if (MODE == 2 && storer) return;
    if (MODE == 3 && !storer) return;
    // decode like march_pos order 0 but with wave & 3
    const int npairs = nstrips * nsegs, p8 = (npairs + 7) >> 3;
    int id = blockIdx.x, xcd = id & 7, j = id >> 3, zg = j % nzg, pl = j / nzg, pair = xcd * p8 + pl;
    if (!(pl < p8 && pair < npairs)) return;
    int strip = pair % nstrips, seg = pair / nstrips;
    int lane = threadIdx.x & 63, d = zg * 4 + (wave8 & 3);
    if (d >= Dloc) return;
    int x0 = strip * OUT_PER_WAVE, cs = r101c(x0 - 4 + lane, W), xo = x0 + lane;
    bool ovalid = lane < OUT_PER_WAVE && xo < W;
    int y0 = seg * seg_rows, y1 = min(H, y0 + seg_rows);
    const int i1 = ((lane + 1) & 63) << 2, i2 = ((lane + 2) & 63) << 2, i4 = ((lane + 4) & 63) << 2;
    const size_t HW = (size_t)H * W;
    const float *vd = vol + (size_t)d * HW;
    float4 *abd = ab + (size_t)d * HW;
    const int n = (y1 - y0) + 7, ybase = y0 - 4, xoc = min(xo, W - 1);
    if (storer) {
        float4 r = make_float4((float)lane, 1.f, 2.f, 3.f);
        for (int step = 7; step < n; ++step) {
            if (ovalid) abd[(size_t)(ybase + step - 3) * W + xo] = r;
            r.x += 1.f;
        }
        return;
    }
    VTree t0 = {}, t1 = {}, t2 = {}, t3 = {};
    float pin[4]; float4 gin[4], o2[4], o3[4]; float2 o4[4];
#define ISSUE(SLOT, STEP) { const size_t off_ = (size_t)r101c(ybase + (STEP), H) * W + cs; pin[SLOT] = vd[off_]; gin[SLOT] = G1[off_]; \
        int yo_ = ybase + (STEP) - 3; yo_ = yo_ < 0 ? 0 : (yo_ > H - 1 ? H - 1 : yo_); const size_t oo_ = (size_t)yo_ * W + xoc; \
        o2[SLOT] = G2[oo_]; o3[SLOT] = G3[oo_]; o4[SLOT] = G4[oo_]; }
    ISSUE(0, 0) __builtin_amdgcn_sched_barrier(0); ISSUE(1, 1) __builtin_amdgcn_sched_barrier(0); ISSUE(2, 2) __builtin_amdgcn_sched_barrier(0);
    for (int i = 0; i < n; i += 4) {
#define STEP(K) { const int step = i + K; ISSUE((K + 3) & 3, step + 3) const float p = pin[K]; \
        double h0 = hsum8(p, i1, i2, i4), h1 = hsum8(__fmul_rn(gin[K].x, p), i1, i2, i4), h2 = hsum8(__fmul_rn(gin[K].y, p), i1, i2, i4), h3 = hsum8(__fmul_rn(gin[K].z, p), i1, i2, i4); \
        double n0 = vstep<K>(t0, h0), n1 = vstep<K>(t1, h1), n2 = vstep<K>(t2, h2), n3 = vstep<K>(t3, h3); \
        float4 r = solve_ab(box_out(n0), box_out(n1), box_out(n2), box_out(n3), o2[K], o3[K], o4[K]); \
        if (MODE == 0) { if (step >= 7 && step < n && ovalid) abd[(size_t)(ybase + step - 3) * W + xo] = r; } \
        else { KEEP(r.x); KEEP(r.y); KEEP(r.z); KEEP(r.w); } \
        __builtin_amdgcn_sched_barrier(0); }
        STEP(0) STEP(1) STEP(2) STEP(3)
    }
}
template <int MODE> float run(const float *vol, float4 *ab, const float4 *g1, const float4 *g2, const float4 *g3, const float2 *g4, int W, int H, int D, int seg_rows)
{
    int nstrips = (W + OUT_PER_WAVE - 1) / OUT_PER_WAVE, nsegs = (H + seg_rows - 1) / seg_rows, nzg = (D + 3) / 4;
    int npairs = nstrips * nsegs, nblocks = 8 * ((npairs + 7) / 8) * nzg;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b); float best = 1e9;
    for (int it = 0; it < 4; ++it) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(k_dec<MODE>, dim3(nblocks), dim3(512), 0, 0, vol, ab, g1, g2, g3, g4, W, H, D, nstrips, nsegs, seg_rows, nzg);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b); float ms; (void)hipEventElapsedTime(&ms, a, b); if (it > 0 && ms < best) best = ms;
    }
    return best;
}
int main(int argc, char **argv)
{
    int W = 1920, H = 1080, D = 256, seg = argc > 1 ? atoi(argv[1]) : 135;
    size_t HW = (size_t)W * H;
    float *vol; float4 *ab, *g1, *g2, *g3; float2 *g4;
    (void)hipMalloc(&vol, HW * D * 4); (void)hipMalloc(&ab, HW * D * 16); (void)hipMalloc(&g1, HW * 16); (void)hipMalloc(&g2, HW * 16); (void)hipMalloc(&g3, HW * 16); (void)hipMalloc(&g4, HW * 8);
    std::vector<float> h(HW * 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) * 1e-3f + 0.01f;
    (void)hipMemcpy(g1, h.data(), HW * 16, hipMemcpyHostToDevice); (void)hipMemcpy(g2, h.data(), HW * 16, hipMemcpyHostToDevice);
    (void)hipMemcpy(g3, h.data(), HW * 16, hipMemcpyHostToDevice); (void)hipMemcpy(g4, h.data(), HW * 8, hipMemcpyHostToDevice);
    (void)hipMemset(vol, 0, HW * D * 4);
    printf("normal (store in compute waves)      %.3f ms\n", run<0>(vol, ab, g1, g2, g3, g4, W, H, D, seg));
    printf("compute waves + separate store waves %.3f ms\n", run<1>(vol, ab, g1, g2, g3, g4, W, H, D, seg));
    printf("compute waves only (no stores)       %.3f ms\n", run<2>(vol, ab, g1, g2, g3, g4, W, H, D, seg));
    printf("store waves only                     %.3f ms\n", run<3>(vol, ab, g1, g2, g3, g4, W, H, D, seg));
    return 0;
}
