// Scratch experiment (not product): ablation of the stage-A marching kernel at 1920x1080x256.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -I primestereomatch_amd/csrc scripts/exp/ablate_a.hip -o /tmp/ablate_a
#include "psm_kernels.hip"
#include <cstdio>
#include <vector>
using namespace psm;

#define KEEP(x) asm volatile("" ::"v"(x))

template <int NW, int ABL>
__global__ __launch_bounds__(NW * 64) void k_a(const float *__restrict__ vol, float4 *__restrict__ ab,
                                              const float4 *__restrict__ G1, const float4 *__restrict__ G2,
                                              const float4 *__restrict__ G3, const float2 *__restrict__ G4,
                                              int W, int H, int Dloc, int nstrips, int nsegs, int seg_rows, int nzg, int order, size_t sv, size_t sa)
{
    const MarchPos pos = march_pos<NW>(W, H, Dloc, nstrips, nsegs, seg_rows, nzg, order);
    if (!pos.ok) return;
    PSM_LANE_IDX();
    const size_t HW = (size_t)H * W;
    const float *vd = vol + (size_t)pos.d * sv;
    float4 *abd = ab + (size_t)pos.d * sa; (void)HW;
    VTree t0 = {}, t1 = {}, t2 = {}, t3 = {};
    const int n = (pos.y1 - pos.y0) + 7;
    const int ybase = pos.y0 - 4;
    const int xoc = min(pos.xo, W - 1);
    float pin[4]; float4 gin[4]; float4 o2[4], o3[4]; float2 o4[4];
    const float fl = (float)pos.lane * 1e-3f;
#define ISSUE(SLOT, STEP)                                                               \
    {                                                                                   \
        const size_t off_ = (size_t)r101c(ybase + (STEP), H) * W + pos.cs;              \
        if (ABL & 16) pin[SLOT] = fl + (float)(STEP); else pin[SLOT] = vd[off_];        \
        if (ABL & 8) gin[SLOT] = make_float4(fl, fl * 2, fl * 3, 0.f); else gin[SLOT] = G1[off_]; \
        int yo_ = ybase + (STEP) - 3;                                                   \
        yo_ = yo_ < 0 ? 0 : (yo_ > H - 1 ? H - 1 : yo_);                                \
        const size_t oo_ = (size_t)yo_ * W + xoc;                                       \
        if (ABL & 2) { o2[SLOT] = make_float4(fl, fl, fl, 1.f); o3[SLOT] = make_float4(fl, fl, 1.f, fl); o4[SLOT] = make_float2(fl, 1.f); } \
        else { o2[SLOT] = G2[oo_]; o3[SLOT] = G3[oo_]; o4[SLOT] = G4[oo_]; }            \
    }
    auto HS = [&](float v) -> double {
        if (ABL & 1) { double a = (double)v; double s2 = a + (double)(v * 1.5f); double s4 = s2 + s2 * 0.5; return s4 + s4 * 0.25; }
        return hsum8(v, i1, i2, i4);
    };
    ISSUE(0, 0) __builtin_amdgcn_sched_barrier(0);
    ISSUE(1, 1) __builtin_amdgcn_sched_barrier(0);
    ISSUE(2, 2) __builtin_amdgcn_sched_barrier(0);
    for (int i = 0; i < n; i += 4) {
#define STEP(K)                                                                                     \
    {                                                                                               \
        const int step = i + K;                                                                     \
        ISSUE((K + 3) & 3, step + 3)                                                                \
        const float p = pin[K];                                                                     \
        double h0 = HS(p), h1 = HS(__fmul_rn(gin[K].x, p)), h2 = HS(__fmul_rn(gin[K].y, p)), h3 = HS(__fmul_rn(gin[K].z, p)); \
        double n0, n1, n2, n3;                                                                      \
        if (ABL & 32) { n0 = h0; n1 = h1; n2 = h2; n3 = h3; }                                       \
        else { n0 = vstep<K>(t0, h0); n1 = vstep<K>(t1, h1); n2 = vstep<K>(t2, h2); n3 = vstep<K>(t3, h3); } \
        float4 r;                                                                                   \
        if (ABL & 64) r = make_float4(box_out(n0) + o2[K].x, box_out(n1) + o3[K].x, box_out(n2) + o4[K].x, box_out(n3)); \
        else r = solve_ab(box_out(n0), box_out(n1), box_out(n2), box_out(n3), o2[K], o3[K], o4[K]); \
        if (ABL & 4) { KEEP(r.x); KEEP(r.y); KEEP(r.z); KEEP(r.w); }                                \
        else if (step >= 7 && step < n && pos.ovalid) abd[(size_t)(ybase + step - 3) * W + pos.xo] = r; \
        __builtin_amdgcn_sched_barrier(0);                                                          \
    }
        STEP(0) STEP(1) STEP(2) STEP(3)
    }
}

template <int ABL>
float run(const float *vol, float4 *ab, const float4 *g1, const float4 *g2, const float4 *g3, const float2 *g4,
          int W, int H, int D, int seg_rows, int order, size_t sv, size_t sa)
{
    int nstrips = (W + 56) / 57, nsegs = (H + seg_rows - 1) / seg_rows, nzg = (D + 3) / 4;
    int npairs = nstrips * nsegs, nblocks = order == 0 ? 8 * ((npairs + 7) / 8) * nzg : 8 * ((nstrips + 7) / 8) * nsegs * nzg;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9;
    for (int it = 0; it < 4; ++it) {
        hipEventRecord(a);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_a<4, ABL>), dim3(nblocks), dim3(256), 0, 0, vol, ab, g1, g2, g3, g4, W, H, D, nstrips, nsegs, seg_rows, nzg, order, sv, sa);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (it > 0 && ms < best) best = ms;
    }
    return best;
}

int main(int argc, char **argv)
{
    int W = 1920, H = 1080, D = 256, seg = argc > 1 ? atoi(argv[1]) : 135;
    size_t HW = (size_t)W * H, maxpad = 65536;
    float *vol; float4 *ab, *g1, *g2, *g3; float2 *g4;
    (void)hipMalloc(&vol, (HW + maxpad) * D * 4); (void)hipMalloc(&ab, (HW + maxpad) * D * 16);
    (void)hipMalloc(&g1, HW * 16); (void)hipMalloc(&g2, HW * 16); (void)hipMalloc(&g3, HW * 16); (void)hipMalloc(&g4, HW * 8);
    std::vector<float> h(HW * 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) * 1e-3f + 0.01f;
    (void)hipMemcpy(g1, h.data(), HW * 16, hipMemcpyHostToDevice); (void)hipMemcpy(g2, h.data(), HW * 16, hipMemcpyHostToDevice);
    (void)hipMemcpy(g3, h.data(), HW * 16, hipMemcpyHostToDevice); (void)hipMemcpy(g4, h.data(), HW * 8, hipMemcpyHostToDevice);
    (void)hipMemset(vol, 0, (HW + maxpad) * D * 4);
    for (int order = 0; order < 2; ++order) {
        size_t sv = HW, sa = HW;
        printf("order %d: stores+nothing %.3f | nothing %.3f | loads-only(no compute,no store) %.3f | p-load only %.3f | guidance-out only %.3f | G1 only %.3f\n", order,
               run<1 | 2 | 8 | 16 | 32 | 64>(vol, ab, g1, g2, g3, g4, W, H, D, seg, order, sv, sa),
               run<127>(vol, ab, g1, g2, g3, g4, W, H, D, seg, order, sv, sa),
               run<1 | 4 | 32 | 64>(vol, ab, g1, g2, g3, g4, W, H, D, seg, order, sv, sa),
               run<1 | 2 | 4 | 8 | 32 | 64>(vol, ab, g1, g2, g3, g4, W, H, D, seg, order, sv, sa),
               run<1 | 4 | 8 | 16 | 32 | 64>(vol, ab, g1, g2, g3, g4, W, H, D, seg, order, sv, sa),
               run<1 | 2 | 4 | 16 | 32 | 64>(vol, ab, g1, g2, g3, g4, W, H, D, seg, order, sv, sa));
    }
    return 0;
}
