// psm_api.cpp - the C ABI of libprimesm_hip.so (include/primesm_hip.h): context, device memory,
// stage sequencing, timing.  Takes the place of the reference's oclUtil + CVC_cl/CVF_cl/DispSel_cl
// host wrappers (src/oclUtil.cpp, src/CVC_cl.cpp, src/CVF_cl.cpp, src/DispSel_cl.cpp).
#include "../../include/primesm_hip.h"
#include "psm_kernels.h"

#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

using namespace psm;

namespace {

std::string g_create_error;

struct KernelTimer {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    double total_ms = 0.0;
    int launches = 0;
};

double now_us()
{
    using namespace std::chrono;
    return duration_cast<duration<double, std::micro>>(steady_clock::now().time_since_epoch()).count();
}

}  // namespace

extern "C" int filter_u8_stored(struct psm_ctx *c, int side);   // defined below; internal (not part of the ABI header)

struct psm_ctx {
    int W = 0, H = 0, D = 0, d0 = 0, d1 = 0, Dloc = 0, dtype = PSM_F32, device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;

    // device memory (DESIGN.md "HBM layout")
    void *raw[2] = {nullptr, nullptr};  // staged copy of the interleaved host images
    size_t raw_bytes = 0;
    int raw_depth = -1;                 // PSM_IMG_* of the staged pair, -1 = nothing uploaded
    Guidance g[2] = {};
    double *hs9 = nullptr;              // scratch of the two-pass guidance kernels (PSM_OPT_FLAGS 256), allocated on first use
    void *vol[2] = {nullptr, nullptr};  // [Dloc][H][W] float (PSM_F32) or uint8 (PSM_U8)
    float *fvol = nullptr;              // PSM_U8 only: float work volume of one side
    float *spare = nullptr;             // PSM_F32: output volume of the fused filter (ping-pong with vol[side])
    float4 *ab = nullptr;               // [Dloc][H][W] {a0,a1,a2,b}; also box8 output
    long long *keys = nullptr;          // [2][H][W]
    long long *keys_cur = nullptr;      // where the packed minima go: `keys`, or the caller's buffer (psm_set_key_buffer)
    long long *gather = nullptr;        // [gather_ranks][2][H][W], psm_disp_merge_ctx
    int gather_ranks = 0;
    uint8_t *maps = nullptr;            // [2][H][W]: maps_own, or the caller's buffer (psm_set_map_buffer)
    uint8_t *maps_own = nullptr;
    bool have_rows = false;             // the maps hold the rows of this context's stripe only (psm_set_rows) for this frame
    uint8_t *valid = nullptr;           // [2][H][W]
    uint8_t *pinned = nullptr;          // [2][H][W] page-locked bounce buffer for map / mask downloads (on first use)
    int *wm = nullptr;                  // psm_wgt_median scratch: nxt[H][W+1], prog[H], err[1]; allocated on first use
    uint8_t *wm_par = nullptr;          // scratch of its parallel (sweep) form, per side: orig, newv (bytes), stamp, 2 active lists, changed list, counters
    int wm_sweeps[2] = {0, 0};          // last call: sweeps until the fixed point (-1: dataflow form), evaluations
    long long wm_evals[2] = {0, 0};
    uint8_t *p4[2] = {nullptr, nullptr};  // PSM_U8 only: {c0,c1,c2,grad} words
    float *soa[2] = {nullptr, nullptr};   // planar copies of g1..g4 (14 planes) for the two-columns-per-lane filter
    int soa_state[2] = {0, 0};            // 0 nothing, 1 g1 planes, 2 all planes (of the current image pair)
    // After psm_cost_filter_fgf the filtered volume of a side may stay virtual (fgf_virtual[side] = subsample rate):
    // it is fully described by the smoothed low-resolution models fgf_mab[side]; the WTA consumes them directly
    // (upsample + model + argmin in one pass), any other reader of vol[side] materialises it first (materialize()).
    int fgf_virtual[2] = {0, 0};
    // After psm_cost_filter (default path) the filtered volume of a side is virtual as well (gf_virtual[side]): the fused
    // kernel ran in "select" mode - cost build, guided filter and the WTA over the local slices in one pass - and left
    // the packed per-pixel minima in keys[side].  vol[side] is then untouched (raw_rows[side] still describes the
    // UNFILTERED volume); any reader of the filtered volume re-runs the filter in "store" mode first (materialize()).
    bool gf_virtual[2] = {false, false};
    bool have_guid[2] = {false, false};   // g2..g4 of a side are those of the current image pair
    int guid_y0 = 0, guid_y1 = 0;         // ... or, while have_guid is false, only their rows [guid_y0, guid_y1) of both sides are (row stripes)
    int g1_y0 = 0, g1_y1 = 0;             // likewise for g1 (and the 8-bit planes) while have_g1 is false
    void *gf_scratch = nullptr;         // chunk planes of the select-mode kernel (PcPlan::scratch_bytes)
    size_t gf_scratch_bytes = 0;
    // Stride of the seeding phase, tuned in place: the first frames of a geometry run the candidates once each (same
    // results whatever the stride), timed with events on the launch stream; the fastest is kept.
    struct Tune {
        int W = 0, H = 0, D = 0, y0 = 0, y1 = 0, dtype = -1;
        int calls = 0, best = 0;
        int n[3] = {0, 0, 0};                     // measurements taken per candidate
        float ms[3] = {-1.f, -1.f, -1.f};         // fastest of them
        hipEvent_t e0[3] = {nullptr, nullptr, nullptr}, e1[3] = {nullptr, nullptr, nullptr};
        bool pend[3] = {false, false, false};
    } tune;
    int *gf_cnt = nullptr;              // slice counters of the dynamic select form (one per side and pair)
    size_t gf_cnt_n = 0;
    float4 *fgf_mab[2] = {nullptr, nullptr};
    void *fgf = nullptr;                // psm_cost_filter_fgf scratch (small planes), fgf_bytes long
    size_t fgf_bytes = 0;

    bool have_images = false, have_g1 = false, have_cost = false, have_maps = false, have_valid = false;
    bool have_keys = false;             // keys_cur holds the packed minima of the current frame's local slices (both sides)
    bool have_keys_side[2] = {false, false};
    // raw_rows[side]: which rows of the unfiltered cost volume exist in memory.  psm_cost_construct may
    // leave the volume virtual (RAW_NONE): the fused filter builds the costs on the fly from the g1
    // planes.  Anything else that reads the volume materialises it first (materialize()).
    enum { RAW_ALL = 0, RAW_NONE = 1, RAW_BANDS = 2 };
    int raw_rows[2] = {RAW_ALL, RAW_ALL};

    // options
    int opt_async = 0, opt_variant = 0, opt_profile = 0;
    March march = {0, 4, 0};

    double stage_us[PSM_STAGE_COUNT] = {0, 0, 0, 0};
    KernelTimer timers[PSM_K_COUNT];
    std::vector<hipEvent_t> event_pool;
    std::string err;
};

namespace {

int fail(psm_ctx *c, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    fprintf(stderr, "primesm_hip: %s\n", buf);  // the _cl wrappers print to stderr too (src/CVC_cl.cpp:185-210)
    return 1;
}

#define PSM_HIP(c, call)                                                                         \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) return fail((c), "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

size_t velem(const psm_ctx *c) { return c->dtype == PSM_U8 ? 1 : 4; }

hipEvent_t get_event(psm_ctx *c)
{
    if (!c->event_pool.empty()) {
        hipEvent_t e = c->event_pool.back();
        c->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;    // Prof then skips this launch's timing instead of recording a null event
    }
    return e;
}

// RAII bracket of one kernel launch with hipEvents on the launch stream (PSM_OPT_PROFILE)
struct Prof {
    psm_ctx *c;
    int k;
    hipEvent_t a = nullptr, b = nullptr;
    Prof(psm_ctx *c_, int k_) : c(c_), k(k_)
    {
        if (c->opt_profile) {
            a = get_event(c);
            b = a ? get_event(c) : nullptr;
            if (a && !b) { c->event_pool.push_back(a); a = nullptr; }
            if (a) (void)hipEventRecord(a, c->stream);
        }
    }
    ~Prof()
    {
        if (a) {
            (void)hipEventRecord(b, c->stream);
            c->timers[k].pending.emplace_back(a, b);
        }
    }
};

int check_launch(psm_ctx *c, const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(c, "launch of %s failed: %s", what, hipGetErrorString(e));
    return 0;
}

int end_stage(psm_ctx *c, int stage, double t0)
{
    if (!c->opt_async) PSM_HIP(c, hipStreamSynchronize(c->stream));
    c->stage_us[stage] = now_us() - t0;
    return 0;
}

int bind(psm_ctx *c)
{
    PSM_HIP(c, hipSetDevice(c->device));
    return 0;
}

void free_all(psm_ctx *c)
{
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (int s = 0; s < 2; ++s) {
        (void)hipFree(c->raw[s]);
        (void)hipFree(c->g[s].g1);
        (void)hipFree(c->g[s].g2);
        (void)hipFree(c->g[s].g3);
        (void)hipFree(c->g[s].g4);
        (void)hipFree(c->vol[s]);
        (void)hipFree(c->p4[s]);
        (void)hipFree(c->soa[s]);
    }
    (void)hipFree(c->hs9);
    (void)hipFree(c->fvol);
    (void)hipFree(c->spare);
    (void)hipFree(c->ab);
    (void)hipFree(c->keys);
    (void)hipFree(c->gather);
    (void)hipFree(c->maps_own);
    (void)hipFree(c->valid);
    if (c->pinned) (void)hipHostFree(c->pinned);
    (void)hipFree(c->wm);
    (void)hipFree(c->wm_par);
    (void)hipFree(c->gf_scratch);
    (void)hipFree(c->gf_cnt);
    (void)hipFree(c->fgf);
    for (auto &t : c->timers)
        for (auto &p : t.pending) {
            (void)hipEventDestroy(p.first);
            (void)hipEventDestroy(p.second);
        }
    for (auto e : c->event_pool) (void)hipEventDestroy(e);
    for (int i = 0; i < 3; ++i) {
        if (c->tune.e0[i]) (void)hipEventDestroy(c->tune.e0[i]);
        if (c->tune.e1[i]) (void)hipEventDestroy(c->tune.e1[i]);
    }
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
}

// planarise + scale + gray + x-gradient of both staged images -> g1 (and the 8-bit planes)
// (rows [ya, yb) only - the kernels are row-independent - when a row stripe is all the following filter will read: have_g1
// then stays false and g1_y0/1 say what is there; every other consumer finds have_g1 false and prepares the whole image)
int run_prep(psm_ctx *c, int ya = 0, int yb = 0)
{
    const bool whole = yb <= ya || (ya <= 0 && yb >= c->H);
    if (whole) { ya = 0; yb = c->H; }
    ya = ya < 0 ? 0 : ya;
    yb = yb > c->H ? c->H : yb;
    const size_t row = (size_t)c->W * 3 * (c->raw_depth == PSM_IMG_F32 ? 4 : 1);
    const size_t o = (size_t)ya * c->W;
    {   // both images in one launch
        Prof p(c, PSM_K_PREP);
        launch_prep(c->stream, (const char *)c->raw[0] + ya * row, row, c->raw_depth == PSM_IMG_F32, c->W, yb - ya, c->g[0].g1 + o,
                    (const char *)c->raw[1] + ya * row, c->g[1].g1 + o);
    }
    for (int s = 0; s < 2 && c->dtype == PSM_U8; ++s) {
        Prof p(c, PSM_K_PREP);
        launch_prep_u8(c->stream, (const uint8_t *)c->raw[s] + ya * row, row, c->W, yb - ya, c->p4[s] + 4 * o);
    }
    if (check_launch(c, "prep")) return 1;
    c->soa_state[0] = c->soa_state[1] = 0;
    c->have_guid[0] = c->have_guid[1] = false;
    c->guid_y0 = c->guid_y1 = 0;
    c->have_g1 = whole;
    c->g1_y0 = ya;
    c->g1_y1 = yb;
    return 0;
}

// The float volumes are allocated on first use: the default path (lazy costs + select-mode filter) never touches them.
int ensure_vol(psm_ctx *c, int side)
{
    if (c->vol[side]) return 0;
    const size_t V = (size_t)c->W * c->H * c->Dloc;
    PSM_HIP(c, hipMalloc(&c->vol[side], V * velem(c)));
    return 0;
}

// chunk planes of the select-mode fused kernel (shared by both sides: each side reduces them to keys[side] right away)
int ensure_gf_scratch(psm_ctx *c, size_t bytes)
{
    if (c->gf_scratch && c->gf_scratch_bytes >= bytes) return 0;
    PSM_HIP(c, hipStreamSynchronize(c->stream));
    (void)hipFree(c->gf_scratch);
    c->gf_scratch = nullptr;
    c->gf_scratch_bytes = 0;
    PSM_HIP(c, hipMalloc(&c->gf_scratch, bytes));
    c->gf_scratch_bytes = bytes;
    return 0;
}

int ensure_gf_cnt(psm_ctx *c, size_t n)
{
    if (c->gf_cnt && c->gf_cnt_n >= n) return 0;
    PSM_HIP(c, hipStreamSynchronize(c->stream));
    (void)hipFree(c->gf_cnt);
    c->gf_cnt = nullptr;
    c->gf_cnt_n = 0;
    PSM_HIP(c, hipMalloc((void **)&c->gf_cnt, n * sizeof(int)));
    c->gf_cnt_n = n;
    return 0;
}

// The 16 B/voxel (a0,a1,a2,b) scratch is only needed by the two-stage filter, psm_filter_stage_a and
// psm_box8_volume: allocate it on first use.
int ensure_ab(psm_ctx *c)
{
    if (c->ab) return 0;
    const size_t V = (size_t)c->W * c->H * c->Dloc;
    PSM_HIP(c, hipMalloc((void **)&c->ab, V * sizeof(float4)));
    return 0;
}

// planar guidance of `side` for k_cvf_pc2: level 1 = g1 planes, 2 = all 14 planes (needs launch_guidance(side) done)
int ensure_soa(psm_ctx *c, int side, int level)
{
    if (!c->soa[side]) PSM_HIP(c, hipMalloc((void **)&c->soa[side], (size_t)14 * c->W * c->H * sizeof(float)));
    if (c->soa_state[side] >= level) return 0;
    launch_soa(c->stream, c->g[side], c->W, c->H, c->soa[side], level == 1);
    c->soa_state[side] = level;
    return check_launch(c, "soa");
}

// second float volume for the fused filter when it has to READ a materialised cost volume (out of place)
int ensure_spare(psm_ctx *c)
{
    if (c->spare) return 0;
    const size_t V = (size_t)c->W * c->H * c->Dloc;
    PSM_HIP(c, hipMalloc((void **)&c->spare, V * sizeof(float)));
    return 0;
}

// build the (float) cost slices of `side` for rows [ybeg, yend)
void launch_cvc_rows(psm_ctx *c, int side, int ybeg, int yend)
{
    Prof p(c, PSM_K_CVC);
    // buildCV_right is called with the images swapped (src/DispEst.cpp:217,260)
    launch_cvc(c->stream, c->g[side].g1, c->g[1 - side].g1, (float *)c->vol[side], c->W, c->H, c->d0, c->Dloc, side,
               c->march.flags, ybeg, yend);
}

// a virtual Fast-Guided-Filter result becomes a real volume
int fgf_flush(psm_ctx *c, int side)
{
    if (!c->fgf_virtual[side]) return 0;
    if (ensure_vol(c, side)) return 1;
    {
        Prof p(c, PSM_K_FGF);
        launch_fgf_apply(c->stream, (float *)c->vol[side], c->g[side].g1, c->W, c->H, c->Dloc, c->fgf_virtual[side], c->fgf_mab[side]);
    }
    c->fgf_virtual[side] = 0;
    return check_launch(c, "fgf (upsample)");
}

// make sure the whole volume of `side` (unfiltered, or filtered by psm_cost_filter_fgf) is in memory
// g1 (and the 8-bit planes) and the guidance of the whole image: a row-stripe filter leaves only its own rows behind
static int ensure_whole_planes(psm_ctx *c)
{
    if (!c->have_g1 && run_prep(c)) return 1;
    if (!(c->have_guid[0] && c->have_guid[1])) {
        {
            Prof p(c, PSM_K_GUIDE);
            launch_guidance(c->stream, c->g[0], nullptr, c->W, c->H, 0, &c->g[1]);
        }
        c->have_guid[0] = c->have_guid[1] = true;
        c->guid_y0 = 0;
        c->guid_y1 = c->H;
        return check_launch(c, "guidance");
    }
    return 0;
}

int materialize(psm_ctx *c, int side)
{
    if (fgf_flush(c, side)) return 1;
    if ((c->gf_virtual[side] || c->raw_rows[side] != psm_ctx::RAW_ALL) && ensure_whole_planes(c)) return 1;
    if (c->dtype == PSM_U8) {
        if (c->raw_rows[side] != psm_ctx::RAW_ALL) {      // the 8-bit costs exist only as a recipe: build them
            Prof p(c, PSM_K_CVC);
            launch_cvc_u8(c->stream, c->p4[side], c->p4[1 - side], (uint8_t *)c->vol[side], c->W, c->H, c->d0, c->Dloc, side);
            c->raw_rows[side] = psm_ctx::RAW_ALL;
        }
        if (c->gf_virtual[side]) {                         // ... and the filtered volume only as WTA keys: filter in the storing form
            if (filter_u8_stored(c, side)) return 1;
            c->gf_virtual[side] = false;
        }
        return check_launch(c, "8-bit volume (materialize)");
    }
    if (c->gf_virtual[side]) {
        // the guided-filter result exists only as WTA keys: run the same fused kernel again, this time storing q
        if (ensure_vol(c, side)) return 1;
        Prof p(c, PSM_K_CVF_F);
        if (c->raw_rows[side] == psm_ctx::RAW_ALL) {          // materialised costs in vol[side]: out of place
            if (ensure_spare(c)) return 1;
            launch_cvf_fused(c->stream, c->march, (const float *)c->vol[side], c->spare, c->g[side], c->W, c->H, c->Dloc, 0, c->H,
                             c->g[1 - side].g1, c->d0, 0);
            float *t = (float *)c->vol[side];
            c->vol[side] = c->spare;
            c->spare = t;
        } else {
            launch_cvf_fused(c->stream, c->march, nullptr, (float *)c->vol[side], c->g[side], c->W, c->H, c->Dloc, 0, c->H,
                             c->g[1 - side].g1, c->d0, 1 + side);
        }
        c->gf_virtual[side] = false;
        c->raw_rows[side] = psm_ctx::RAW_ALL;                 // vol[side] now holds real (filtered) data
        return check_launch(c, "cvf (materialize)");
    }
    if (c->dtype != PSM_F32 || c->raw_rows[side] == psm_ctx::RAW_ALL) return 0;
    if (ensure_vol(c, side)) return 1;
    launch_cvc_rows(c, side, 0, c->H);
    c->raw_rows[side] = psm_ctx::RAW_ALL;
    return check_launch(c, "cvc (materialize)");
}

int flush_timers(psm_ctx *c)
{
    PSM_HIP(c, hipStreamSynchronize(c->stream));
    for (auto &t : c->timers) {
        for (auto &p : t.pending) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) {
                t.total_ms += ms;
                t.launches += 1;
            }
            c->event_pool.push_back(p.first);
            c->event_pool.push_back(p.second);
        }
        t.pending.clear();
    }
    return 0;
}

}  // namespace

extern "C" {

int psm_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int psm_create_shard(psm_ctx **out, int width, int height, int max_disp, int d_begin, int d_end, int dtype, int device)
{
    if (!out) return fail(nullptr, "psm_create: out is NULL");
    *out = nullptr;
    if (width < 8 || height < 8) return fail(nullptr, "psm_create: image %dx%d smaller than the 8x8 filter window", width, height);
    // the marching kernels address the 16-byte guidance planes with 32-bit byte offsets (buffer descriptors of W*H*16
    // bytes, row offsets y*W*16): W*H < 2^27; the post-processing row kernels keep two ints per column in LDS: W <= 8192
    if ((long long)width * height >= (1LL << 27)) return fail(nullptr, "psm_create: image %dx%d too large (W*H must be < 2^27)", width, height);
    if (width > 8192) return fail(nullptr, "psm_create: width %d > 8192", width);
    if (max_disp < 1 || max_disp > 256) return fail(nullptr, "psm_create: max_disp %d outside [1,256] (maps are 8-bit)", max_disp);
    // lrCheck indexes (x - d + W) % W (src/PP.cpp:28): negative - undefined in the reference - once d > W
    if (max_disp > width) return fail(nullptr, "psm_create: max_disp %d > width %d", max_disp, width);
    if (d_begin < 0 || d_end > max_disp || d_begin >= d_end) return fail(nullptr, "psm_create: bad slice range [%d,%d) of %d", d_begin, d_end, max_disp);
    if (dtype != PSM_F32 && dtype != PSM_U8) return fail(nullptr, "psm_create: unknown dtype %d", dtype);
    int ndev = psm_device_count();
    if (ndev <= 0) return fail(nullptr, "psm_create: no HIP device available");
    if (device < 0 || device >= ndev) return fail(nullptr, "psm_create: device %d not in [0,%d)", device, ndev);

    psm_ctx *c = new psm_ctx();
    c->W = width; c->H = height; c->D = max_disp; c->d0 = d_begin; c->d1 = d_end; c->Dloc = d_end - d_begin;
    c->dtype = dtype; c->device = device;
    const size_t HW = (size_t)width * height;
    const size_t V = HW * (size_t)c->Dloc;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    c->stream = c->own_stream;
    c->raw_bytes = HW * 3 * sizeof(float);
    for (int s = 0; s < 2 && e == hipSuccess; ++s) {
        e = hipMalloc(&c->raw[s], c->raw_bytes);
        if (e == hipSuccess) e = hipMalloc((void **)&c->g[s].g1, HW * sizeof(float4));
        if (e == hipSuccess) e = hipMalloc((void **)&c->g[s].g2, HW * sizeof(float4));
        if (e == hipSuccess) e = hipMalloc((void **)&c->g[s].g3, HW * sizeof(float4));
        if (e == hipSuccess) e = hipMalloc((void **)&c->g[s].g4, HW * sizeof(float2));
        if (e == hipSuccess && dtype == PSM_U8) e = hipMalloc(&c->vol[s], V * velem(c));   // PSM_F32: on first use (ensure_vol)
        if (e == hipSuccess && dtype == PSM_U8) e = hipMalloc((void **)&c->p4[s], HW * 4);
    }
    // (fvol, the float work copy of the 8-bit storing path, is allocated on first use)
    if (e == hipSuccess) e = hipMalloc((void **)&c->keys, 2 * HW * sizeof(long long));
    c->keys_cur = c->keys;
    if (e == hipSuccess) e = hipMalloc((void **)&c->maps_own, 2 * HW + 4);   // +4: psm_wgt_median reads/updates whole aligned dwords
    c->maps = c->maps_own;
    if (e == hipSuccess) e = hipMalloc((void **)&c->valid, 2 * HW);
    if (e != hipSuccess) {
        fail(nullptr, "psm_create: device setup failed: %s", hipGetErrorString(e));
        free_all(c);
        delete c;
        return 1;
    }
    *out = c;
    return 0;
}

int psm_create(psm_ctx **out, int width, int height, int max_disp, int dtype, int device)
{
    return psm_create_shard(out, width, height, max_disp, 0, max_disp, dtype, device);
}

void psm_destroy(psm_ctx *ctx)
{
    if (!ctx) return;
    free_all(ctx);
    delete ctx;
}

const char *psm_last_error(const psm_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int psm_get_info(const psm_ctx *c, int *width, int *height, int *max_disp, int *d_begin, int *d_end, int *dtype, int *device)
{
    if (!c) return 1;
    if (width) *width = c->W;
    if (height) *height = c->H;
    if (max_disp) *max_disp = c->D;
    if (d_begin) *d_begin = c->d0;
    if (d_end) *d_end = c->d1;
    if (dtype) *dtype = c->dtype;
    if (device) *device = c->device;
    return 0;
}

int psm_set_option(psm_ctx *c, int option, int value)
{
    if (!c) return 1;
    switch (option) {
    case PSM_OPT_ASYNC: c->opt_async = value != 0; return 0;
    case PSM_OPT_KERNEL_VARIANT:
        if (value != 0 && value != 1) return fail(c, "psm_set_option: kernel variant %d unknown", value);
        c->opt_variant = value; return 0;
    case PSM_OPT_PROFILE: c->opt_profile = value != 0; return 0;
    case PSM_OPT_SEG_ROWS:
        if (value < 0) return fail(c, "psm_set_option: seg_rows %d < 0", value);
        c->march.seg_rows = value; return 0;
    case PSM_OPT_WAVES:
        if (value != 1 && value != 2 && value != 4 && value != 8) return fail(c, "psm_set_option: waves %d not in {1,2,4,8}", value);
        c->march.waves = value; return 0;
    case PSM_OPT_FLAGS: c->march.flags = value; return 0;
    default: return fail(c, "psm_set_option: unknown option %d", option);
    }
}

int psm_set_stream(psm_ctx *c, void *hip_stream)
{
    if (!c) return 1;
    if (bind(c)) return 1;
    PSM_HIP(c, hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return 0;
}

int psm_synchronize(psm_ctx *c)
{
    if (!c) return 1;
    if (bind(c)) return 1;
    PSM_HIP(c, hipStreamSynchronize(c->stream));
    return 0;
}

int psm_upload_pair(psm_ctx *c, const void *l, const void *r, int channels, size_t stride_bytes, int depth)
{
    if (!c) return 1;
    if (!l || !r) return fail(c, "psm_upload_pair: NULL image");
    if (channels != 3) return fail(c, "psm_upload_pair: %d channels (3 required, B,G,R interleaved)", channels);
    if (depth != PSM_IMG_U8 && depth != PSM_IMG_F32) return fail(c, "psm_upload_pair: unknown depth %d", depth);
    if (c->dtype == PSM_U8 && depth != PSM_IMG_U8) return fail(c, "psm_upload_pair: 8-bit mode needs 8-bit images");
    const size_t row = (size_t)c->W * 3 * (depth == PSM_IMG_F32 ? 4 : 1);
    if (stride_bytes == 0) stride_bytes = row;
    if (stride_bytes < row) return fail(c, "psm_upload_pair: stride %zu < row size %zu", stride_bytes, row);
    if (bind(c)) return 1;
    const void *src[2] = {l, r};
    for (int s = 0; s < 2; ++s)
        PSM_HIP(c, hipMemcpy2DAsync(c->raw[s], row, src[s], stride_bytes, row, c->H, hipMemcpyHostToDevice, c->stream));
    // the copy reads caller memory: always complete it before returning (CVC_cl::buildCV copies
    // out of the cv::Mats synchronously, src/CVC_cl.cpp:113-160)
    PSM_HIP(c, hipStreamSynchronize(c->stream));
    c->raw_depth = depth;
    c->have_images = true;
    c->have_g1 = false;
    c->g1_y0 = c->g1_y1 = 0;
    c->have_cost = false;
    c->have_maps = false;
    c->have_valid = false;
    c->have_keys = c->have_keys_side[0] = c->have_keys_side[1] = false;
    c->raw_rows[0] = c->raw_rows[1] = psm_ctx::RAW_ALL;   // nothing virtual survives a new pair
    c->fgf_virtual[0] = c->fgf_virtual[1] = 0;
    c->gf_virtual[0] = c->gf_virtual[1] = false;
    return 0;
}

int psm_cost_construct(psm_ctx *c)
{
    if (!c) return 1;
    if (!c->have_images) return fail(c, "psm_cost_construct: no image pair uploaded");
    if (bind(c)) return 1;
    const double t0 = now_us();
    // Lazy cost volume: when the fused filter will consume the costs (float mode, marching kernels,
    // fusion not disabled) they are built inside that kernel and never written to HBM.
    // (8-bit mode: lazy only when the select-mode kernel will consume the costs - its storing form reads a float copy)
    const bool lazy = c->opt_variant == 0 && !(c->march.flags & (16 | 128)) && c->H >= 8 &&
                      (c->dtype == PSM_F32 || !(c->march.flags & (512 | 8192)));
    // CVC::preprocess belongs to this stage (src/DispEst.cpp:232-233).  A row stripe [y0, y1) with lazy costs reads the image
    // planes of rows y0 - 8 .. y1 + 7 only (costs of the model rows y0 - 4 .. y1 + 2, +- 4 for their box sums, and the guidance)
    const bool striped = c->march.yend > c->march.ybeg;
    if (striped && lazy ? run_prep(c, c->march.ybeg - 8, c->march.yend + 8) : run_prep(c)) return 1;
    c->fgf_virtual[0] = c->fgf_virtual[1] = 0;   // a new cost volume replaces whatever was pending
    c->gf_virtual[0] = c->gf_virtual[1] = false;
    for (int s = 0; s < 2; ++s) {
        if (lazy) {
            c->raw_rows[s] = psm_ctx::RAW_NONE;
        } else if (c->dtype == PSM_U8) {
            Prof p(c, PSM_K_CVC);
            launch_cvc_u8(c->stream, c->p4[s], c->p4[1 - s], (uint8_t *)c->vol[s], c->W, c->H, c->d0, c->Dloc, s);
            c->raw_rows[s] = psm_ctx::RAW_ALL;
        } else if (lazy) {
            c->raw_rows[s] = psm_ctx::RAW_NONE;
        } else {
            if (ensure_vol(c, s)) return 1;
            launch_cvc_rows(c, s, 0, c->H);
            c->raw_rows[s] = psm_ctx::RAW_ALL;
        }
    }
    if (check_launch(c, "cvc")) return 1;
    c->have_cost = true;
    c->have_maps = false;
    c->have_keys = c->have_keys_side[0] = c->have_keys_side[1] = false;
    return end_stage(c, PSM_STAGE_CVC, t0);
}

static int filter_side(psm_ctx *c, int side, bool stage_b)
{
    const size_t V = (size_t)c->W * c->H * c->Dloc;
    const int W = c->W, H = c->H;
    if (!c->have_g1 && run_prep(c)) return 1;  // volume came from psm_upload_volume
    if (fgf_flush(c, side)) return 1;
    if (c->gf_virtual[side] && materialize(c, side)) return 1;   // filtering an already filtered (virtual) volume: make it real first
    if ((c->march.flags & 256) && !c->hs9) PSM_HIP(c, hipMalloc((void **)&c->hs9, (size_t)9 * W * H * sizeof(double)));
    if (c->march.flags & (256 | 65536)) {
        Prof p(c, PSM_K_GUIDE);
        launch_guidance(c->stream, c->g[side], c->hs9, W, H, (c->march.flags & 256) ? 1 : 0);
        c->have_guid[side] = true;
    } else if (!c->have_guid[side]) {
        // the guidance of BOTH images in one launch the first time either side asks (the other side's call then finds it)
        Prof p(c, PSM_K_GUIDE);
        launch_guidance(c->stream, c->g[0], nullptr, W, H, 0, &c->g[1]);
        c->have_guid[0] = c->have_guid[1] = true;
    }
    // Default: the fused kernel in "select" mode - the WTA over the local slices runs inside the filter, the filtered
    // volume stays virtual (flag 8192 forces the storing form; 16 / 512 / the direct variant select other filters)
    const bool sel8 = c->dtype == PSM_U8 && c->raw_rows[side] != psm_ctx::RAW_ALL;   // 8-bit mode: select form only with costs on the fly
    if (stage_b && (c->dtype == PSM_F32 || sel8) && c->opt_variant == 0 && !(c->march.flags & (16 | 512 | 8192)) && H >= 8) {
        const bool lazy = c->raw_rows[side] != psm_ctx::RAW_ALL;
        // flag 16384: the two-columns-per-lane, channel-split form (k_cvf_q2, psm_q2.hip: 14 % fewer VALU instructions,
        // but its four-stage workgroups keep the SIMDs less busy - measured slower, kept as a tested variant)
        const bool q2 = lazy && c->dtype == PSM_F32 && (c->march.flags & 16384);
        if (!q2 && (c->march.flags & 262144)) {
            // flag 262144: shared key plane + atomicMin instead of minima planes (measured slower: ~100 atomics per pixel)
            Prof p(c, PSM_K_CVF_F);
            launch_cvf_select_keys(c->stream, c->march, lazy ? nullptr : (const float *)c->vol[side], c->g[side], W, H, c->Dloc, c->g[1 - side].g1,
                                   c->d0, lazy ? 1 + side : 0, c->keys_cur + side * (size_t)W * H, sel8 ? c->p4[side] : nullptr,
                                   sel8 ? c->p4[1 - side] : nullptr);
            c->gf_virtual[side] = true;
            return check_launch(c, "cvf (fused, select mode, shared keys)");
        }
        const bool dynsel = !q2 && (c->march.flags & 524288);    // flag 524288: dynamic slice distribution (measured slower)
        const PcPlan pl = q2 ? q2_plan(W, H, c->Dloc, c->march.seg_rows) : pc_plan(W, c->march.rows(H), c->Dloc, c->march.seg_rows, dynsel ? 5 : 1);
        if (ensure_gf_scratch(c, pl.scratch_bytes())) return 1;
        if (dynsel && ensure_gf_cnt(c, 2 * (size_t)pl.ngroups * pl.nsegs)) return 1;
        const size_t HW = (size_t)W * H;
        {
            Prof p(c, PSM_K_CVF_F);
            if (q2) launch_cvf_q2(c->stream, c->march, c->g[side], W, H, c->Dloc, c->g[1 - side].g1, c->d0, 1 + side, c->gf_scratch);
            else launch_cvf_select(c->stream, c->march, lazy ? nullptr : (const float *)c->vol[side], c->g[side], W, H, c->Dloc, c->g[1 - side].g1,
                                   c->d0, lazy ? 1 + side : 0, c->gf_scratch, dynsel ? c->gf_cnt : nullptr, sel8 ? c->p4[side] : nullptr,
                                   sel8 ? c->p4[1 - side] : nullptr);
        }
        {
            Prof p(c, PSM_K_WTA);
            if (q2) launch_chunk_min2(c->stream, c->march, W, H, c->Dloc, c->gf_scratch, c->keys_cur + side * HW, nullptr);
            else launch_chunk_min(c->stream, c->march, W, H, c->Dloc, c->gf_scratch, c->keys_cur + side * HW, nullptr, dynsel);
        }
        c->gf_virtual[side] = true;
        return check_launch(c, "cvf (fused, select mode)");
    }
    if (c->dtype == PSM_F32 && ensure_vol(c, side)) return 1;
    if (c->dtype == PSM_U8 && stage_b && c->raw_rows[side] != psm_ctx::RAW_ALL && materialize(c, side)) return 1;   // storing forms read the 8-bit volume
    float *fv = (float *)c->vol[side];
    if (c->dtype == PSM_U8) {
        if (!c->fvol) PSM_HIP(c, hipMalloc((void **)&c->fvol, V * sizeof(float)));
        fv = c->fvol;
        launch_u8_to_f32(c->stream, (const uint8_t *)c->vol[side], fv, V);
    }
    // (8-bit mode: the float copy of the volume goes through the same fused kernel and is re-quantised afterwards)
    const bool fused = stage_b && c->opt_variant == 0 && !(c->march.flags & 16) && H >= 8;
    if (!fused && materialize(c, side)) return 1;
    // flag 512: the two-columns-per-lane form of the producer/consumer kernel (k_cvf_pc2: 31 % fewer VALU
    // instructions per voxel but only two waves per SIMD; measured slower, kept as a tested variant - DESIGN.md 4.2)
    const bool pc2 = fused && c->dtype == PSM_F32 && (c->march.flags & 512) && (W & 3) == 0 && W >= 8;
    if (pc2) {
        {
            Prof p(c, PSM_K_GUIDE);
            if (ensure_soa(c, side, 2) || ensure_soa(c, 1 - side, 1)) return 1;
        }
        const bool lazy = c->raw_rows[side] != psm_ctx::RAW_ALL;
        float *out = fv;
        if (!lazy) {
            if (ensure_spare(c)) return 1;
            out = c->spare;
        }
        {
            Prof p(c, PSM_K_CVF_F);
            launch_cvf_pc2(c->stream, lazy ? nullptr : fv, out, c->soa[side], c->soa[1 - side], W, H, c->Dloc, 0, H, c->d0,
                           lazy ? 1 + side : 0, c->march.seg_rows);
        }
        if (!lazy) {
            c->spare = fv;          // ping-pong: the filtered volume becomes vol[side]
            c->vol[side] = out;
        }
        c->raw_rows[side] = psm_ctx::RAW_ALL;   // vol[side] now holds real (filtered) data
        return check_launch(c, "cvf (fused, two columns per lane)");
    }
    if (fused) {
        if (c->raw_rows[side] != psm_ctx::RAW_ALL) {
            // producer/consumer kernel on a virtual cost volume: nothing is read from vol[side], so the
            // filtered volume is written straight into it
            {
                Prof p(c, PSM_K_CVF_F);
                launch_cvf_fused(c->stream, c->march, nullptr, fv, c->g[side], W, H, c->Dloc, 0, H, c->g[1 - side].g1, c->d0, 1 + side);
            }
            c->raw_rows[side] = psm_ctx::RAW_ALL;   // vol[side] now holds real (filtered) data
            return check_launch(c, "cvf (fused, lazy costs)");
        }
        // producer/consumer kernel reading a materialised cost volume: out of place, all rows in one launch
        if (ensure_spare(c)) return 1;
        float *out = c->spare;
        {
            Prof p(c, PSM_K_CVF_F);
            launch_cvf_fused(c->stream, c->march, fv, out, c->g[side], W, H, c->Dloc, 0, H, c->g[1 - side].g1, c->d0, 0);
        }
        if (c->dtype == PSM_U8) {
            launch_f32_to_u8(c->stream, out, (uint8_t *)c->vol[side], V);   // q8 = sat_u8(rintf(q * 255))
        } else {
            c->spare = fv;          // ping-pong: the filtered volume becomes vol[side]
            c->vol[side] = out;
        }
        return check_launch(c, "cvf (fused)");
    }
    if (ensure_ab(c)) return 1;
    {
        Prof p(c, PSM_K_CVF_A);
        launch_cvf_a(c->stream, c->opt_variant, c->march, fv, c->ab, c->g[side], W, H, c->Dloc, 0, H);
    }
    if (stage_b) {
        {
            Prof p(c, PSM_K_CVF_B);
            launch_cvf_b(c->stream, c->opt_variant, c->march, c->ab, fv, c->g[side], W, H, c->Dloc, 0, H);
        }
        if (c->dtype == PSM_U8) launch_f32_to_u8(c->stream, fv, (uint8_t *)c->vol[side], V);
    }
    return check_launch(c, "cvf");
}

// 8-bit mode, storing form: float copy of the 8-bit cost volume -> fused filter (store mode, out of place) -> re-quantise
// (what psm_download_volume etc. see; the default path never runs it).  Needs the guidance of `side`.
int filter_u8_stored(psm_ctx *c, int side)
{
    const size_t V = (size_t)c->W * c->H * c->Dloc;
    if (!c->fvol) PSM_HIP(c, hipMalloc((void **)&c->fvol, V * sizeof(float)));
    if (ensure_spare(c)) return 1;
    launch_u8_to_f32(c->stream, (const uint8_t *)c->vol[side], c->fvol, V);
    {
        Prof p(c, PSM_K_CVF_F);
        launch_cvf_fused(c->stream, c->march, c->fvol, c->spare, c->g[side], c->W, c->H, c->Dloc, 0, c->H, c->g[1 - side].g1, c->d0, 0);
    }
    launch_f32_to_u8(c->stream, c->spare, (uint8_t *)c->vol[side], V);
    return check_launch(c, "cvf (8-bit, storing form)");
}

// Both volumes per launch: guidance of both images, select-mode fused filter of both volumes, chunk reduction of both - three
// launches per frame instead of six (shorter ramp / tail per launch; matters most for a disparity shard and for small
// images).  Only for the default path (float mode, costs built on the fly); flag 65536 turns it off.
static bool can_filter_both(const psm_ctx *c)
{
    return c->opt_variant == 0 && !(c->march.flags & (16 | 256 | 512 | 8192 | 16384 | 65536)) && c->H >= 8 &&
           c->raw_rows[0] != psm_ctx::RAW_ALL && c->raw_rows[1] != psm_ctx::RAW_ALL && !c->gf_virtual[0] && !c->gf_virtual[1] &&
           !c->fgf_virtual[0] && !c->fgf_virtual[1];
}

static int R_env_on()
{
    static const int r = getenv("PSM_PC_S0") ? atoi(getenv("PSM_PC_S0")) : 0;
    return r > 1 ? r : 0;
}

static int filter_both(psm_ctx *c)
{
    {   // g1 rows this launch reads: everything, or the stripe's rows - 8 .. + 8
        const bool striped = c->march.yend > c->march.ybeg;
        const int ya = striped ? (c->march.ybeg - 8 > 0 ? c->march.ybeg - 8 : 0) : 0;
        const int yb = striped ? (c->march.yend + 8 < c->H ? c->march.yend + 8 : c->H) : c->H;
        if (!c->have_g1 && !(c->g1_y1 > c->g1_y0 && c->g1_y0 <= ya && c->g1_y1 >= yb) && run_prep(c)) return 1;
    }
    if (!(c->have_guid[0] && c->have_guid[1])) {
        // a row stripe needs the guidance of its model rows only: y0 - 4 .. y1 + 2 (have_guid stays false: the planes are not
        // whole, any other consumer recomputes them; guid_y0/1 remember what is there for the next frame's check)
        const bool striped = c->march.yend > c->march.ybeg;
        const int gy0 = striped ? (c->march.ybeg - 4 > 0 ? c->march.ybeg - 4 : 0) : 0;
        const int gy1 = striped ? (c->march.yend + 4 < c->H ? c->march.yend + 4 : c->H) : c->H;
        if (!(c->guid_y1 > c->guid_y0 && c->guid_y0 <= gy0 && c->guid_y1 >= gy1)) {
            Prof p(c, PSM_K_GUIDE);
            launch_guidance(c->stream, c->g[0], nullptr, c->W, c->H, 0, &c->g[1], gy0, gy1);
            c->guid_y0 = gy0;
            c->guid_y1 = gy1;
            if (gy0 == 0 && gy1 == c->H) c->have_guid[0] = c->have_guid[1] = true;
        }
    }
    if (c->march.flags & 262144) {
        Prof p(c, PSM_K_CVF_F);
        launch_cvf_select_keys2(c->stream, c->march, c->g, c->W, c->H, c->Dloc, c->d0, c->keys_cur, c->dtype == PSM_U8 ? c->p4 : nullptr);
        c->gf_virtual[0] = c->gf_virtual[1] = true;
        return check_launch(c, "cvf (fused, select mode, shared keys, both volumes)");
    }
    const bool dynsel = (c->march.flags & 524288) != 0;
    // Two-phase selection (default from 112 local slices up - measured: -10 % at 1080p x 256, -13 % at 4K x 256, -4 % at
    // 720p x 128, worse at 64 slices and below; flag 1048576 forces it for any Dloc >= 2, flag 2097152 turns it off): every S-th slice goes through the
    // minima planes -> k_chunk_min -> keys; the other slices then run against that seeded key plane (MODE 2: one key load
    // per voxel, an atomic only where a slice beats the current minimum - rare after the seeding), so they write no
    // planes and need no reduction.  PSM_PC_S overrides S (default 5; 4 from 4 Mpixel up).
    static const int S_env = getenv("PSM_PC_S") ? atoi(getenv("PSM_PC_S")) : 0;
    // (measured, S = 4 / 5 / 6: 1080p x 256 7.60 / 7.20-7.36 / 7.43-7.60 ms, 4K x 256 28.6-29.4 / 30.0-30.8 / 29.6-30.4,
    // 720p x 128 1.97 / 1.90 / 1.90, 1/8 stripe of 1080p 1.08 / 1.10 / 1.10: the optimum moves with how the two launches fill
    // their rounds of resident workgroups)
    int S = S_env > 1 ? S_env : ((size_t)c->W * c->H >= ((size_t)1 << 22) ? 4 : 5);
    // ... so PSM_OPT_FLAGS 16777216 (for hosts that run many frames of one geometry) tunes it in place: the candidates 5, 4, 6
    // are timed twice each, round robin, on the frames 4-9 of a geometry (the first frames carry allocations and ramping
    // clocks) and the one with the fastest frame is kept - psm_debug_seed_stride reports it.  On the measured
    // configurations it settles on the static defaults above (1080p: 5, 4K: 4; 720p x 128 and stripes: 6, within 0.5 %).
    static const int TUNE_S[3] = {5, 4, 6};
    int tune_i = -1;
    psm_ctx::Tune &tn = c->tune;
    if (S_env <= 1 && (c->march.flags & 16777216) && R_env_on() == 0 && !dynsel && !(c->march.flags & 2097152) &&
        (c->Dloc >= 112 || (c->march.flags & 1048576)) && c->Dloc >= 2) {
        if (tn.W != c->W || tn.H != c->H || tn.D != c->Dloc || tn.y0 != c->march.ybeg || tn.y1 != c->march.yend || tn.dtype != c->dtype) {
            for (int i = 0; i < 3; ++i) {      // new geometry: start over (events still in flight are simply ignored)
                tn.ms[i] = -1.f;
                tn.n[i] = 0;
                tn.pend[i] = false;
            }
            tn.W = c->W; tn.H = c->H; tn.D = c->Dloc; tn.y0 = c->march.ybeg; tn.y1 = c->march.yend; tn.dtype = c->dtype;
            tn.calls = 0;
            tn.best = 0;
        }
        for (int i = 0; i < 3; ++i)
            if (tn.pend[i] && hipEventQuery(tn.e1[i]) == hipSuccess) {
                float ms = -1.f;
                if (hipEventElapsedTime(&ms, tn.e0[i], tn.e1[i]) != hipSuccess) ms = 1e30f;
                tn.ms[i] = (tn.n[i] == 0 || ms < tn.ms[i]) ? ms : tn.ms[i];
                tn.n[i] += 1;
                tn.pend[i] = false;
            }
        (void)hipGetLastError();               // (hipEventQuery reports "not ready" through the error state)
        const int TUNE_ROUNDS = 2, TUNE_SKIP = 3;   // frames before the first measurement: allocations, clocks still ramping up
        if (!tn.best && tn.n[0] >= TUNE_ROUNDS && tn.n[1] >= TUNE_ROUNDS && tn.n[2] >= TUNE_ROUNDS) {
            int b = 0;
            for (int i = 1; i < 3; ++i)
                if (tn.ms[i] < tn.ms[b]) b = i;
            tn.best = TUNE_S[b];
        }
        if (tn.best) S = tn.best;
        else if (++tn.calls > TUNE_SKIP) {
            // round robin 5, 4, 6, 5, 4, 6: the candidate with the fewest measurements (taken + in flight) next
            int fewest = 1 << 30;
            for (int i = 0; i < 3; ++i) {
                const int k = tn.n[i] + (tn.pend[i] ? 1 : 0);
                if (k < TUNE_ROUNDS && !tn.pend[i] && k < fewest) { fewest = k; tune_i = i; }
            }
            if (tune_i >= 0) {
                if (!tn.e0[tune_i] && (hipEventCreate(&tn.e0[tune_i]) != hipSuccess || hipEventCreate(&tn.e1[tune_i]) != hipSuccess)) {
                    (void)hipGetLastError();
                    tune_i = -1;               // no events: no tuning, the default stride stays
                    tn.best = S;
                } else {
                    S = TUNE_S[tune_i];
                    (void)hipEventRecord(tn.e0[tune_i], c->stream);
                }
            }
        }
    }
    const bool two_phase = !dynsel && !(c->march.flags & 2097152) && c->Dloc >= 2 && (c->Dloc >= 112 || (c->march.flags & 1048576));
    // Three phases (PSM_PC_S0 = r > 1, experiment): the seeding itself in two steps - every (S*r)-th slice through the planes,
    // then the other multiples of S against those few seeds (key form), then the rest.
    static const int R_env = getenv("PSM_PC_S0") ? atoi(getenv("PSM_PC_S0")) : 0;
    if (two_phase && R_env > 1 && c->Dloc > S * R_env) {
        const int R = R_env, D = c->Dloc;
        const int n0 = (D + S * R - 1) / (S * R);                  // multiples of S*R below D
        const int m1 = (D + S - 1) / S - 1;                        // k = 1..m1: S*k < D
        const int n1 = m1 - m1 / R;                                // ... that are not multiples of R
        const int n2 = D - (m1 + 1);                               // slices that are not multiples of S
        const PcPlan pl0 = pc_plan(c->W, c->march.rows(c->H), n0, c->march.seg_rows, 2);
        if (ensure_gf_scratch(c, 2 * pl0.scratch_bytes())) return 1;
        const uint8_t *const *p4 = c->dtype == PSM_U8 ? c->p4 : nullptr;
        {
            Prof p(c, PSM_K_CVF_F);
            launch_cvf_select2(c->stream, c->march, c->g, c->W, c->H, n0, c->d0, c->gf_scratch, nullptr, p4, 1, S * R);
        }
        {
            Prof p(c, PSM_K_WTA);
            launch_chunk_min2sides(c->stream, c->march, c->W, c->H, n0, c->gf_scratch, c->keys_cur, nullptr, 0);
        }
        if (n1 > 0) {
            Prof p(c, PSM_K_CVF_F);
            launch_cvf_select_keys2(c->stream, c->march, c->g, c->W, c->H, n1, c->d0, c->keys_cur, p4, 0, 2, R, S);
        }
        if (n2 > 0) {
            Prof p(c, PSM_K_CVF_F);
            launch_cvf_select_keys2(c->stream, c->march, c->g, c->W, c->H, n2, c->d0, c->keys_cur, p4, 0, 2, S, 1);
        }
        c->gf_virtual[0] = c->gf_virtual[1] = true;
        return check_launch(c, "cvf (fused, select mode, three phases, both volumes)");
    }
    if (two_phase) {
        const int n1 = (c->Dloc + S - 1) / S, n2 = c->Dloc - n1;
        const PcPlan pl1 = pc_plan(c->W, c->march.rows(c->H), n1, c->march.seg_rows, 2);
        {   // (sized for the smallest candidate stride as well: no reallocation - a pipeline stall - while tuning)
            size_t need = 2 * pl1.scratch_bytes();
            if (S_env <= 1 && c->Dloc >= 8) {
                const size_t n4 = 2 * pc_plan(c->W, c->march.rows(c->H), (c->Dloc + 3) / 4, c->march.seg_rows, 2).scratch_bytes();
                need = n4 > need ? n4 : need;
            }
            if (ensure_gf_scratch(c, need)) return 1;
        }
        {
            Prof p(c, PSM_K_CVF_F);
            launch_cvf_select2(c->stream, c->march, c->g, c->W, c->H, n1, c->d0, c->gf_scratch, nullptr, c->dtype == PSM_U8 ? c->p4 : nullptr, 1, S);
        }
        {
            Prof p(c, PSM_K_WTA);
            launch_chunk_min2sides(c->stream, c->march, c->W, c->H, n1, c->gf_scratch, c->keys_cur, nullptr, 0);
        }
        {
            Prof p(c, PSM_K_CVF_F);
            launch_cvf_select_keys2(c->stream, c->march, c->g, c->W, c->H, n2, c->d0, c->keys_cur, c->dtype == PSM_U8 ? c->p4 : nullptr, 0, 2, S);
        }
        if (tune_i >= 0) {
            (void)hipEventRecord(tn.e1[tune_i], c->stream);
            tn.pend[tune_i] = true;
        }
        c->gf_virtual[0] = c->gf_virtual[1] = true;
        return check_launch(c, "cvf (fused, select mode, two phases, both volumes)");
    }
    const PcPlan pl = pc_plan(c->W, c->march.rows(c->H), c->Dloc, c->march.seg_rows, dynsel ? 6 : 2);
    if (ensure_gf_scratch(c, 2 * pl.scratch_bytes())) return 1;
    if (dynsel && ensure_gf_cnt(c, 2 * (size_t)pl.ngroups * pl.nsegs)) return 1;
    {
        Prof p(c, PSM_K_CVF_F);
        launch_cvf_select2(c->stream, c->march, c->g, c->W, c->H, c->Dloc, c->d0, c->gf_scratch, dynsel ? c->gf_cnt : nullptr,
                           c->dtype == PSM_U8 ? c->p4 : nullptr);
    }
    {
        Prof p(c, PSM_K_WTA);
        launch_chunk_min2sides(c->stream, c->march, c->W, c->H, c->Dloc, c->gf_scratch, c->keys_cur, nullptr, dynsel);
    }
    c->gf_virtual[0] = c->gf_virtual[1] = true;
    return check_launch(c, "cvf (fused, select mode, both volumes)");
}

int psm_cost_filter(psm_ctx *c)
{
    if (!c) return 1;
    if (!c->have_cost) return fail(c, "psm_cost_filter: no cost volume (call psm_cost_construct or psm_upload_volume)");
    if (!c->have_images) return fail(c, "psm_cost_filter: no image pair uploaded (guidance)");
    if (bind(c)) return 1;
    const double t0 = now_us();
    // preprocess L, filter L, preprocess R, filter R (src/DispEst.cpp:302-305)
    const bool striped = c->march.yend > c->march.ybeg;
    if (can_filter_both(c)) {
        if (filter_both(c)) return 1;
    } else {
        if (striped) return fail(c, "psm_cost_filter: a row stripe (psm_set_rows) needs the default select form of the filter "
                                    "(no variant / storing / per-side flags, cost volumes not materialised)");
        for (int s = 0; s < 2; ++s)
            if (filter_side(c, s, true)) return 1;
    }
    c->have_maps = false;
    c->have_rows = striped;
    return end_stage(c, PSM_STAGE_CVF, t0);
}

int psm_cost_filter_side(psm_ctx *c, int side)
{
    if (!c) return 1;
    if (side != PSM_LEFT && side != PSM_RIGHT) return fail(c, "psm_cost_filter_side: bad side %d", side);
    if (!c->have_cost) return fail(c, "psm_cost_filter_side: no cost volume");
    if (!c->have_images) return fail(c, "psm_cost_filter_side: no image pair uploaded (guidance)");
    if (bind(c)) return 1;
    const double t0 = now_us();
    if (filter_side(c, side, true)) return 1;
    c->have_maps = false;
    if (!c->opt_async) PSM_HIP(c, hipStreamSynchronize(c->stream));
    c->stage_us[PSM_STAGE_CVF] = (side == PSM_LEFT ? 0.0 : c->stage_us[PSM_STAGE_CVF]) + (now_us() - t0);
    return 0;
}

int psm_cost_filter_fgf(psm_ctx *c, int sub)
{
    if (!c) return 1;
    if (c->dtype != PSM_F32) return fail(c, "psm_cost_filter_fgf: float contexts only");
    if (sub != 2 && sub != 4 && sub != 8) return fail(c, "psm_cost_filter_fgf: subsample_rate %d not in {2,4,8}", sub);
    if (!c->have_cost) return fail(c, "psm_cost_filter_fgf: no cost volume (call psm_cost_construct or psm_upload_volume)");
    if (!c->have_images) return fail(c, "psm_cost_filter_fgf: no image pair uploaded (guidance)");
    const int ws = c->W / sub, hs = c->H / sub, rad = 8 / sub;
    if (ws <= rad || hs <= rad) return fail(c, "psm_cost_filter_fgf: %dx%d too small for subsample_rate %d", c->W, c->H, sub);
    if (bind(c)) return 1;
    const double t0 = now_us();
    if (!c->have_g1 && run_prep(c)) return 1;
    // small planes: ism, msm, v1 (float4), v2 (float2) per pixel; ab (scratch) and one mab per side (float4) per small voxel
    const size_t n = (size_t)ws * hs, need = n * (3 * sizeof(float4) + sizeof(float2)) + 3 * n * c->Dloc * sizeof(float4);
    if (fgf_flush(c, 0) || fgf_flush(c, 1)) return 1;   // filtering an already FGF-filtered volume: make it real first
    for (int side = 0; side < 2; ++side)
        if (c->gf_virtual[side] && materialize(c, side)) return 1;
    if (c->fgf_bytes < need) {
        PSM_HIP(c, hipStreamSynchronize(c->stream));
        (void)hipFree(c->fgf);
        c->fgf = nullptr;
        c->fgf_bytes = 0;
        PSM_HIP(c, hipMalloc(&c->fgf, need));
        c->fgf_bytes = need;
    }
    float4 *ism = (float4 *)c->fgf, *msm = ism + n, *v1 = msm + n, *ab = v1 + n;
    c->fgf_mab[0] = ab + n * c->Dloc;
    c->fgf_mab[1] = c->fgf_mab[0] + n * c->Dloc;
    float2 *v2 = (float2 *)(c->fgf_mab[1] + n * c->Dloc);
    // flag 4096: always write the filtered volume (default: it stays virtual until something other than the WTA reads it)
    const bool keep_virtual = fgf_can_fuse_wta(c->W) && !(c->march.flags & 4096);
    // left volume with the left image as guidance, then the right one (src/DispEst.cpp:283-295)
    for (int side = 0; side < 2; ++side) {
        // a virtual (lazy) cost volume stays virtual: the filter samples 1/sub^2 of it straight from the g1 planes
        const int mode = c->raw_rows[side] == psm_ctx::RAW_ALL ? 0 : 1 + side;
        Prof p(c, PSM_K_FGF);
        launch_fgf_setup(c->stream, c->g[side].g1, c->W, c->H, sub, ism, msm, v1, v2);
        launch_fgf_model(c->stream, (const float *)c->vol[side], c->g[side].g1, c->g[1 - side].g1, c->W, c->H, c->Dloc, c->d0, sub, mode,
                         msm, v1, v2, ab, c->fgf_mab[side]);
        if (keep_virtual) c->fgf_virtual[side] = sub;
        else if (ensure_vol(c, side)) return 1;
        else launch_fgf_apply(c->stream, (float *)c->vol[side], c->g[side].g1, c->W, c->H, c->Dloc, sub, c->fgf_mab[side]);
        c->raw_rows[side] = psm_ctx::RAW_ALL;   // vol[side] holds (or, while virtual, stands for) filtered data
    }
    if (check_launch(c, "cvf (fast guided filter)")) return 1;
    c->have_maps = false;
    return end_stage(c, PSM_STAGE_CVF, t0);
}

int psm_filter_stage_a(psm_ctx *c, int side)
{
    if (!c) return 1;
    if (side != PSM_LEFT && side != PSM_RIGHT) return fail(c, "psm_filter_stage_a: bad side %d", side);
    if (!c->have_cost || !c->have_images) return fail(c, "psm_filter_stage_a: needs images and a cost volume");
    if (bind(c)) return 1;
    if (filter_side(c, side, false)) return 1;
    PSM_HIP(c, hipStreamSynchronize(c->stream));
    return 0;
}

static int copy_maps_out(psm_ctx *c, const uint8_t *dev, uint8_t *lmap, uint8_t *rmap, size_t stride)
{
    const size_t HW = (size_t)c->W * c->H;
    if (stride == 0) stride = c->W;
    if (stride < (size_t)c->W) return fail(c, "map stride %zu < width %d", stride, c->W);
    if (!lmap && !rmap) return 0;
    // device -> page-locked bounce buffer (one DMA at link speed) -> the caller's (pageable, possibly strided) rows.
    // A direct copy into pageable memory took 6-16 ms for two 1080p maps; this way it is ~0.5 ms.
    if (!c->pinned && hipHostMalloc((void **)&c->pinned, 2 * HW, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        c->pinned = nullptr;
    }
    if (!c->pinned) {   // no page-locked memory: plain copies
        if (lmap) PSM_HIP(c, hipMemcpy2DAsync(lmap, stride, dev, c->W, c->W, c->H, hipMemcpyDeviceToHost, c->stream));
        if (rmap) PSM_HIP(c, hipMemcpy2DAsync(rmap, stride, dev + HW, c->W, c->W, c->H, hipMemcpyDeviceToHost, c->stream));
        PSM_HIP(c, hipStreamSynchronize(c->stream));
        return 0;
    }
    if (lmap && rmap) PSM_HIP(c, hipMemcpyAsync(c->pinned, dev, 2 * HW, hipMemcpyDeviceToHost, c->stream));
    else if (lmap) PSM_HIP(c, hipMemcpyAsync(c->pinned, dev, HW, hipMemcpyDeviceToHost, c->stream));
    else PSM_HIP(c, hipMemcpyAsync(c->pinned + HW, dev + HW, HW, hipMemcpyDeviceToHost, c->stream));
    PSM_HIP(c, hipStreamSynchronize(c->stream));
    uint8_t *dst[2] = {lmap, rmap};
    for (int s2 = 0; s2 < 2; ++s2) {
        if (!dst[s2]) continue;
        const uint8_t *src = c->pinned + s2 * HW;
        if (stride == (size_t)c->W) memcpy(dst[s2], src, HW);
        else for (int y = 0; y < c->H; ++y) memcpy(dst[s2] + (size_t)y * stride, src + (size_t)y * c->W, (size_t)c->W);
    }
    return 0;
}

// WTA of one side into keys_s (may be NULL) and / or map_s (may be NULL).  A side whose Fast-Guided-Filter result is
// still virtual is selected straight from the smoothed models (upsample + linear model + argmin in one pass).
static int wta_side(psm_ctx *c, int s, long long *keys_s, uint8_t *map_s)
{
    const size_t HW = (size_t)c->W * c->H;
    Prof p(c, PSM_K_WTA);
    if (c->gf_virtual[s]) {
        // the select-mode filter already reduced this side: keys[s] holds the packed minima over the local slices
        const long long *src = c->keys_cur + s * HW;
        if (keys_s && keys_s != src) PSM_HIP(c, hipMemcpyAsync(keys_s, src, HW * sizeof(long long), hipMemcpyDeviceToDevice, c->stream));
        if (map_s) launch_merge(c->stream, src, HW, 1, (int)HW, map_s);
    } else if (c->fgf_virtual[s]) {
        long long *k = keys_s ? keys_s : c->keys_cur + s * HW;
        launch_fgf_apply_wta(c->stream, c->g[s].g1, c->W, c->H, c->Dloc, c->d0, c->fgf_virtual[s], c->fgf_mab[s], k);
        if (map_s) launch_merge(c->stream, k, HW, 1, (int)HW, map_s);
    } else if (c->dtype == PSM_U8) {
        launch_wta_u8(c->stream, (const uint8_t *)c->vol[s], c->W, c->H, c->d0, c->Dloc, keys_s, map_s);
    } else {
        launch_wta(c->stream, (const float *)c->vol[s], c->W, c->H, c->d0, c->Dloc, keys_s, map_s);
    }
    return 0;
}

static int wta_launch(psm_ctx *c, long long *keys, uint8_t *maps)
{
    const size_t HW = (size_t)c->W * c->H;
    if (c->gf_virtual[0] && c->gf_virtual[1] && !keys && maps) {   // both sides already reduced to keys: one launch for both maps
        Prof p(c, PSM_K_WTA);
        launch_merge(c->stream, c->keys_cur, 2 * HW, 1, (int)(2 * HW), maps);
        return check_launch(c, "wta");
    }
    for (int s = 0; s < 2; ++s)
        if (wta_side(c, s, keys ? keys + s * HW : nullptr, maps ? maps + s * HW : nullptr)) return 1;
    return check_launch(c, "wta");
}

// the volume a WTA is about to read: real data, or a virtual FGF result (consumed without materialising it)
static int wta_ready(psm_ctx *c, int side)
{
    return (c->fgf_virtual[side] || c->gf_virtual[side]) ? 0 : materialize(c, side);
}

int psm_disp_select(psm_ctx *c, uint8_t *lmap, uint8_t *rmap, size_t stride)
{
    if (!c) return 1;
    if (c->Dloc != c->D) return fail(c, "psm_disp_select: context holds slices [%d,%d) of %d; use psm_disp_select_partial + psm_disp_merge", c->d0, c->d1, c->D);
    if (!c->have_cost) return fail(c, "psm_disp_select: no cost volume");
    if (bind(c)) return 1;
    const double t0 = now_us();
    if (wta_ready(c, 0) || wta_ready(c, 1)) return 1;
    if (wta_launch(c, nullptr, c->maps)) return 1;
    c->have_maps = true;
    c->have_valid = false;
    if (copy_maps_out(c, c->maps, lmap, rmap, stride)) return 1;
    return end_stage(c, PSM_STAGE_DISPSEL, t0);
}

int psm_disp_select_partial(psm_ctx *c, void *dev_keys)
{
    if (!c) return 1;
    if (!c->have_cost) return fail(c, "psm_disp_select_partial: no cost volume");
    if (bind(c)) return 1;
    const double t0 = now_us();
    if (wta_ready(c, 0) || wta_ready(c, 1)) return 1;
    if (wta_launch(c, dev_keys ? (long long *)dev_keys : c->keys_cur, nullptr)) return 1;
    if (!dev_keys) c->have_keys = c->have_keys_side[0] = c->have_keys_side[1] = true;
    return end_stage(c, PSM_STAGE_DISPSEL, t0);
}

int psm_disp_select_partial_side(psm_ctx *c, int side, void *dev_keys_side)
{
    if (!c) return 1;
    if (side != PSM_LEFT && side != PSM_RIGHT) return fail(c, "psm_disp_select_partial_side: bad side %d", side);
    if (!c->have_cost) return fail(c, "psm_disp_select_partial_side: no cost volume");
    if (bind(c)) return 1;
    const double t0 = now_us();
    if (wta_ready(c, side)) return 1;
    const size_t HW = (size_t)c->W * c->H;
    long long *keys = dev_keys_side ? (long long *)dev_keys_side : c->keys_cur + side * HW;
    if (wta_side(c, side, keys, nullptr)) return 1;
    if (check_launch(c, "wta")) return 1;
    if (!dev_keys_side) {
        c->have_keys_side[side] = true;
        c->have_keys = c->have_keys_side[0] && c->have_keys_side[1];
    }
    if (!c->opt_async) PSM_HIP(c, hipStreamSynchronize(c->stream));
    c->stage_us[PSM_STAGE_DISPSEL] = (side == PSM_LEFT ? 0.0 : c->stage_us[PSM_STAGE_DISPSEL]) + (now_us() - t0);
    return 0;
}

int psm_debug_seed_stride(psm_ctx *c)
{   // stride of the seeding phase the tuner settled on for the current geometry; 0 while it is still measuring / not in use
    return c ? c->tune.best : 0;
}

int psm_set_rows(psm_ctx *c, int y_begin, int y_end)
{
    if (!c) return 1;
    // (takes effect with the next psm_cost_filter; minima / maps already computed keep describing the stripe they were made for)
    if (y_begin == 0 && (y_end == 0 || y_end == c->H)) {   // whole image
        c->march.ybeg = c->march.yend = 0;
        return 0;
    }
    if (y_begin < 0 || y_end > c->H || y_begin >= y_end) return fail(c, "psm_set_rows: bad stripe [%d,%d) of %d rows", y_begin, y_end, c->H);
    c->march.ybeg = y_begin;
    c->march.yend = y_end;
    return 0;
}

int psm_set_map_buffer(psm_ctx *c, void *dev_maps, int whole)
{
    if (!c) return 1;
    uint8_t *m = dev_maps ? (uint8_t *)dev_maps : c->maps_own;
    if (m != c->maps) { c->have_maps = false; c->have_valid = false; }
    c->maps = m;
    if (whole) {    // the caller filled the buffer with both complete maps of the current frame (e.g. gathered row stripes)
        c->have_maps = true;
        c->have_rows = false;
        c->have_valid = false;
    }
    return 0;
}

int psm_gather_rows_ctx(psm_ctx *root, psm_ctx *const *stripes, int nstripes, uint8_t *lmap, uint8_t *rmap, size_t stride)
{
    if (!root) return 1;
    if (!stripes || nstripes < 1) return fail(root, "psm_gather_rows_ctx: bad arguments");
    std::vector<char> covered((size_t)root->H, 0);
    for (int i = 0; i < nstripes; ++i) {
        const psm_ctx *s = stripes[i];
        if (!s || s->W != root->W || s->H != root->H || s->D != root->D || s->dtype != root->dtype)
            return fail(root, "psm_gather_rows_ctx: stripe %d does not belong to this job", i);
        if (!s->have_maps) return fail(root, "psm_gather_rows_ctx: stripe %d has no maps for this frame (call psm_disp_select first)", i);
        for (int y = s->march.y0(s->H); y < s->march.y1(s->H); ++y) {
            if (covered[y]) return fail(root, "psm_gather_rows_ctx: row %d is held by more than one stripe", y);
            covered[y] = 1;
        }
    }
    for (int y = 0; y < root->H; ++y)
        if (!covered[y]) return fail(root, "psm_gather_rows_ctx: no stripe holds row %d", y);
    if (bind(root)) return 1;
    const size_t HW = (size_t)root->W * root->H;
    for (int i = 0; i < nstripes; ++i) {
        psm_ctx *s = stripes[i];
        if (s == root) continue;
        (void)hipSetDevice(s->device);
        PSM_HIP(root, hipStreamSynchronize(s->stream));     // the stripe's maps must be complete before they are read
        (void)hipSetDevice(root->device);
        const size_t o = (size_t)s->march.y0(s->H) * root->W, n = (size_t)s->march.rows(s->H) * root->W;
        for (int side = 0; side < 2; ++side) {
            if (s->device == root->device)
                PSM_HIP(root, hipMemcpyAsync(root->maps + side * HW + o, s->maps + side * HW + o, n, hipMemcpyDeviceToDevice, root->stream));
            else
                PSM_HIP(root, hipMemcpyPeerAsync(root->maps + side * HW + o, root->device, s->maps + side * HW + o, s->device, n, root->stream));
        }
    }
    root->have_maps = true;
    root->have_rows = false;      // the root's maps are whole now
    root->have_valid = false;
    if (copy_maps_out(root, root->maps, lmap, rmap, stride)) return 1;
    if (!root->opt_async) PSM_HIP(root, hipStreamSynchronize(root->stream));
    return 0;
}

int psm_set_key_buffer(psm_ctx *c, void *dev_keys)
{
    if (!c) return 1;
    long long *k = dev_keys ? (long long *)dev_keys : c->keys;
    if (k != c->keys_cur && (c->gf_virtual[0] || c->gf_virtual[1]))
        return fail(c, "psm_set_key_buffer: the current minima are still pending in the previous buffer (call before psm_cost_filter)");
    c->keys_cur = k;
    return 0;
}

int psm_partial_keys(psm_ctx *c, void **dev_keys, size_t *bytes)
{
    if (!c) return 1;
    if (dev_keys) *dev_keys = c->keys_cur;
    if (bytes) *bytes = 2 * (size_t)c->W * c->H * sizeof(long long);
    return 0;
}

int psm_disp_merge(psm_ctx *c, const void *dev_keys_all, int nranks, uint8_t *lmap, uint8_t *rmap, size_t stride)
{
    if (!c) return 1;
    if (!dev_keys_all || nranks < 1) return fail(c, "psm_disp_merge: bad arguments");
    if (bind(c)) return 1;
    const double t0 = now_us();
    const size_t n = 2 * (size_t)c->W * c->H;
    {
        Prof p(c, PSM_K_MERGE);
        launch_merge(c->stream, (const long long *)dev_keys_all, n, nranks, (int)n, c->maps);
    }
    if (check_launch(c, "merge")) return 1;
    c->have_maps = true;
    c->have_valid = false;   // new maps: a validity mask of an earlier frame does not describe them
    if (copy_maps_out(c, c->maps, lmap, rmap, stride)) return 1;
    if (!c->opt_async) PSM_HIP(c, hipStreamSynchronize(c->stream));
    c->stage_us[PSM_STAGE_DISPSEL] += now_us() - t0;
    return 0;
}

int psm_disp_merge_ctx(psm_ctx *root, psm_ctx *const *shards, int nshards, uint8_t *lmap, uint8_t *rmap, size_t stride)
{
    if (!root) return 1;
    if (!shards || nshards < 1) return fail(root, "psm_disp_merge_ctx: bad arguments");
    const size_t bytes = 2 * (size_t)root->W * root->H * sizeof(long long);
    std::vector<char> covered((size_t)root->D, 0);
    for (int i = 0; i < nshards; ++i) {
        const psm_ctx *s = shards[i];
        if (!s || s->W != root->W || s->H != root->H || s->D != root->D || s->dtype != root->dtype)
            return fail(root, "psm_disp_merge_ctx: shard %d does not belong to this job", i);
        if (!s->have_keys)
            return fail(root, "psm_disp_merge_ctx: shard %d has no partial minima for this frame (call psm_disp_select_partial(ctx, NULL) first)", i);
        for (int d = s->d0; d < s->d1; ++d) {
            if (covered[d]) return fail(root, "psm_disp_merge_ctx: slice %d is held by more than one shard", d);
            covered[d] = 1;
        }
    }
    for (int d = 0; d < root->D; ++d)
        if (!covered[d]) return fail(root, "psm_disp_merge_ctx: no shard holds slice %d", d);
    if (bind(root)) return 1;
    if (root->gather_ranks < nshards) {
        PSM_HIP(root, hipStreamSynchronize(root->stream));
        (void)hipFree(root->gather);
        root->gather = nullptr;
        root->gather_ranks = 0;
        PSM_HIP(root, hipMalloc((void **)&root->gather, bytes * nshards));
        root->gather_ranks = nshards;
    }
    for (int i = 0; i < nshards; ++i) {
        psm_ctx *s = shards[i];
        // the shard's partial WTA must have finished before its keys are read
        (void)hipSetDevice(s->device);
        PSM_HIP(root, hipStreamSynchronize(s->stream));
        (void)hipSetDevice(root->device);
        if (s->device == root->device)
            PSM_HIP(root, hipMemcpyAsync((char *)root->gather + bytes * i, s->keys_cur, bytes, hipMemcpyDeviceToDevice, root->stream));
        else
            PSM_HIP(root, hipMemcpyPeerAsync((char *)root->gather + bytes * i, root->device, s->keys_cur, s->device, bytes, root->stream));
    }
    return psm_disp_merge(root, root->gather, nshards, lmap, rmap, stride);
}

int psm_download_maps(psm_ctx *c, uint8_t *lmap, uint8_t *rmap, size_t stride)
{
    if (!c) return 1;
    if (!c->have_maps) return fail(c, "psm_download_maps: no disparity maps computed");
    if (bind(c)) return 1;
    return copy_maps_out(c, c->maps, lmap, rmap, stride);
}

int psm_lr_check(psm_ctx *c, uint8_t *lvalid, uint8_t *rvalid, size_t stride)
{
    if (!c) return 1;
    if (!c->have_maps) return fail(c, "psm_lr_check: no disparity maps computed");
    if (c->have_rows) return fail(c, "psm_lr_check: the maps hold this context's row stripe only (gather the stripes first)");
    if (bind(c)) return 1;
    const double t0 = now_us();
    const size_t HW = (size_t)c->W * c->H;
    {
        Prof p(c, PSM_K_LRC);
        launch_lr_check(c->stream, c->maps, c->maps + HW, c->W, c->H, c->valid, c->valid + HW);
    }
    if (check_launch(c, "lr_check")) return 1;
    c->have_valid = true;
    if (copy_maps_out(c, c->valid, lvalid, rvalid, stride)) return 1;
    return end_stage(c, PSM_STAGE_PP, t0);
}

int psm_fill_invalid(psm_ctx *c, uint8_t *lmap, uint8_t *rmap, size_t stride)
{
    if (!c) return 1;
    if (!c->have_maps || !c->have_valid) return fail(c, "psm_fill_invalid: needs disparity maps and psm_lr_check");
    if (bind(c)) return 1;
    const double t0 = now_us();
    const size_t HW = (size_t)c->W * c->H;
    {
        Prof p(c, PSM_K_LRC);
        launch_fill_inv(c->stream, c->maps, c->valid, c->W, c->H);
        launch_fill_inv(c->stream, c->maps + HW, c->valid + HW, c->W, c->H);
    }
    if (check_launch(c, "fill_inv")) return 1;
    // have_valid stays set: the mask still says which pixels the L-R check rejected, which is what the next stage of
    // PP::processDM (wgtMedian, src/PP.cpp:405-410) filters
    if (copy_maps_out(c, c->maps, lmap, rmap, stride)) return 1;
    if (!c->opt_async) PSM_HIP(c, hipStreamSynchronize(c->stream));
    c->stage_us[PSM_STAGE_PP] += now_us() - t0;
    return 0;
}

// The row-dataflow form of wgtMedian: exact for any input, but as sequential as the reference wherever invalid pixels chain
static int wgt_median_dataflow(psm_ctx *c, int side)
{
    const size_t HW = (size_t)c->W * c->H, nn = (size_t)c->H * (c->W + 1);
    if (!c->wm) PSM_HIP(c, hipMalloc((void **)&c->wm, (nn + c->H + 1) * sizeof(int)));
    int *nxt = c->wm, *prog = c->wm + nn, *err = prog + c->H;
    PSM_HIP(c, hipMemsetAsync(err, 0, sizeof(int), c->stream));
    {
        Prof p(c, PSM_K_WMF);
        launch_wgt_median(c->stream, c->maps + side * HW, c->valid + side * HW, c->g[side].g1, c->W, c->H, c->D, side, nxt, prog, err);
    }
    if (check_launch(c, "wgt_median")) return 1;
    int herr = 0;
    PSM_HIP(c, hipMemcpyAsync(&herr, err, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    PSM_HIP(c, hipStreamSynchronize(c->stream));
    if (herr) return fail(c, "psm_wgt_median: row pipeline stalled (watchdog); maps are not valid");
    c->wm_sweeps[side] = -1;
    c->wm_evals[side] = 0;
    return 0;
}

int psm_wgt_median(psm_ctx *c, uint8_t *lmap, uint8_t *rmap, size_t stride)
{
    if (!c) return 1;
    if (!c->have_maps || !c->have_valid) return fail(c, "psm_wgt_median: needs disparity maps and psm_lr_check");
    if (!c->have_images) return fail(c, "psm_wgt_median: no image pair uploaded (colour weights)");
    if (c->W < 9 || c->H < 9) return fail(c, "psm_wgt_median: image %dx%d smaller than the 19x19 window's wrap allows", c->W, c->H);
    if (bind(c)) return 1;
    const double t0 = now_us();
    if (!c->have_g1 && run_prep(c)) return 1;
    const size_t HW = (size_t)c->W * c->H;
    // PSM_OPT_FLAGS 4194304: dataflow form only; 8388608: at most 2 sweeps (test hook for the fall-back)
    const bool dataflow_only = (c->march.flags & 4194304) != 0;
    const int CAP = (c->march.flags & 8388608) ? 2 : 96, CHK = 4;
    bool done[2] = {false, false};
    if (!dataflow_only) {
        // parallel form: sweeps to the fixed point of the in-place recursion (psm_pp.hip), both maps side by side
        const size_t nb = (HW + 255) / 256 * 256, ncnt = 2 * (size_t)(96 + 2);
        const size_t per_side = 2 * nb + (4 * nb + ncnt) * sizeof(int);
        if (!c->wm_par) PSM_HIP(c, hipMalloc((void **)&c->wm_par, 2 * per_side));
        uint8_t *orig[2], *newv[2];
        int *stamp[2], *list[2][2], *chg[2], *cnt[2];
        for (int s = 0; s < 2; ++s) {
            uint8_t *b = c->wm_par + s * per_side;
            orig[s] = b; newv[s] = b + nb;
            int *ip = reinterpret_cast<int *>(b + 2 * nb);
            stamp[s] = ip; list[s][0] = ip + nb; list[s][1] = ip + 2 * nb; chg[s] = ip + 3 * nb; cnt[s] = ip + 4 * nb;
            PSM_HIP(c, hipMemcpyAsync(orig[s], c->maps + s * HW, HW, hipMemcpyDeviceToDevice, c->stream));
            PSM_HIP(c, hipMemsetAsync(stamp[s], 0, nb * sizeof(int), c->stream));
            PSM_HIP(c, hipMemsetAsync(cnt[s], 0, ncnt * sizeof(int), c->stream));
            launch_wm_seed(c->stream, c->valid + s * HW, c->W, c->H, list[s][0], cnt[s]);
        }
        std::vector<int> hc(2 * ncnt);
        int sw = 0;
        while (sw < CAP && !(done[0] && done[1])) {
            const int upto = sw + CHK < CAP ? sw + CHK : CAP;
            {
                Prof p(c, PSM_K_WMF);
                for (; sw < upto; ++sw)
                    for (int s = 0; s < 2; ++s)
                        if (!done[s])
                            launch_wm_sweep(c->stream, c->maps + s * HW, orig[s], c->valid + s * HW, c->g[s].g1, c->W, c->H, c->D, s,
                                            list[s][sw & 1], cnt[s] + 2 * sw, newv[s], chg[s], cnt[s] + 2 * sw + 1, stamp[s], sw + 1,
                                            list[s][(sw + 1) & 1], cnt[s] + 2 * (sw + 1));
            }
            if (check_launch(c, "wgt_median (sweeps)")) return 1;
            for (int s = 0; s < 2; ++s)
                PSM_HIP(c, hipMemcpyAsync(hc.data() + s * ncnt, cnt[s], ncnt * sizeof(int), hipMemcpyDeviceToHost, c->stream));
            PSM_HIP(c, hipStreamSynchronize(c->stream));
            for (int s = 0; s < 2; ++s) {
                if (done[s]) continue;
                long long ev = 0;
                for (int k = 0; k < sw; ++k) {
                    ev += hc[s * ncnt + 2 * k];
                    if (hc[s * ncnt + 2 * k + 1] == 0) {      // sweep k changed nothing: fixed point
                        done[s] = true;
                        c->wm_sweeps[s] = k + 1;
                        c->wm_evals[s] = ev;
                        break;
                    }
                }
            }
        }
        // no fixed point within CAP sweeps (long chains of pixels that keep flipping each other): start over from the input
        // with the dataflow form, which is exact for any input
        for (int s = 0; s < 2; ++s)
            if (!done[s]) PSM_HIP(c, hipMemcpyAsync(c->maps + s * HW, orig[s], HW, hipMemcpyDeviceToDevice, c->stream));
    }
    for (int s = 0; s < 2; ++s)
        if (!done[s] && wgt_median_dataflow(c, s)) return 1;
    if (copy_maps_out(c, c->maps, lmap, rmap, stride)) return 1;
    c->stage_us[PSM_STAGE_PP] += now_us() - t0;
    return 0;
}

int psm_wgt_median_stats(psm_ctx *c, int *sweeps, long long *evals)
{
    if (!c) return 1;
    for (int s = 0; s < 2; ++s) {
        if (sweeps) sweeps[s] = c->wm_sweeps[s];
        if (evals) evals[s] = c->wm_evals[s];
    }
    return 0;
}

int psm_upload_maps(psm_ctx *c, const uint8_t *lmap, const uint8_t *rmap, const uint8_t *lvalid, const uint8_t *rvalid, size_t stride)
{
    if (!c) return 1;
    if (stride == 0) stride = c->W;
    if (stride < (size_t)c->W) return fail(c, "psm_upload_maps: stride %zu < width %d", stride, c->W);
    if (bind(c)) return 1;
    const size_t HW = (size_t)c->W * c->H;
    const uint8_t *src[4] = {lmap, rmap, lvalid, rvalid};
    uint8_t *dst[4] = {c->maps, c->maps + HW, c->valid, c->valid + HW};
    for (int i = 0; i < 4; ++i)
        if (src[i]) PSM_HIP(c, hipMemcpy2DAsync(dst[i], c->W, src[i], stride, c->W, c->H, hipMemcpyHostToDevice, c->stream));
    PSM_HIP(c, hipStreamSynchronize(c->stream));
    if (lmap && rmap) { c->have_maps = true; c->have_valid = false; }
    if (lvalid && rvalid && c->have_maps) c->have_valid = true;
    return 0;
}

static int check_slices(psm_ctx *c, const char *who, int side, int d0, int d1)
{
    if (side != PSM_LEFT && side != PSM_RIGHT) return fail(c, "%s: bad side %d", who, side);
    if (d0 < c->d0 || d1 > c->d1 || d0 >= d1) return fail(c, "%s: slices [%d,%d) not inside this context's [%d,%d)", who, d0, d1, c->d0, c->d1);
    return 0;
}

int psm_download_volume(psm_ctx *c, int side, int d0, int d1, void *host)
{
    if (!c || !host) return 1;
    if (check_slices(c, "psm_download_volume", side, d0, d1)) return 1;
    if (bind(c)) return 1;
    if (!c->have_cost) return fail(c, "psm_download_volume: no cost volume");
    if (materialize(c, side)) return 1;
    const size_t S = (size_t)c->W * c->H * velem(c);
    PSM_HIP(c, hipMemcpyAsync(host, (const char *)c->vol[side] + (size_t)(d0 - c->d0) * S, (size_t)(d1 - d0) * S, hipMemcpyDeviceToHost, c->stream));
    PSM_HIP(c, hipStreamSynchronize(c->stream));
    return 0;
}

int psm_upload_volume(psm_ctx *c, int side, int d0, int d1, const void *host)
{
    if (!c || !host) return 1;
    if (check_slices(c, "psm_upload_volume", side, d0, d1)) return 1;
    if (bind(c)) return 1;
    if (c->have_cost && c->have_g1 && materialize(c, side)) return 1;   // a partial upload must not leave virtual slices
    if (fgf_flush(c, side)) return 1;
    c->gf_virtual[side] = false;
    if (ensure_vol(c, side)) return 1;
    const size_t S = (size_t)c->W * c->H * velem(c);
    PSM_HIP(c, hipMemcpyAsync((char *)c->vol[side] + (size_t)(d0 - c->d0) * S, host, (size_t)(d1 - d0) * S, hipMemcpyHostToDevice, c->stream));
    PSM_HIP(c, hipStreamSynchronize(c->stream));
    c->have_cost = true;
    c->have_maps = false;
    c->raw_rows[side] = psm_ctx::RAW_ALL;
    return 0;
}

int psm_download_ab(psm_ctx *c, int d0, int d1, float *host)
{
    if (!c || !host) return 1;
    if (check_slices(c, "psm_download_ab", 0, d0, d1)) return 1;
    if (bind(c)) return 1;
    if (!c->ab) return fail(c, "psm_download_ab: no stage-A result (call psm_filter_stage_a first)");
    const size_t S = (size_t)c->W * c->H * sizeof(float4);
    PSM_HIP(c, hipMemcpyAsync(host, (const char *)c->ab + (size_t)(d0 - c->d0) * S, (size_t)(d1 - d0) * S, hipMemcpyDeviceToHost, c->stream));
    PSM_HIP(c, hipStreamSynchronize(c->stream));
    return 0;
}

int psm_download_guidance(psm_ctx *c, int side, float *host)
{
    if (!c || !host) return 1;
    if (side != PSM_LEFT && side != PSM_RIGHT) return fail(c, "psm_download_guidance: bad side %d", side);
    if (bind(c)) return 1;
    const size_t HW = (size_t)c->W * c->H;
    std::vector<float4> b1(HW), b2(HW), b3(HW);
    std::vector<float2> b4(HW);
    PSM_HIP(c, hipStreamSynchronize(c->stream));
    PSM_HIP(c, hipMemcpy(b1.data(), c->g[side].g1, HW * sizeof(float4), hipMemcpyDeviceToHost));
    PSM_HIP(c, hipMemcpy(b2.data(), c->g[side].g2, HW * sizeof(float4), hipMemcpyDeviceToHost));
    PSM_HIP(c, hipMemcpy(b3.data(), c->g[side].g3, HW * sizeof(float4), hipMemcpyDeviceToHost));
    PSM_HIP(c, hipMemcpy(b4.data(), c->g[side].g4, HW * sizeof(float2), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < HW; ++i) {
        host[0 * HW + i] = b1[i].x; host[1 * HW + i] = b1[i].y; host[2 * HW + i] = b1[i].z; host[3 * HW + i] = b1[i].w;
        host[4 * HW + i] = b2[i].x; host[5 * HW + i] = b2[i].y; host[6 * HW + i] = b2[i].z; host[7 * HW + i] = b2[i].w;
        host[8 * HW + i] = b3[i].x; host[9 * HW + i] = b3[i].y; host[10 * HW + i] = b3[i].z; host[11 * HW + i] = b3[i].w;
        host[12 * HW + i] = b4[i].x; host[13 * HW + i] = b4[i].y;
    }
    return 0;
}

int psm_box8_volume(psm_ctx *c, int side, float *host)
{
    if (!c) return 1;
    if (side != PSM_LEFT && side != PSM_RIGHT) return fail(c, "psm_box8_volume: bad side %d", side);
    if (c->dtype != PSM_F32) return fail(c, "psm_box8_volume: float mode only");
    if (!c->have_cost) return fail(c, "psm_box8_volume: no cost volume");
    if (bind(c)) return 1;
    if (materialize(c, side) || ensure_ab(c)) return 1;
    {
        Prof p(c, PSM_K_BOX);
        launch_box8(c->stream, c->opt_variant, c->march, (const float *)c->vol[side], (float *)c->ab, c->W, c->H, c->Dloc);
    }
    if (check_launch(c, "box8")) return 1;
    if (host) {
        const size_t V = (size_t)c->W * c->H * c->Dloc;
        PSM_HIP(c, hipMemcpyAsync(host, c->ab, V * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    }
    if (host || !c->opt_async) PSM_HIP(c, hipStreamSynchronize(c->stream));
    return 0;
}

int psm_stage_time_us(psm_ctx *c, int stage, double *us)
{
    if (!c || !us || stage < 0 || stage >= PSM_STAGE_COUNT) return 1;
    *us = c->stage_us[stage];
    return 0;
}

int psm_kernel_time_ms(psm_ctx *c, int kernel, double *total_ms, int *launches)
{
    if (!c || kernel < 0 || kernel >= PSM_K_COUNT) return 1;
    if (bind(c)) return 1;
    if (flush_timers(c)) return 1;
    if (total_ms) *total_ms = c->timers[kernel].total_ms;
    if (launches) *launches = c->timers[kernel].launches;
    return 0;
}

int psm_reset_kernel_times(psm_ctx *c)
{
    if (!c) return 1;
    if (bind(c)) return 1;
    if (flush_timers(c)) return 1;
    for (auto &t : c->timers) {
        t.total_ms = 0.0;
        t.launches = 0;
    }
    return 0;
}

}  // extern "C"
