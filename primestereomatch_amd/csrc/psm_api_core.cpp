// psm_api_core.cpp - C ABI of libprimesm_hip.so (include/primesm_hip.h): context life cycle, options, the PCIe legs
// (image upload, volume / guidance downloads), stage and kernel timers.  Replaces createContext / createCommandQueue /
// clCreateBuffer and the three `_cl` constructors (src/DispEst.cpp:57-140) and the host half of CVC_cl::buildCV
// (src/CVC_cl.cpp:95-160).
#include "psm_ctx.h"

#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>

using namespace psm;

namespace {
std::string g_create_error;
}

namespace psm {

double now_us()
{
    using namespace std::chrono;
    return duration_cast<duration<double, std::micro>>(steady_clock::now().time_since_epoch()).count();
}

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


// a new image pair is current: nothing derived from the previous one survives
void adopt_new_pair(psm_ctx *c, int depth)
{
    c->raw_depth = depth;
    c->have_images = true;
    c->have_g1 = false;
    c->g1_y0 = c->g1_y1 = 0;
    c->have_guid[0] = c->have_guid[1] = false;
    c->guid_y0 = c->guid_y1 = 0;
    c->have_cost = false;
    c->have_maps = false;
    c->have_valid = false;
    c->have_keys = c->have_keys_side[0] = c->have_keys_side[1] = false;
    c->raw_rows[0] = c->raw_rows[1] = psm_ctx::RAW_ALL;   // nothing virtual survives a new pair
    c->fgf_virtual[0] = c->fgf_virtual[1] = 0;
    c->gf_virtual[0] = c->gf_virtual[1] = false;
    c->maps_early = nullptr;
}

}  // namespace psm

namespace {

void free_all(psm_ctx *c)
{
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    if (c->down_stream) (void)hipStreamSynchronize(c->down_stream);
    for (int s = 0; s < 2; ++s) {
        (void)hipFree(c->raw[s]);
        (void)hipFree(c->raw_next[s]);
        (void)hipFree(c->g[s].g1);
        (void)hipFree(c->g[s].g2);
        (void)hipFree(c->g[s].g3);
        (void)hipFree(c->g[s].g4);
        (void)hipFree(c->vol[s]);
        (void)hipFree(c->p4[s]);
    }
    (void)hipFree(c->fvol);
    (void)hipFree(c->spare);
    (void)hipFree(c->ab);
    (void)hipFree(c->keys);
    (void)hipFree(c->gather);
    (void)hipFree(c->maps_own);
    (void)hipFree(c->valid);
    for (int k = 0; k < 2; ++k) {
        if (c->xfer_pin[k]) (void)hipHostFree(c->xfer_pin[k]);
        if (c->ev_xfer[k]) (void)hipEventDestroy(c->ev_xfer[k]);
    }
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->pinned2) (void)hipHostFree(c->pinned2);
    if (c->pin_up) (void)hipHostFree(c->pin_up);
    (void)hipFree(c->wm);
    (void)hipFree(c->wm_par);
    (void)hipFree(c->wm_wts);
    if (c->wm_pin) (void)hipHostFree(c->wm_pin);
    for (hipEvent_t e : {c->ev_wm[0], c->ev_wm[1]})
        if (e) (void)hipEventDestroy(e);
    (void)hipFree(c->gf_scratch);
    (void)hipFree(c->pc_ts);
    (void)hipFree(c->fgf);
    if (c->batch_graph) (void)hipGraphExecDestroy(c->batch_graph);
    (void)hipFree(c->range_dev);
    if (c->range_pin) (void)hipHostFree(c->range_pin);
    (void)hipFree(c->batch_tab);
    if (c->batch_pin) (void)hipHostFree(c->batch_pin);
    for (hipEvent_t e : {c->ev_batch, c->ev_tab[0], c->ev_tab[1]})
        if (e) (void)hipEventDestroy(e);
    for (auto &t : c->timers)
        for (auto &p : t.pending) {
            (void)hipEventDestroy(p.first);
            (void)hipEventDestroy(p.second);
        }
    for (auto e : c->event_pool) (void)hipEventDestroy(e);
    for (hipEvent_t e : {c->ev_up, c->ev_maps, c->ev_down, c->ev_free, c->ev_stage[0], c->ev_stage[1]})
        if (e) (void)hipEventDestroy(e);
    if (c->shared) {
        if (--c->shared->refs == 0) {
            (void)hipStreamDestroy(c->shared->main); (void)hipStreamDestroy(c->shared->up); (void)hipStreamDestroy(c->shared->down);
            delete c->shared;
        }
    } else if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
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

int check_pair_args(psm_ctx *c, const char *who, const void *l, const void *r, int channels, size_t *stride_bytes, int depth, size_t *row)
{
    if (!l || !r) return fail(c, "%s: NULL image", who);
    if (channels != 3) return fail(c, "%s: %d channels (3 required, B,G,R interleaved)", who, channels);
    if (depth != PSM_IMG_U8 && depth != PSM_IMG_F32) return fail(c, "%s: unknown depth %d", who, depth);
    if (c->dtype == PSM_U8 && depth != PSM_IMG_U8) return fail(c, "%s: 8-bit mode needs 8-bit images", who);
    *row = (size_t)c->W * 3 * (depth == PSM_IMG_F32 ? 4 : 1);
    if (*stride_bytes == 0) *stride_bytes = *row;
    if (*stride_bytes < *row) return fail(c, "%s: stride %zu < row size %zu", who, *stride_bytes, *row);
    return 0;
}

int check_slices(psm_ctx *c, const char *who, int side, int d0, int d1)
{
    if (side != PSM_LEFT && side != PSM_RIGHT) return fail(c, "%s: bad side %d", who, side);
    if (d0 < c->d0 || d1 > c->d1 || d0 >= d1) return fail(c, "%s: slices [%d,%d) not inside this context's [%d,%d)", who, d0, d1, c->d0, c->d1);
    return 0;
}

}  // namespace

namespace psm {

int range_enqueue(psm_ctx *c, hipStream_t stream, int slot, const float *p0, size_t n0, const float *p1, size_t n1)
{
    if (!c->range_dev) {
        PSM_HIP(c, hipMalloc((void **)&c->range_dev, 8 * sizeof(unsigned)));
        PSM_HIP(c, hipHostMalloc((void **)&c->range_pin, 8 * sizeof(unsigned), hipHostMallocDefault));
    }
    c->range_pin[2 * slot] = 0u; c->range_pin[2 * slot + 1] = 255u;
    PSM_HIP(c, hipMemsetD32Async((hipDeviceptr_t)(c->range_dev + 2 * slot), 0, 1, stream));
    PSM_HIP(c, hipMemsetD32Async((hipDeviceptr_t)(c->range_dev + 2 * slot + 1), 255, 1, stream));
    if (p0 && n0) launch_range_f32(stream, p0, n0, c->range_dev + 2 * slot);
    if (p1 && n1) launch_range_f32(stream, p1, n1, c->range_dev + 2 * slot);
    PSM_HIP(c, hipMemcpyAsync(c->range_pin + 2 * slot, c->range_dev + 2 * slot, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, stream));
    return 0;
}

bool range_inside(const psm_ctx *c, int slot, int lo_exp, int hi_exp)
{
    const unsigned emax = c->range_pin[2 * slot], emin = c->range_pin[2 * slot + 1];
    if (emax == 0 && emin == 255) return true;                    // all zero
    return (int)emax - 127 <= hi_exp && (int)emin - 127 >= lo_exp;
}

// Host rows -> packed device rows on the context's stream.  hipMemcpy2D takes a row-by-row path when the row length is not a
// multiple of 4 bytes (measured: 6.5 ms for a 450 x 375 x 3 pair, 18.9 ms for 1919 x 1080 x 3, against 0.05 / 0.25 ms): contiguous
// images go as one linear copy, odd-length rows with a pitch are packed on the host first.
int h2d_rows(psm_ctx *c, void *dst, const void *src, size_t row, size_t stride, int rows)
{
    if (stride == row) {
        PSM_HIP(c, hipMemcpyAsync(dst, src, row * rows, hipMemcpyHostToDevice, c->stream));
    } else if (row % 4 == 0 && stride % 4 == 0) {
        PSM_HIP(c, hipMemcpy2DAsync(dst, row, src, stride, row, rows, hipMemcpyHostToDevice, c->stream));
    } else {
        std::vector<uint8_t> packed(row * rows);
        for (int y = 0; y < rows; ++y) memcpy(packed.data() + (size_t)y * row, (const uint8_t *)src + (size_t)y * stride, row);
        PSM_HIP(c, hipMemcpyAsync(dst, packed.data(), row * rows, hipMemcpyHostToDevice, c->stream));
        PSM_HIP(c, hipStreamSynchronize(c->stream));      // (the packed copy lives only here)
    }
    return 0;
}

}  // namespace psm

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

int psm_create_shard_strided(psm_ctx **out, int width, int height, int max_disp, int d_first, int d_step, int dtype, int device)
{
    if (!out) return fail(nullptr, "psm_create: out is NULL");
    *out = nullptr;
    if (d_step < 1 || d_first < 0 || d_first >= max_disp) return fail(nullptr, "psm_create: bad strided slice set (first %d, step %d, of %d)", d_first, d_step, max_disp);
    const int n = (max_disp - d_first + d_step - 1) / d_step;          // d_first, d_first + d_step, ... < max_disp
    // (created as the contiguous shard of the same slice COUNT - the buffers depend on nothing else -, then re-labelled)
    if (psm_create_shard(out, width, height, max_disp, d_first, d_first + n, dtype, device)) return 1;
    psm_ctx *c = *out;
    c->march.dstep = d_step;
    c->d1 = d_first + (n - 1) * d_step + 1;
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
    case PSM_OPT_PROFILE:
        if (value < 0 || value > 2) return fail(c, "psm_set_option: profile mode %d not in {0,1,2}", value);
        c->opt_profile = value; return 0;
    case PSM_OPT_SEG_ROWS:
        if (value < 0) return fail(c, "psm_set_option: seg_rows %d < 0", value);
        c->march.seg_rows = value; return 0;
    case PSM_OPT_WAVES:
        if (value != 1 && value != 2 && value != 4 && value != 8) return fail(c, "psm_set_option: waves %d not in {1,2,4,8}", value);
        c->march.waves = value; return 0;
    case PSM_OPT_FLAGS:
        if (value & ~PSM_FLAGS_ALL) return fail(c, "psm_set_option: unknown flag bits 0x%x", value & ~PSM_FLAGS_ALL);
        // the two arithmetic variants are float-mode forms: an 8-bit context ignores both bits - alone or together - so they are
        // stripped before the mutual-exclusion check (it used to fail on the combination while accepting each one alone)
        if (c->dtype == PSM_U8) value &= ~(PSM_FLAG_FMA_SOLVE | PSM_FLAG_F32_TOL);
        if ((value & PSM_FLAG_F32_TOL) && (value & PSM_FLAG_FMA_SOLVE))
            return fail(c, "psm_set_option: PSM_FLAG_F32_TOL and PSM_FLAG_FMA_SOLVE exclude each other (one arithmetic variant at a time)");
        if ((value ^ c->march.flags) & PSM_FLAG_FMA_SOLVE) {       // the minors and 1/DET in the guidance planes are those of the other reading
            c->have_guid[0] = c->have_guid[1] = false;
            c->guid_y0 = c->guid_y1 = 0;
        }
        c->march.flags = value; return 0;
    case PSM_OPT_GRAPH:
#ifdef PSM_EXPERIMENTS
        c->opt_graph = value != 0; return 0;
#else
        // measured slower than the plain launches in rounds 4 and 5 (one pair 0.29 -> 0.43-0.77 ms, batch of 8 1.62 -> 1.74 ms): the
        // replay exists in experiment builds only (make -C primestereomatch_amd/csrc exp); 0 is accepted so old hosts keep working
        if (value != 0) return fail(c, "psm_set_option: PSM_OPT_GRAPH is not part of the product library (hipGraph replay of a batch measured slower than its plain launches)");
        return 0;
#endif
    case PSM_OPT_GATHER_STAGED: c->opt_gather_staged = value != 0; return 0;
    case PSM_OPT_FRAMES_IN_FLIGHT:
        if (value < 1 || value > 64) return fail(c, "psm_set_option: frames in flight %d not in [1, 64]", value);
        c->march.inflight = value; return 0;
    default: return fail(c, "psm_set_option: unknown option %d", option);
    }
}

int psm_set_stream(psm_ctx *c, void *hip_stream)
{
    if (!c) return 1;
    if (bind(c)) return 1;
    if (c->shared) return fail(c, "psm_set_stream: the context shares its streams with a batch (psm_share_streams)");
    PSM_HIP(c, hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return 0;
}

int psm_synchronize(psm_ctx *c)
{
    if (!c) return 1;
    if (bind(c)) return 1;
    PSM_HIP(c, hipStreamSynchronize(c->stream));
    if (c->copy_stream) PSM_HIP(c, hipStreamSynchronize(c->copy_stream));
    if (c->down_stream && c->down_stream != c->copy_stream) PSM_HIP(c, hipStreamSynchronize(c->down_stream));
    return 0;
}

int psm_release_scratch(psm_ctx *c)
{
    if (!c) return 1;
    if (psm_synchronize(c)) return 1;
    (void)hipFree(c->wm_wts); c->wm_wts = nullptr; c->wm_wts_n = 0;
    (void)hipFree(c->wm_par); c->wm_par = nullptr;
    (void)hipFree(c->wm); c->wm = nullptr;
    (void)hipFree(c->gf_scratch); c->gf_scratch = nullptr; c->gf_scratch_bytes = 0;
    (void)hipFree(c->gather); c->gather = nullptr; c->gather_ranks = 0;
    (void)hipFree(c->fvol); c->fvol = nullptr;
    for (int k = 0; k < 2; ++k)
        if (c->xfer_pin[k]) { (void)hipHostFree(c->xfer_pin[k]); c->xfer_pin[k] = nullptr; c->xfer_pin_bytes[k] = 0; }
    (void)hipGetLastError();
    return 0;
}

int psm_share_streams(psm_ctx *const *ctxs, int n)
{
    if (!ctxs || n < 1 || !ctxs[0]) return fail(nullptr, "psm_share_streams: bad arguments");
    psm_ctx *c0 = ctxs[0];
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i] || ctxs[i]->device != c0->device) return fail(c0, "psm_share_streams: context %d is NULL or on another device", i);
        if (ctxs[i]->shared) return fail(c0, "psm_share_streams: context %d already shares streams", i);
        for (int j = 0; j < i; ++j)      // (a second visit would destroy the set's own upload stream and count the reference twice)
            if (ctxs[j] == ctxs[i]) return fail(c0, "psm_share_streams: context %d appears twice", i);
        if (ctxs[i]->stream != ctxs[i]->own_stream) return fail(c0, "psm_share_streams: context %d runs on a caller's stream (psm_set_stream)", i);
    }
    if (bind(c0)) return 1;
    StreamSet *set = new StreamSet();
    hipError_t e = hipStreamCreateWithFlags(&set->main, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&set->up, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&set->down, hipStreamNonBlocking);
    if (e != hipSuccess) {
        if (set->main) (void)hipStreamDestroy(set->main);
        if (set->up) (void)hipStreamDestroy(set->up);
        delete set;
        return fail(c0, "psm_share_streams: %s", hipGetErrorString(e));
    }
    for (int i = 0; i < n; ++i) {
        psm_ctx *c = ctxs[i];
        (void)hipStreamSynchronize(c->stream);                  // nothing of this context is in flight on the old streams
        if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
        c->shared = set;
        ++set->refs;
        c->stream = set->main;
        c->copy_stream = set->up;
        c->down_stream = set->down;
    }
    return 0;
}

int psm_upload_pair(psm_ctx *c, const void *l, const void *r, int channels, size_t stride_bytes, int depth)
{
    if (!c) return 1;
    size_t row = 0;
    if (check_pair_args(c, "psm_upload_pair", l, r, channels, &stride_bytes, depth, &row)) return 1;
    if (bind(c)) return 1;
    const void *src[2] = {l, r};
    for (int s = 0; s < 2; ++s)
        if (h2d_rows(c, c->raw[s], src[s], row, stride_bytes, c->H)) return 1;
    // float images are used as they are: their range decides whether the select forms' scaled window sums apply
    const size_t nf = (size_t)c->W * c->H * 3;
    if (depth == PSM_IMG_F32 && range_enqueue(c, c->stream, 0, (const float *)c->raw[0], nf, (const float *)c->raw[1], nf)) return 1;
    // the copy reads caller memory: always complete it before returning (CVC_cl::buildCV copies
    // out of the cv::Mats synchronously, src/CVC_cl.cpp:113-160)
    PSM_HIP(c, hipStreamSynchronize(c->stream));
    c->next_depth = -1;                 // a pair staged by psm_upload_pair_async is superseded
    c->range_next_pending = false;
    adopt_new_pair(c, depth);
    c->img_domain_ok = depth != PSM_IMG_F32 || range_inside(c, 0, -PSM_IMG_EXP, PSM_IMG_EXP);
    return 0;
}

// Frame loop (src/main.cpp:64-73): the NEXT pair travels while the current frame is being computed.  The images are copied
// into page-locked staging memory before the call returns (the caller's buffers are free again), the H2D copy runs on the
// context's copy stream into a second image slot; the next psm_cost_construct adopts that pair (its kernels wait for the
// copy on the device - no host synchronisation).  So the call belongs right AFTER psm_cost_construct of the current frame:
//   psm_cost_construct(i); psm_upload_pair_async(pair i+1); psm_cost_filter(i); psm_disp_select(i); ...
int psm_upload_pair_async(psm_ctx *c, const void *l, const void *r, int channels, size_t stride_bytes, int depth)
{
    if (!c) return 1;
    size_t row = 0;
    if (check_pair_args(c, "psm_upload_pair_async", l, r, channels, &stride_bytes, depth, &row)) return 1;
    if (bind(c)) return 1;
    const size_t img = row * c->H;
    if (!c->copy_stream) PSM_HIP(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    for (hipEvent_t *e : {&c->ev_up, &c->ev_free})
        if (!*e) PSM_HIP(c, hipEventCreateWithFlags(e, hipEventDisableTiming));
    const size_t sstride = (c->raw_bytes + 15) & ~(size_t)15;      // (image slots of the staging memory: 16-byte aligned for the copy kernel)
    if (!c->pin_up) PSM_HIP(c, hipHostMalloc((void **)&c->pin_up, 4 * sstride, hipHostMallocDefault));
    for (int s = 0; s < 2; ++s)
        if (!c->raw_next[s]) PSM_HIP(c, hipMalloc(&c->raw_next[s], c->raw_bytes));
    // two staging slots, used alternately: the host only waits for the copy issued TWO uploads ago (the previous one may still
    // be queued behind the current frame's kernels - waiting for it would tie the host to the device's pace)
    const int slot = c->stage_slot ^= 1;
    if (!c->ev_stage[slot]) PSM_HIP(c, hipEventCreateWithFlags(&c->ev_stage[slot], hipEventDisableTiming));
    else PSM_HIP(c, hipEventSynchronize(c->ev_stage[slot]));
    uint8_t *stage = c->pin_up + (size_t)slot * 2 * sstride;
    const void *src[2] = {l, r};
    for (int s = 0; s < 2; ++s) {
        uint8_t *dst = stage + s * sstride;
        if (stride_bytes == row) memcpy(dst, src[s], img);
        else for (int y = 0; y < c->H; ++y) memcpy(dst + (size_t)y * row, (const uint8_t *)src[s] + (size_t)y * stride_bytes, row);
    }
    // raw_next was the current pair two frames ago: its k_prep (recorded as ev_free by psm_cost_construct) must be over
    PSM_HIP(c, hipStreamWaitEvent(c->copy_stream, c->ev_free, 0));
    // Small images (Middlebury size: 0.5 MB) travel through a kernel that reads the page-locked slot - submitting a copy-engine
    // transfer behind a stream wait costs 0.2 - 0.4 ms of HOST time on this stack, more than such a pair takes to compute; large
    // ones keep the copy engines, which overlap a 12 MB pair with the filter without taking CU slots (1080p: +0.10 vs +0.27 ms).
    for (int s = 0; s < 2; ++s) {
        if (img <= PSM_COPY_KERNEL_MAX) launch_copy_bytes(c->copy_stream, c->raw_next[s], stage + s * sstride, img);
        else PSM_HIP(c, hipMemcpyAsync(c->raw_next[s], stage + s * sstride, img, hipMemcpyHostToDevice, c->copy_stream));
    }
    if (check_launch(c, "upload (copy kernel)")) return 1;
    c->range_next_pending = false;
    if (depth == PSM_IMG_F32) {         // (measured behind the copy on the copy stream; read when the pair is adopted)
        const size_t nf = (size_t)c->W * c->H * 3;
        if (range_enqueue(c, c->copy_stream, 1, (const float *)c->raw_next[0], nf, (const float *)c->raw_next[1], nf)) return 1;
        c->range_next_pending = true;
    }
    PSM_HIP(c, hipEventRecord(c->ev_up, c->copy_stream));
    PSM_HIP(c, hipEventRecord(c->ev_stage[slot], c->copy_stream));
    c->up_recorded = true;
    c->next_depth = depth;
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
    PSM_NOT_STRIDED(c, "psm_upload_volume");
    if (bind(c)) return 1;
    // a partial upload must not leave virtual slices behind: whatever of this side exists only as a recipe (lazy costs - also
    // after a striped psm_cost_construct, which leaves have_g1 false -, packed minima, FGF models) becomes real data first
    if (c->have_cost && (c->raw_rows[side] != psm_ctx::RAW_ALL || c->gf_virtual[side] || c->fgf_virtual[side]) && materialize(c, side)) return 1;
    c->gf_virtual[side] = false;
    c->maps_early = nullptr;
    if (ensure_vol(c, side)) return 1;
    const size_t S = (size_t)c->W * c->H * velem(c);
    PSM_HIP(c, hipMemcpyAsync((char *)c->vol[side] + (size_t)(d0 - c->d0) * S, host, (size_t)(d1 - d0) * S, hipMemcpyHostToDevice, c->stream));
    if (c->dtype == PSM_F32 && range_enqueue(c, c->stream, 2 + side, (const float *)((char *)c->vol[side] + (size_t)(d0 - c->d0) * S),
                                             (size_t)(d1 - d0) * c->W * c->H, nullptr, 0)) return 1;
    PSM_HIP(c, hipStreamSynchronize(c->stream));
    // costs of arbitrary scale: outside 2^-60 .. 2^60 the filter of this volume runs the storing form (sticky until the next
    // psm_cost_construct replaces the volume)
    if (c->dtype == PSM_F32 && !range_inside(c, 2 + side, -PSM_VOL_EXP, PSM_VOL_EXP)) c->vol_domain_ok[side] = false;
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
    if (c->have_images && !c->have_g1 && run_prep(c)) return 1;     // (image preparation is lazy: psm_cost_construct may have left it to the filter)
    PSM_HIP(c, hipStreamSynchronize(c->stream));
    PSM_HIP(c, hipMemcpy(b1.data(), c->g[side].g1, HW * sizeof(float4), hipMemcpyDeviceToHost));
    PSM_HIP(c, hipMemcpy(b2.data(), c->g[side].g2, HW * sizeof(float4), hipMemcpyDeviceToHost));
    PSM_HIP(c, hipMemcpy(b3.data(), c->g[side].g3, HW * sizeof(float4), hipMemcpyDeviceToHost));
    PSM_HIP(c, hipMemcpy(b4.data(), c->g[side].g4, HW * sizeof(float2), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < HW; ++i) {
        host[0 * HW + i] = b1[i].x; host[1 * HW + i] = b1[i].y; host[2 * HW + i] = b1[i].z; host[3 * HW + i] = b1[i].w;
        if (c->dtype == PSM_U8) {   // 8-bit contexts carry the pixel's {c0,c1,c2,grad} bytes in that slot: hand out the 8-bit gradient
            unsigned w; memcpy(&w, &b1[i].w, 4);
            host[3 * HW + i] = (float)(w >> 24);
        }
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

// PSM_OPT_PROFILE 2: every launch of the fused filter kernel (k_cvf_pc) stamps the device's constant-rate clock when its
// first workgroup starts and when its last one ends (two 64-bit atomics per workgroup; nothing else changes, no events sit
// between the kernels) - the durations of the launches of a TIMED region, not of a separate profiling pass.
int psm_filter_launch_times(psm_ctx *c, double *ms, int *form, int max_launches, int *n_launches)
{
    if (!c || !n_launches) return 1;
    *n_launches = 0;
    if (bind(c)) return 1;
    PSM_HIP(c, hipStreamSynchronize(c->stream));
    const int n = c->pc_ts_n < PC_TS_SLOTS ? c->pc_ts_n : PC_TS_SLOTS;
    if (n > 0 && c->pc_ts) {
        std::vector<unsigned long long> h(3 * (size_t)PC_TS_SLOTS);
        PSM_HIP(c, hipMemcpy(h.data(), c->pc_ts, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        int khz = 0;
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device) != hipSuccess || khz <= 0) khz = 100000;
        for (int i = 0; i < n && i < max_launches; ++i) {
            const unsigned long long a = h[i], b = h[PC_TS_SLOTS + i];
            if (ms) ms[i] = b > a ? (double)(b - a) / (double)khz : 0.0;
            if (form) form[i] = (int)h[2 * PC_TS_SLOTS + i];
        }
        *n_launches = n < max_launches ? n : max_launches;
    }
    // start over: starts <- all ones, ends / forms <- 0
    if (c->pc_ts) {
        PSM_HIP(c, hipMemsetAsync(c->pc_ts, 0xff, PC_TS_SLOTS * sizeof(unsigned long long), c->stream));
        PSM_HIP(c, hipMemsetAsync(c->pc_ts + PC_TS_SLOTS, 0, 2 * PC_TS_SLOTS * sizeof(unsigned long long), c->stream));
    }
    c->pc_ts_n = 0;
    return 0;
}

}  // extern "C"
