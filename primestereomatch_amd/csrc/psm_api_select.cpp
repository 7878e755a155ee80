// psm_api_select.cpp - DispSelect behind the C ABI: the two u8 maps, the packed per-pixel minima a sharded job exchanges,
// row stripes and disparity shards, and the D2H leg of the maps.  Replaces DispSel_cl::CVSelect
// (src/DispSel_cl.cpp:69-140) as called by DispEst::DispSelect_GPU (src/DispEst.cpp:323-328); arithmetic of
// DispSel::CVSelect (src/DispSel.cpp:83-109).
#include "psm_ctx.h"

#include <cstring>

using namespace psm;

namespace psm {

int copy_maps_out(psm_ctx *c, const uint8_t *dev, uint8_t *lmap, uint8_t *rmap, size_t stride)
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

}  // namespace psm

namespace {

// WTA of one side into keys_s (may be NULL) and / or map_s (may be NULL).  A side whose Fast-Guided-Filter result is
// still virtual is selected straight from the smoothed models (upsample + linear model + argmin in one pass).
int wta_side(psm_ctx *c, int s, long long *keys_s, uint8_t *map_s)
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

int wta_launch(psm_ctx *c, long long *keys, uint8_t *maps)
{
    const size_t HW = (size_t)c->W * c->H;
    // the map buffer may still be the source of an asynchronous download of the previous frame
    if (maps && maps_writable(c)) return 1;
    // a side that is not the select filter's packed minima was selected from a whole volume: its map is whole
    if (!(c->gf_virtual[0] && c->gf_virtual[1])) { c->have_rows = false; c->rows_y0 = 0; c->rows_y1 = c->H; }
    if (c->gf_virtual[0] && c->gf_virtual[1] && !keys && maps) {   // both sides already reduced to keys: one launch for both maps
        const bool done = c->maps_early == maps;                    // ... unless the filter's reduction wrote them already
        c->maps_early = nullptr;                                    // (once: post-processing rewrites the maps in place)
        if (done) return 0;
        Prof p(c, PSM_K_WTA);
        launch_merge(c->stream, c->keys_cur, 2 * HW, 1, (int)(2 * HW), maps);
        return check_launch(c, "wta");
    }
    for (int s = 0; s < 2; ++s)
        if (wta_side(c, s, keys ? keys + s * HW : nullptr, maps ? maps + s * HW : nullptr)) return 1;
    return check_launch(c, "wta");
}

// the volume a WTA is about to read: real data, or a virtual result (consumed without materialising it)
int wta_ready(psm_ctx *c, int side)
{
    return (c->fgf_virtual[side] || c->gf_virtual[side]) ? 0 : materialize(c, side);
}

}  // namespace

extern "C" {

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

int psm_set_rows(psm_ctx *c, int y_begin, int y_end)
{
    if (!c) return 1;
    // (takes effect with the next psm_cost_filter; minima / maps already computed keep describing the stripe they were made
    // for - rows_y0 / rows_y1, recorded when they were filtered)
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
    c->maps_early = nullptr;
    c->maps = m;
    if (whole) {    // the caller filled the buffer with both complete maps of the current frame (e.g. gathered row stripes)
        c->have_maps = true;
        c->have_rows = false;
        c->rows_y0 = 0;
        c->rows_y1 = c->H;
        c->have_valid = false;
    }
    return 0;
}

// One leg of the single-process exchange (psm_gather_rows_ctx / psm_disp_merge_ctx): n bytes of context s (its stream already
// synchronised) into root's memory, ordered on root's stream.  Same device: a device copy.  Other device: a peer copy when
// hipDeviceCanAccessPeer(root, s) says the two can reach each other (xGMI / PCIe P2P), else - or with PSM_OPT_GATHER_STAGED, the
// test hook for exactly this path on a one-GPU box - through a page-locked bounce buffer of root: device -> host on s's device,
// host -> device on root's stream (two buffers used alternately: the next leg's device -> host copy overlaps this leg's host -> device
// copy, a slot is refilled only after the copy out of it has executed - the slow but always available way).
static int gather_leg(psm_ctx *root, void *dst, psm_ctx *s, const void *src, size_t n)
{
    bool staged = root->opt_gather_staged != 0;
    if (!staged && s->device != root->device) {
        signed char &ok = root->peer_ok[s->device & 63];      // asked once per device pair, not on every leg of every frame
        if (ok < 0) {
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, root->device, s->device) != hipSuccess) { can = 0; (void)hipGetLastError(); }
            ok = can ? 1 : 0;
        }
        staged = !ok;
    }
    if (!staged) {
        if (s->device == root->device) PSM_HIP(root, hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, root->stream));
        else PSM_HIP(root, hipMemcpyPeerAsync(dst, root->device, src, s->device, n, root->stream));
        return 0;
    }
    const int k = root->xfer_slot;
    root->xfer_slot ^= 1;
    if (!root->ev_xfer[k]) PSM_HIP(root, hipEventCreateWithFlags(&root->ev_xfer[k], hipEventDisableTiming));
    else PSM_HIP(root, hipEventSynchronize(root->ev_xfer[k]));    // the copy out of this slot (two legs ago) has executed
    if (root->xfer_pin_bytes[k] < n) {
        if (root->xfer_pin[k]) (void)hipHostFree(root->xfer_pin[k]);
        root->xfer_pin[k] = nullptr;
        root->xfer_pin_bytes[k] = 0;
        PSM_HIP(root, hipHostMalloc((void **)&root->xfer_pin[k], n, hipHostMallocPortable));    // (portable: both devices copy to / from it)
        root->xfer_pin_bytes[k] = n;
    }
    (void)hipSetDevice(s->device);
    const hipError_t e = hipMemcpy(root->xfer_pin[k], src, n, hipMemcpyDeviceToHost);    // (blocking: the slot is complete when it returns)
    (void)hipSetDevice(root->device);
    if (e != hipSuccess) return fail(root, "exchange leg (device %d -> host): %s", s->device, hipGetErrorString(e));
    PSM_HIP(root, hipMemcpyAsync(dst, root->xfer_pin[k], n, hipMemcpyHostToDevice, root->stream));
    PSM_HIP(root, hipEventRecord(root->ev_xfer[k], root->stream));
    ++root->gather_staged_legs;
    return 0;
}

int psm_gather_rows_ctx(psm_ctx *root, psm_ctx *const *stripes, int nstripes, uint8_t *lmap, uint8_t *rmap, size_t stride)
{
    if (!root) return 1;
    if (!stripes || nstripes < 1) return fail(root, "psm_gather_rows_ctx: bad arguments");
    std::vector<char> covered((size_t)root->H, 0);
    // which rows a context's maps hold: the stripe they were FILTERED with (not what psm_set_rows says now)
    auto y0_of = [](const psm_ctx *s) { return s->have_rows ? s->rows_y0 : 0; };
    auto y1_of = [](const psm_ctx *s) { return s->have_rows ? s->rows_y1 : s->H; };
    for (int i = 0; i < nstripes; ++i) {
        const psm_ctx *s = stripes[i];
        if (!s || s->W != root->W || s->H != root->H || s->D != root->D || s->dtype != root->dtype)
            return fail(root, "psm_gather_rows_ctx: stripe %d does not belong to this job", i);
        if (!s->have_maps) return fail(root, "psm_gather_rows_ctx: stripe %d has no maps for this frame (call psm_disp_select first)", i);
        for (int y = y0_of(s); y < y1_of(s); ++y) {
            if (covered[y]) return fail(root, "psm_gather_rows_ctx: row %d is held by more than one stripe", y);
            covered[y] = 1;
        }
    }
    for (int y = 0; y < root->H; ++y)
        if (!covered[y]) return fail(root, "psm_gather_rows_ctx: no stripe holds row %d", y);
    if (bind(root) || maps_writable(root)) return 1;
    const size_t HW = (size_t)root->W * root->H;
    for (int i = 0; i < nstripes; ++i) {
        psm_ctx *s = stripes[i];
        if (s == root) continue;
        (void)hipSetDevice(s->device);
        PSM_HIP(root, hipStreamSynchronize(s->stream));     // the stripe's maps must be complete before they are read
        (void)hipSetDevice(root->device);
        const size_t o = (size_t)y0_of(s) * root->W, n = (size_t)(y1_of(s) - y0_of(s)) * root->W;
        for (int side = 0; side < 2; ++side)
            if (gather_leg(root, root->maps + side * HW + o, s, s->maps + side * HW + o, n)) return 1;
    }
    root->have_maps = true;
    root->have_rows = false;      // the root's maps are whole now
    root->rows_y0 = 0;
    root->rows_y1 = root->H;
    root->have_valid = false;
    if (copy_maps_out(root, root->maps, lmap, rmap, stride)) return 1;
    if (!root->opt_async) PSM_HIP(root, hipStreamSynchronize(root->stream));
    return 0;
}

int psm_gather_staged_legs(const psm_ctx *root) { return root ? root->gather_staged_legs : -1; }

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
    if (maps_writable(c)) return 1;
    {
        Prof p(c, PSM_K_MERGE);
        launch_merge(c->stream, (const long long *)dev_keys_all, n, nranks, (int)n, c->maps);
    }
    if (check_launch(c, "merge")) return 1;
    // (the merged keys are those of this job's shards, filtered under the stripe this context recorded - rows_y0 / rows_y1
    // stay as psm_cost_filter left them: stripes of disparity shards merge to a stripe)
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
        if (s->have_rows != shards[0]->have_rows || (s->have_rows && (s->rows_y0 != shards[0]->rows_y0 || s->rows_y1 != shards[0]->rows_y1)))
            return fail(root, "psm_disp_merge_ctx: shard %d was filtered under another row stripe than shard 0", i);
        for (int d = s->d0; d < s->d1; d += s->march.dstep) {       // (a strided shard holds d0, d0 + dstep, ...)
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
        if (gather_leg(root, (char *)root->gather + bytes * i, s, s->keys_cur, bytes)) return 1;
    }
    // the merged maps cover what the shards' minima cover
    root->have_rows = shards[0]->have_rows;
    root->rows_y0 = shards[0]->have_rows ? shards[0]->rows_y0 : 0;
    root->rows_y1 = shards[0]->have_rows ? shards[0]->rows_y1 : root->H;
    return psm_disp_merge(root, root->gather, nshards, lmap, rmap, stride);
}

int psm_download_maps(psm_ctx *c, uint8_t *lmap, uint8_t *rmap, size_t stride)
{
    if (!c) return 1;
    if (!c->have_maps) return fail(c, "psm_download_maps: no disparity maps computed");
    if (bind(c)) return 1;
    return copy_maps_out(c, c->maps, lmap, rmap, stride);
}

// Frame loop: the D2H leg of frame i next to the kernels of frame i+1.  psm_download_maps_async starts the copy of the
// current maps into page-locked memory on the copy stream (after the kernels that produce them, before any later kernel
// overwrites them); psm_download_maps_wait hands them to the caller.  One download may be in flight.
int psm_download_maps_async(psm_ctx *c)
{
    if (!c) return 1;
    if (!c->have_maps) return fail(c, "psm_download_maps_async: no disparity maps computed");
    if (bind(c)) return 1;
    const size_t HW = (size_t)c->W * c->H;
    if (!c->copy_stream) PSM_HIP(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    if (!c->down_stream) c->down_stream = c->copy_stream;
    for (hipEvent_t *e : {&c->ev_maps, &c->ev_down})
        if (!*e) PSM_HIP(c, hipEventCreateWithFlags(e, hipEventDisableTiming));
    if (!c->pinned2) PSM_HIP(c, hipHostMalloc((void **)&c->pinned2, 2 * HW, hipHostMallocDefault));
    PSM_HIP(c, hipEventRecord(c->ev_maps, c->stream));
    PSM_HIP(c, hipStreamWaitEvent(c->down_stream, c->ev_maps, 0));
    if (((uintptr_t)c->maps & 15) == 0 && 2 * HW <= PSM_COPY_KERNEL_MAX) {     // small maps: a kernel writing page-locked host memory (k_copy16; see psm_upload_pair_async)
        launch_copy_bytes(c->down_stream, c->pinned2, c->maps, 2 * HW);
        if (check_launch(c, "download (copy kernel)")) return 1;
    } else PSM_HIP(c, hipMemcpyAsync(c->pinned2, c->maps, 2 * HW, hipMemcpyDeviceToHost, c->down_stream));
    PSM_HIP(c, hipEventRecord(c->ev_down, c->down_stream));
    return 0;
}

int psm_download_maps_wait(psm_ctx *c, uint8_t *lmap, uint8_t *rmap, size_t stride)
{
    if (!c) return 1;
    if (!c->ev_down || !c->pinned2) return fail(c, "psm_download_maps_wait: no asynchronous download started");
    if (stride == 0) stride = c->W;
    if (stride < (size_t)c->W) return fail(c, "map stride %zu < width %d", stride, c->W);
    if (bind(c)) return 1;
    PSM_HIP(c, hipEventSynchronize(c->ev_down));
    const size_t HW = (size_t)c->W * c->H;
    uint8_t *dst[2] = {lmap, rmap};
    for (int s = 0; s < 2; ++s) {
        if (!dst[s]) continue;
        const uint8_t *src = c->pinned2 + s * HW;
        if (stride == (size_t)c->W) memcpy(dst[s], src, HW);
        else for (int y = 0; y < c->H; ++y) memcpy(dst[s] + (size_t)y * stride, src + (size_t)y * c->W, (size_t)c->W);
    }
    return 0;
}

}  // extern "C"
