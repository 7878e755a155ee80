// psm_api_pp.cpp - the post-processing rows behind the C ABI: lrCheck, fillInv and the plain weighted
// median of PP::processDM (src/PP.cpp:17-247,405-410) on the device maps of the last DispSelect.  The reference runs these
// on the CPU; here they sit behind the same boundary so that the maps never leave the device between the stages.
#include "psm_ctx.h"

using namespace psm;

extern "C" {

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
    if (bind(c) || maps_writable(c)) return 1;
    const double t0 = now_us();
    const size_t HW = (size_t)c->W * c->H;
    {
        Prof p(c, PSM_K_LRC);
        launch_fill_inv(c->stream, c->maps, c->valid, HW, c->W, c->H);
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
    if (herr) {
        // the kernel gave up on unfinished rows: the in-place maps are partially filtered - not results
        c->have_maps = false;
        c->have_valid = false;
        return fail(c, "psm_wgt_median: row pipeline stalled (watchdog); the device maps are no longer valid - select again");
    }
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
    if (bind(c) || maps_writable(c)) return 1;
    const double t0 = now_us();
    if (!c->have_g1 && run_prep(c)) return 1;
    const size_t HW = (size_t)c->W * c->H;
    // PSM_FLAG_WMF_DATAFLOW: dataflow form only; PSM_FLAG_WMF_TWO_SWEEPS: at most 2 sweeps (test hook for the fall-back)
    const bool dataflow_only = (c->march.flags & PSM_FLAG_WMF_DATAFLOW) != 0;
    const int CAP = (c->march.flags & PSM_FLAG_WMF_TWO_SWEEPS) ? 2 : 96;
    bool done[2] = {false, false};
    if (!dataflow_only) {
        // parallel form: sweeps to the fixed point of the in-place recursion (psm_pp.hip), both maps side by side
        const size_t nb = (HW + 255) / 256 * 256, ncnt = 2 * (size_t)(96 + 2);
        const size_t per_side = 4 * nb + 6 * nb * sizeof(int);       // (the counters of both maps follow the two sides: one fill, one snapshot)
        if (!c->wm_par) PSM_HIP(c, hipMalloc((void **)&c->wm_par, 2 * per_side + 2 * ncnt * sizeof(int)));
        int *const cnt0 = reinterpret_cast<int *>(c->wm_par + 2 * per_side);
        WmPair pr;
        int *cnt[2], *inv[2], *slot_of[2];
        for (int s = 0; s < 2; ++s) {
            uint8_t *b = c->wm_par + s * per_side;
            int *ip = reinterpret_cast<int *>(b + 4 * nb);
            WmSide &a = pr.s[s];
            a.cur = c->maps + s * HW; a.valid = c->valid + s * HW; a.g1 = c->g[s].g1;
            a.orig = b; a.newv = b + nb; a.chgb = b + 2 * nb; a.rowany = b + 3 * nb;
            a.stamp = ip; a.list[0] = ip + nb; a.list[1] = ip + 2 * nb; a.chg = ip + 3 * nb; a.slot_of = ip + 4 * nb; a.inv = ip + 5 * nb;
            a.cnt = cnt0 + s * ncnt; a.wts = nullptr;
            cnt[s] = a.cnt; inv[s] = a.inv; slot_of[s] = a.slot_of;
        }
        const size_t snap = 2 * ncnt;                      // ints per counter snapshot (both maps)
        if (!c->wm_pin) PSM_HIP(c, hipHostMalloc((void **)&c->wm_pin, 2 * snap * sizeof(int), hipHostMallocDefault));
        for (hipEvent_t *e : {&c->ev_wm[0], &c->ev_wm[1]})
            if (!*e) PSM_HIP(c, hipEventCreateWithFlags(e, hipEventDisableTiming));
        PSM_HIP(c, hipMemsetAsync(cnt0, 0, 2 * ncnt * sizeof(int), c->stream));
        launch_wm_seed(c->stream, pr, c->W, c->H);     // all invalid pixels = the first sweep's lists; input copies; per-pixel state zeroed
        // Weight cache: the 19 x 19 weights of an invalid pixel depend on the image only, and the sweeps evaluate it ~5 times.
        // For sides with at least WM_LANE_MIN invalid pixels one pass forms them all (1.5 KB per invalid pixel; beyond
        // WM_CACHE_MAX bytes for the pair the evaluations form their weights themselves, as they do for short lists).
        float *wts[2] = {nullptr, nullptr};
        bool cached = false;
        {
            // at most 12 GB, and never more than half of what the device has free right now: the volumes, the spare volume and
            // the FGF scratch of this or another context on the device are allocated on first use and must still fit
            size_t WM_CACHE_MAX = (size_t)12 << 30, mem_free = 0, mem_total = 0;
            if (hipMemGetInfo(&mem_free, &mem_total) == hipSuccess) {
                const size_t avail = mem_free + c->wm_wts_n * sizeof(float);     // (what we hold already counts as available to us)
                if (avail / 2 < WM_CACHE_MAX) WM_CACHE_MAX = avail / 2;
            } else (void)hipGetLastError();
            // (the counts of invalid pixels through the same page-locked snapshot the sweeps use: one small kernel, no copy engine)
            launch_copy_bytes(c->stream, c->wm_pin, cnt0, snap * sizeof(int));
            PSM_HIP(c, hipStreamSynchronize(c->stream));
            const int n0[2] = {c->wm_pin[0], c->wm_pin[ncnt]};
            size_t need[2], tot = 0;
            for (int s = 0; s < 2; ++s) {      // (whole blocks of 64 pixels: the cache is laid out in such blocks)
                need[s] = (size_t)((n0[s] + 63) / 64 * 64) * WM_WPIX;
                tot += need[s];
            }
            if (n0[0] < WM_LANE_MIN && n0[1] < WM_LANE_MIN) tot = 0;      // (short lists form their weights themselves; a launch takes both maps one way)
            if (tot && tot * sizeof(float) <= WM_CACHE_MAX && !(c->march.flags & PSM_FLAG_WMF_NO_CACHE)) {
                if (c->wm_wts_n < tot) {
                    (void)hipFree(c->wm_wts);
                    c->wm_wts = nullptr;
                    c->wm_wts_n = 0;
                    if (hipMalloc((void **)&c->wm_wts, tot * sizeof(float)) == hipSuccess) c->wm_wts_n = tot;
                    else (void)hipGetLastError();           // no memory for it: recompute, as without the cache
                }
                if (c->wm_wts_n >= tot) {
                    wts[0] = c->wm_wts;
                    wts[1] = c->wm_wts + need[0];
                    cached = true;
                    Prof p(c, PSM_K_WMF);
                    for (int s = 0; s < 2; ++s) {
                        pr.s[s].wts = wts[s];
                        if (n0[s]) launch_wm_weights(c->stream, c->g[s].g1, c->W, c->H, s, inv[s], cnt[s], n0[s], wts[s], slot_of[s]);
                    }
                }
            }
        }
        // Sweeps are launched in groups of `chk`; the host looks at a group's counters (did a sweep change nothing? how long is the
        // next list?) while the NEXT group is already queued - the device never idles behind a host round trip (round 3: ~65 us
        // of idle per check, 0.7 ms of the 4.0 on the 1080p bench pair).  A group launched for a map that had already reached
        // its fixed point is a handful of launches over empty lists.
        int sw = 0;                        // sweeps launched
        bool tail[2] = {false, false};     // the list going into the next sweep is short: two launches per sweep, more sweeps per check
        int upto[2] = {0, 0};              // sweeps covered by the snapshot in slot g & 1
        auto launch_group = [&](int slot) -> int {
            const bool short_lists = (tail[0] || done[0]) && (tail[1] || done[1]);
            // two sweeps per group while the lists are long (the host learns two groups late that they have become short, and a
            // sweep launched the long way costs three launches more than it needs then), eight once they are short
            const int chk = short_lists ? 8 : 2;
            const int end = sw + chk < CAP ? sw + chk : CAP;
            {
                Prof p(c, PSM_K_WMF);
                // (a map that has reached its fixed point has empty lists from then on: its half of a launch returns at once)
                for (; sw < end; ++sw) launch_wm_sweep(c->stream, pr, c->W, c->H, c->D, sw, cached, short_lists);
            }
            if (check_launch(c, "wgt_median (sweeps)")) return 1;
            // (a kernel writing the page-locked snapshot: a copy-engine transfer in the stream stalls it for ~60 us)
            launch_copy_bytes(c->stream, c->wm_pin + slot * snap, cnt0, snap * sizeof(int));
            PSM_HIP(c, hipEventRecord(c->ev_wm[slot], c->stream));
            upto[slot] = sw;
            return 0;
        };
        if (launch_group(0)) return 1;
        for (int g = 0;; ++g) {
            const bool more = sw < CAP;
            if (more && launch_group((g + 1) & 1)) return 1;          // (queued before this group's counters are looked at)
            PSM_HIP(c, hipEventSynchronize(c->ev_wm[g & 1]));
            const int *h = c->wm_pin + (g & 1) * snap;
            const int seen = upto[g & 1];
            for (int s = 0; s < 2; ++s) {
                if (done[s]) continue;
                long long ev = 0;
                for (int k = 0; k < seen; ++k) {
                    ev += h[s * ncnt + 2 * k];
                    if (h[s * ncnt + 2 * k + 1] == 0) {      // sweep k changed nothing: fixed point
                        done[s] = true;
                        c->wm_sweeps[s] = k + 1;
                        c->wm_evals[s] = ev;
                        break;
                    }
                }
                tail[s] = seen < CAP && h[s * ncnt + 2 * seen] < WM_LANE_MIN;      // (the count the last sweep of the group left for the next one: wave form)
            }
            if ((done[0] && done[1]) || !more) break;
        }
#ifdef PSM_EXPERIMENTS
        if (getenv("PSM_WM_TRACE")) {      // per sweep: pixels evaluated / changed (both maps)
            std::vector<int> hc(ncnt);
            for (int s = 0; s < 2; ++s) {
                PSM_HIP(c, hipMemcpy(hc.data(), cnt[s], ncnt * sizeof(int), hipMemcpyDeviceToHost));
                fprintf(stderr, "[wm] side %d:", s);
                for (int k = 0; k < sw && k < 96; ++k) fprintf(stderr, " %d/%d", hc[2 * k], hc[2 * k + 1]);
                fprintf(stderr, "\n");
            }
        }
#endif
        // no fixed point within CAP sweeps (long chains of pixels that keep flipping each other): start over from the input
        // with the dataflow form, which is exact for any input
        for (int s = 0; s < 2; ++s)
            if (!done[s]) PSM_HIP(c, hipMemcpyAsync(c->maps + s * HW, pr.s[s].orig, HW, hipMemcpyDeviceToDevice, c->stream));
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
    if (bind(c) || maps_writable(c)) return 1;
    c->maps_early = nullptr;
    const size_t HW = (size_t)c->W * c->H;
    const uint8_t *src[4] = {lmap, rmap, lvalid, rvalid};
    uint8_t *dst[4] = {c->maps, c->maps + HW, c->valid, c->valid + HW};
    for (int i = 0; i < 4; ++i)
        if (src[i] && h2d_rows(c, dst[i], src[i], (size_t)c->W, stride, c->H)) return 1;
    PSM_HIP(c, hipStreamSynchronize(c->stream));
    if (lmap && rmap) { c->have_maps = true; c->have_valid = false; c->have_rows = false; c->rows_y0 = 0; c->rows_y1 = c->H; }   // whole maps
    if (lvalid && rvalid && c->have_maps) c->have_valid = true;
    return 0;
}

}  // extern "C"
