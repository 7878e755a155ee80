// psm_api_batch.cpp - several stereo pairs of one geometry per launch.  The reference's use on Middlebury-size data is a
// loop over pairs / datasets (src/main.cpp:64-73, src/StereoMatch.cpp:556-607): one 450 x 375 x 64 pair is 1 280 workgroups -
// 1.7 rounds of the chip's resident slots - behind four launches at their latency floor.  psm_compute_batch runs
// DispEst::CostConst_GPU + CostFilter_GPU + DispSelect_GPU (src/DispEst.cpp:272-276,299-308,323-328) of n contexts in shared
// launches: one k_prep, one k_guide_march, one k_cvf_pc grid with blockIdx.z = pair (two in the two-phase form), one
// reduction.  Every context stays a complete context: its maps, minima, validity masks, post-processing and volume readers
// behave as after the three single-pair calls, and the results are the same bits.
#include "psm_ctx.h"

#include <cstring>

using namespace psm;

extern "C" int psm_compute_batch(psm_ctx *const *ctxs, int n)
{
    if (!ctxs || n < 1 || !ctxs[0]) return fail(nullptr, "psm_compute_batch: bad arguments");
    psm_ctx *c0 = ctxs[0];
    if (n > 4096) return fail(c0, "psm_compute_batch: %d pairs (at most 4096 per call)", n);
    for (int i = 0; i < n; ++i) {
        psm_ctx *c = ctxs[i];
        if (!c) return fail(c0, "psm_compute_batch: context %d is NULL", i);
        for (int j = 0; j < i; ++j)
            if (ctxs[j] == c) return fail(c0, "psm_compute_batch: context %d appears twice", i);
        if (c->W != c0->W || c->H != c0->H || c->D != c0->D || c->d0 != c0->d0 || c->d1 != c0->d1 || c->dtype != c0->dtype || c->device != c0->device)
            return fail(c0, "psm_compute_batch: context %d has another geometry / type / device than context 0", i);
        if (c->opt_variant != 0 || (c->march.flags & (PSM_FLAG_STORE_FILTERED | PSM_FLAG_MATERIALISE_COSTS)))
            return fail(c0, "psm_compute_batch: context %d asks for a storing form (the batch runs the default select path)", i);
        if (c->march.flags != c0->march.flags || c->march.seg_rows != c0->march.seg_rows || c->march.dstep != c0->march.dstep)
            return fail(c0, "psm_compute_batch: context %d has other options than context 0", i);
        // the batched launches exist in the bit-exact form only: a silently ignored flag would break "same maps as the three
        // single-pair calls" (which DO run the tolerance form with this flag)
        if (c->march.flags & (PSM_FLAG_F32_TOL | PSM_FLAG_FMA_SOLVE))
            return fail(c0, "psm_compute_batch: context %d asks for PSM_FLAG_F32_TOL / PSM_FLAG_FMA_SOLVE (single-pair entry points only)", i);
        if (c->march.yend > c->march.ybeg) return fail(c0, "psm_compute_batch: context %d is restricted to a row stripe", i);
        if (!c->have_images && c->next_depth < 0) return fail(c0, "psm_compute_batch: context %d has no image pair", i);
        if (c->next_depth >= 0 ? c->next_depth != (c0->next_depth >= 0 ? c0->next_depth : c0->raw_depth)
                               : c->raw_depth != (c0->next_depth >= 0 ? c0->next_depth : c0->raw_depth))
            return fail(c0, "psm_compute_batch: context %d holds images of another depth than context 0", i);
    }
    if (bind(c0)) return 1;
    const double t0 = now_us();
    for (int i = 0; i < n; ++i) {        // (float images staged asynchronously bring their range with them)
        psm_ctx *c = ctxs[i];
        if (c->next_depth >= 0 && c->range_next_pending) {
            PSM_HIP(c0, hipEventSynchronize(c->ev_up));
            if (!range_inside(c, 1, -PSM_IMG_EXP, PSM_IMG_EXP)) return fail(c0, "psm_compute_batch: the float images of context %d are outside the select forms' domain (2^-10 .. 2^10)", i);
        } else if (c->next_depth < 0 && !c->img_domain_ok)      // (only the IMAGES matter: the batch rebuilds the costs from them,
            return fail(c0, "psm_compute_batch: the float images of context %d are outside the select forms' domain (2^-10 .. 2^10); use the single-pair entry points", i);
    }                                                           //  as psm_cost_construct does before it forgets an uploaded volume's range)
    const int W = c0->W, H = c0->H, Dloc = c0->Dloc;
    hipStream_t s = c0->stream;
    if (!c0->ev_batch) PSM_HIP(c0, hipEventCreateWithFlags(&c0->ev_batch, hipEventDisableTiming));

    // ---- plan and scratch (before any launch: growing a scratch buffer synchronises its context's stream) ----
    const bool two_phase = !(c0->march.flags & PSM_FLAG_TWO_PHASE_OFF) && Dloc >= 2 && (Dloc >= 112 || (c0->march.flags & PSM_FLAG_TWO_PHASE_ON));
    const int S = pc_seed_stride(W, H, c0->dtype == PSM_U8);
    const int n1 = two_phase ? (Dloc + S - 1) / S : Dloc, n2 = Dloc - n1;
    const PcPlan pl = pc_plan(W, H, n1, c0->march.seg_rows, PC_PLANES | PC_BOTH, n);
    for (int i = 0; i < n; ++i)
        if (ensure_gf_scratch(ctxs[i], 2 * pl.scratch_bytes())) return fail(c0, "psm_compute_batch: %s", ctxs[i]->err.c_str());

    // ---- every context's earlier work (uploads, downloads of its maps) is ordered before the shared launches ----
    for (int i = 0; i < n; ++i) {
        psm_ctx *c = ctxs[i];
        hipStream_t own = c->stream;
        c->stream = s;                       // (adopt_staged_pair / maps_writable make `s` wait for the copy streams' events)
        const int bad = adopt_staged_pair(c) || maps_writable(c);
        c->stream = own;
        if (bad) return fail(c0, "psm_compute_batch: %s", c->err.c_str());
        if (own != s) {
            if (!c->ev_batch) PSM_HIP(c0, hipEventCreateWithFlags(&c->ev_batch, hipEventDisableTiming));
            PSM_HIP(c0, hipEventRecord(c->ev_batch, own));
            PSM_HIP(c0, hipStreamWaitEvent(s, c->ev_batch, 0));
        }
    }

    // ---- the table of the pairs' pointers (device copy refreshed only when an entry changed) ----
    std::vector<PcPair> tab((size_t)n);
    for (int i = 0; i < n; ++i) {
        const psm_ctx *c = ctxs[i];
        PcPair &p = tab[i];
        memset(&p, 0, sizeof p);
        for (int k = 0; k < 2; ++k) { p.raw[k] = c->raw[k]; p.g[k] = c->g[k]; p.p4[k] = c->p4[k]; }
        p.scratch = c->gf_scratch;
        p.keys = c->keys_cur;
        p.maps = c->maps;
    }
    if (c0->batch_host.size() != tab.size() || memcmp(c0->batch_host.data(), tab.data(), tab.size() * sizeof(PcPair)) != 0) {
        if (c0->batch_cap < tab.size()) {
            PSM_HIP(c0, hipStreamSynchronize(s));
            (void)hipFree(c0->batch_tab);
            if (c0->batch_pin) (void)hipHostFree(c0->batch_pin);
            c0->batch_tab = nullptr;
            c0->batch_pin = nullptr;
            c0->batch_cap = 0;
            c0->batch_host.clear();          // (should an allocation below fail, the next call must not take the old table for current)
            PSM_HIP(c0, hipMalloc((void **)&c0->batch_tab, tab.size() * sizeof(PcPair)));
            PSM_HIP(c0, hipHostMalloc((void **)&c0->batch_pin, 2 * tab.size() * sizeof(PcPair), hipHostMallocDefault));
            c0->batch_cap = tab.size();
        }
        // (the table changes with every frame of a frame loop - the image slots alternate; the copy is stream-ordered behind the
        // previous frame's kernels, which still read the old table, and reads one of two page-locked slots: a slot is rewritten
        // only after the copy that read it has executed)
        const int slot = c0->batch_slot ^= 1;
        if (!c0->ev_tab[slot]) PSM_HIP(c0, hipEventCreateWithFlags(&c0->ev_tab[slot], hipEventDisableTiming));
        else PSM_HIP(c0, hipEventSynchronize(c0->ev_tab[slot]));
        PcPair *pin = c0->batch_pin + (size_t)slot * c0->batch_cap;
        memcpy(pin, tab.data(), tab.size() * sizeof(PcPair));
        c0->batch_host = tab;
        PSM_HIP(c0, hipMemcpyAsync(c0->batch_tab, pin, tab.size() * sizeof(PcPair), hipMemcpyHostToDevice, s));
        PSM_HIP(c0, hipEventRecord(c0->ev_tab[slot], s));
    }
    const PcPair *dt = c0->batch_tab;

    const int depth = c0->raw_depth;
    const size_t row = (size_t)W * 3 * (depth == PSM_IMG_F32 ? 4 : 1);
    const bool u8 = c0->dtype == PSM_U8;
    const bool whole = Dloc == c0->D;        // every slice here: the maps are final (else: packed minima for psm_disp_merge)
    double t1 = t0;
    auto enqueue = [&]() -> int {
        // ---- CostConst: CVC::preprocess of every image (the cost volumes stay virtual) - float mode: inside the guidance launch ----
        if (u8) {
            Prof p(c0, PSM_K_PREP);
            launch_prep_batch(s, dt, n, row, depth == PSM_IMG_F32, W, H, u8);
        }
        t1 = now_us();
        // ---- CostFilter: guidance of every image, the fused select kernel over every pair, the reduction ----
        {
            Prof p(c0, PSM_K_GUIDE);
            launch_guidance_batch(s, dt, n, W, H, row, u8 ? 0 : (depth == PSM_IMG_F32 ? 2 : 1));
        }
        {
            Prof p(c0, PSM_K_CVF_F);
            launch_cvf_select2_batch(s, c0->march, dt, n, W, H, n1, c0->d0, next_pc_stamp(c0), u8, two_phase ? 1 : 0, two_phase ? S : 1);
        }
        {
            Prof p(c0, PSM_K_WTA);
            launch_chunk_min2sides_batch(s, c0->march, dt, n, W, H, n1, whole && !(two_phase && n2 > 0));
        }
        if (two_phase && n2 > 0) {
            {
                Prof p(c0, PSM_K_CVF_F);
                launch_cvf_select_keys2_batch(s, c0->march, dt, n, W, H, n2, c0->d0, next_pc_stamp(c0), u8, 2, S);
            }
            if (whole) {
                Prof p(c0, PSM_K_MERGE);
                launch_merge_batch(s, dt, n, W, H);
            }
        }
        return check_launch(c0, "batch (prep, guidance, fused select filter, reduction)");
    };
#ifdef PSM_EXPERIMENTS      // (PSM_OPT_GRAPH: experiment builds only - the replay measured slower than the plain launches)
    if (c0->opt_graph && c0->opt_profile == 0) {
        // One graph per frame: the kernels read every pair through the device table (whose ADDRESS is all the graph holds), so the
        // captured launches stay valid while the batch size, the geometry and the options do - also across new image pairs.
        const long long sig[8] = {n, (long long)(size_t)dt, c0->march.flags, c0->march.seg_rows, depth, Dloc, c0->d0, (long long)W << 32 | H};
        if (!c0->batch_graph || memcmp(sig, c0->graph_sig, sizeof sig) != 0) {
            if (c0->batch_graph) { (void)hipGraphExecDestroy(c0->batch_graph); c0->batch_graph = nullptr; }
            hipGraph_t g = nullptr;
            PSM_HIP(c0, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            const int bad = enqueue();
            const hipError_t e = hipStreamEndCapture(s, &g);
            if (bad || e != hipSuccess) {
                if (g) (void)hipGraphDestroy(g);
                return bad ? 1 : fail(c0, "psm_compute_batch: graph capture failed: %s", hipGetErrorString(e));
            }
            const hipError_t ei = hipGraphInstantiate(&c0->batch_graph, g, nullptr, nullptr, 0);
            (void)hipGraphDestroy(g);
            if (ei != hipSuccess) { c0->batch_graph = nullptr; return fail(c0, "psm_compute_batch: hipGraphInstantiate failed: %s", hipGetErrorString(ei)); }
            memcpy(c0->graph_sig, sig, sizeof sig);
        }
        PSM_HIP(c0, hipGraphLaunch(c0->batch_graph, s));
        t1 = now_us();
    } else
#endif
    if (enqueue()) return 1;
    for (int i = 0; i < n; ++i)
        if (ctxs[i]->ev_free) PSM_HIP(c0, hipEventRecord(ctxs[i]->ev_free, s));     // the staged images have been read
    const double t2 = now_us();

    // ---- every context is now where psm_cost_construct + psm_cost_filter + psm_disp_select(ctx, NULL, NULL, 0) leave it ----
    PSM_HIP(c0, hipEventRecord(c0->ev_batch, s));
    for (int i = 0; i < n; ++i) {
        psm_ctx *c = ctxs[i];
        c->have_g1 = true; c->g1_y0 = 0; c->g1_y1 = H;
        c->have_guid[0] = c->have_guid[1] = true; c->guid_y0 = 0; c->guid_y1 = H;
        c->fgf_virtual[0] = c->fgf_virtual[1] = 0;
        c->raw_rows[0] = c->raw_rows[1] = psm_ctx::RAW_NONE;
        c->gf_virtual[0] = c->gf_virtual[1] = true;
        c->have_cost = true;
        c->vol_domain_ok[0] = c->vol_domain_ok[1] = true;      // (the costs are those of the images again, as after psm_cost_construct)
        c->have_keys = c->have_keys_side[0] = c->have_keys_side[1] = !whole;   // (a disparity shard: its minima are what psm_disp_merge_ctx takes)
        c->have_maps = whole;
        c->have_valid = false;
        c->maps_early = nullptr;
        c->have_rows = false; c->rows_y0 = 0; c->rows_y1 = H;
        if (c->stream != s) PSM_HIP(c0, hipStreamWaitEvent(c->stream, c0->ev_batch, 0));
    }
    if (!c0->opt_async) PSM_HIP(c0, hipStreamSynchronize(s));
    const double t3 = now_us();
    for (int i = 0; i < n; ++i) {      // the batch's wall time, split like the reference's three stage timers
        ctxs[i]->stage_us[PSM_STAGE_CVC] = t1 - t0;
        ctxs[i]->stage_us[PSM_STAGE_CVF] = t2 - t1;
        ctxs[i]->stage_us[PSM_STAGE_DISPSEL] = t3 - t2;
    }
    return 0;
}
