// psm_api_filter.cpp - CostConst and CostFilter behind the C ABI: what stays virtual (lazy cost volumes, the filtered
// volume as packed per-pixel minima or low-resolution FGF models), which form of the fused kernel runs, and how a virtual
// volume becomes real when something other than the WTA reads it.  Replaces CVC_cl::buildCV (src/CVC_cl.cpp:95-210) and
// CVF_cl::preprocess / filterCV (src/CVF_cl.cpp) as called by DispEst::CostConst_GPU / CostFilter_GPU
// (src/DispEst.cpp:272-276,299-308); the FGF entry follows DispEst::CostFilter_FGF (src/DispEst.cpp:281-296).
#include "psm_ctx.h"

#include <utility>

using namespace psm;

namespace psm {

// planarise + scale + gray + x-gradient of both staged images -> g1 (and the 8-bit planes)
// (rows [ya, yb) only - the kernels are row-independent - when a row stripe is all the following filter will read: have_g1
// then stays false and g1_y0/1 say what is there; every other consumer finds have_g1 false and prepares the whole image)
int run_prep(psm_ctx *c, int ya, int yb)
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
        launch_prep_u8(c->stream, (const uint8_t *)c->raw[s] + ya * row, row, c->W, yb - ya, c->p4[s] + 4 * o, c->g[s].g1 + o);
    }
    if (check_launch(c, "prep")) return 1;
    if (c->ev_free) PSM_HIP(c, hipEventRecord(c->ev_free, c->stream));   // the staged images have been read: their slot may be refilled
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

// The 16 B/voxel (a0,a1,a2,b) scratch is only needed by psm_filter_stage_a, the direct variant and psm_box8_volume
int ensure_ab(psm_ctx *c)
{
    if (c->ab) return 0;
    const size_t V = (size_t)c->W * c->H * c->Dloc;
    PSM_HIP(c, hipMalloc((void **)&c->ab, V * sizeof(float4)));
    return 0;
}

// second float volume for the fused filter when it has to READ a materialised cost volume (out of place)
int ensure_spare(psm_ctx *c)
{
    if (c->spare) return 0;
    const size_t V = (size_t)c->W * c->H * c->Dloc;
    PSM_HIP(c, hipMalloc((void **)&c->spare, V * sizeof(float)));
    return 0;
}

unsigned long long *next_pc_stamp(psm_ctx *c)
{
    if (c->opt_profile != 2) return nullptr;
    if (!c->pc_ts) {
        if (hipMalloc((void **)&c->pc_ts, 3 * (size_t)PC_TS_SLOTS * sizeof(unsigned long long)) != hipSuccess) {
            (void)hipGetLastError();
            c->pc_ts = nullptr;
            return nullptr;
        }
        (void)hipMemsetAsync(c->pc_ts, 0xff, PC_TS_SLOTS * sizeof(unsigned long long), c->stream);
        (void)hipMemsetAsync(c->pc_ts + PC_TS_SLOTS, 0, 2 * PC_TS_SLOTS * sizeof(unsigned long long), c->stream);
        c->pc_ts_n = 0;
    }
    if (c->pc_ts_n >= PC_TS_SLOTS) return nullptr;        // (read them with psm_filter_launch_times to start over)
    return c->pc_ts + c->pc_ts_n++;
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

}  // namespace psm

namespace psm {

// chunk planes of the select-mode fused kernel
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

// the pair psm_upload_pair_async staged becomes the current one: the kernels wait for its copy on the device
int adopt_staged_pair(psm_ctx *c)
{
    if (c->next_depth < 0) return 0;
    bool ok = true;
    if (c->range_next_pending) {        // float images: their range arrived with the copy (long done in a running frame loop)
        PSM_HIP(c, hipEventSynchronize(c->ev_up));
        ok = range_inside(c, 1, -PSM_IMG_EXP, PSM_IMG_EXP);
        c->range_next_pending = false;
    }
    PSM_HIP(c, hipStreamWaitEvent(c->stream, c->ev_up, 0));
    std::swap(c->raw[0], c->raw_next[0]);
    std::swap(c->raw[1], c->raw_next[1]);
    adopt_new_pair(c, c->next_depth);
    c->img_domain_ok = ok;
    c->next_depth = -1;
    return 0;
}

}  // namespace psm

namespace {

// build the (float) cost slices of `side` for rows [ybeg, yend)
void launch_cvc_rows(psm_ctx *c, int side, int ybeg, int yend)
{
    Prof p(c, PSM_K_CVC);
    // buildCV_right is called with the images swapped (src/DispEst.cpp:217,260)
    launch_cvc(c->stream, c->g[side].g1, c->g[1 - side].g1, (float *)c->vol[side], c->W, c->H, c->d0, c->Dloc, side, ybeg, yend);
}

// g1 (and the 8-bit planes) and the guidance of the whole image: a row-stripe filter leaves only its own rows behind
int ensure_whole_planes(psm_ctx *c)
{
    if (!c->have_g1 && run_prep(c)) return 1;
    if (!(c->have_guid[0] && c->have_guid[1])) {
        {
            Prof p(c, PSM_K_GUIDE);
            launch_guidance(c->stream, c->g[0], c->W, c->H, &c->g[1], 0, 0, fma_solve(c));
        }
        c->have_guid[0] = c->have_guid[1] = true;
        c->guid_y0 = 0;
        c->guid_y1 = c->H;
        return check_launch(c, "guidance");
    }
    return 0;
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
        launch_cvf_fused(c->stream, c->march, c->fvol, c->spare, c->g[side], c->W, c->H, c->Dloc, 0, c->H, c->g[1 - side].g1, c->d0, 0, next_pc_stamp(c));
    }
    launch_f32_to_u8(c->stream, c->spare, (uint8_t *)c->vol[side], V);   // q8 = sat_u8(rintf(q * 255))
    return check_launch(c, "cvf (8-bit, storing form)");
}

}  // namespace

namespace psm {

// make sure the whole volume of `side` (unfiltered, or filtered by psm_cost_filter / psm_cost_filter_fgf) is in memory
int materialize(psm_ctx *c, int side)
{
    PSM_NOT_STRIDED(c, "materialising a cost volume");
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
                             c->g[1 - side].g1, c->d0, 0, next_pc_stamp(c));
            std::swap(*(float **)&c->vol[side], c->spare);
        } else {
            launch_cvf_fused(c->stream, c->march, nullptr, (float *)c->vol[side], c->g[side], c->W, c->H, c->Dloc, 0, c->H,
                             c->g[1 - side].g1, c->d0, 1 + side, next_pc_stamp(c));
        }
        c->gf_virtual[side] = false;
        c->raw_rows[side] = psm_ctx::RAW_ALL;                 // vol[side] now holds real (filtered) data
        return check_launch(c, "cvf (materialize)");
    }
    if (c->raw_rows[side] == psm_ctx::RAW_ALL) return 0;
    if (ensure_vol(c, side)) return 1;
    launch_cvc_rows(c, side, 0, c->H);
    c->raw_rows[side] = psm_ctx::RAW_ALL;
    return check_launch(c, "cvc (materialize)");
}

}  // namespace psm

namespace {

// One volume: the path for cost volumes that exist in memory (psm_upload_volume, PSM_FLAG_MATERIALISE_COSTS), for the storing
// form, the direct variant and for stage A alone.
int filter_side(psm_ctx *c, int side, bool stage_b)
{
    PSM_NOT_STRIDED(c, "filtering one side / a materialised volume / the storing form");
    c->maps_early = nullptr;
    const size_t V = (size_t)c->W * c->H * c->Dloc;
    const int W = c->W, H = c->H;
    if (!c->have_g1 && run_prep(c)) return 1;  // volume came from psm_upload_volume
    if (fgf_flush(c, side)) return 1;
    if (c->gf_virtual[side] && materialize(c, side)) return 1;   // filtering an already filtered (virtual) volume: make it real first
    if (!c->have_guid[side]) {
        // the guidance of BOTH images in one launch the first time either side asks (the other side's call then finds it)
        Prof p(c, PSM_K_GUIDE);
        launch_guidance(c->stream, c->g[0], W, H, &c->g[1], 0, 0, fma_solve(c));
        c->have_guid[0] = c->have_guid[1] = true;
    }
    // Default: the fused kernel in "select" mode - the WTA over the local slices runs inside the filter, the filtered
    // volume stays virtual (PSM_FLAG_STORE_FILTERED forces the storing form; the direct variant is its own filter)
    const bool sel8 = c->dtype == PSM_U8 && c->raw_rows[side] != psm_ctx::RAW_ALL;   // 8-bit mode: select form only with costs on the fly
    if (stage_b && (c->dtype == PSM_F32 || sel8) && c->opt_variant == 0 && !(c->march.flags & PSM_FLAG_STORE_FILTERED) && scaled_forms_ok(c)) {
        const bool lazy = c->raw_rows[side] != psm_ctx::RAW_ALL;
        const PcPlan pl = pc_plan(W, c->march.rows(H), c->Dloc, c->march.seg_rows, PC_PLANES);
        if (ensure_gf_scratch(c, pl.scratch_bytes())) return 1;
        const size_t HW = (size_t)W * H;
        {
            Prof p(c, PSM_K_CVF_F);
            launch_cvf_select(c->stream, c->march, lazy ? nullptr : (const float *)c->vol[side], c->g[side], W, H, c->Dloc, c->g[1 - side].g1,
                              c->d0, lazy ? 1 + side : 0, c->gf_scratch, next_pc_stamp(c), sel8 ? c->p4[side] : nullptr, sel8 ? c->p4[1 - side] : nullptr);
        }
        {
            Prof p(c, PSM_K_WTA);
            launch_chunk_min(c->stream, c->march, W, H, c->Dloc, c->gf_scratch, c->keys_cur + side * HW, nullptr);
        }
        c->gf_virtual[side] = true;
        return check_launch(c, "cvf (fused, select mode)");
    }
    if (c->dtype == PSM_F32 && ensure_vol(c, side)) return 1;
    if (c->dtype == PSM_U8 && stage_b && c->raw_rows[side] != psm_ctx::RAW_ALL && materialize(c, side)) return 1;   // storing forms read the 8-bit volume
    float *fv = (float *)c->vol[side];
    if (c->dtype == PSM_U8) {
        // (8-bit mode: a float copy of the volume goes through the same kernels and is re-quantised afterwards)
        if (!c->fvol) PSM_HIP(c, hipMalloc((void **)&c->fvol, V * sizeof(float)));
        fv = c->fvol;
        launch_u8_to_f32(c->stream, (const uint8_t *)c->vol[side], fv, V);
    }
    const bool fused = stage_b && c->opt_variant == 0;
    if (!fused && fma_solve(c))
        return fail(c, "PSM_FLAG_FMA_SOLVE: stage A alone (psm_filter_stage_a) and the direct kernel variant exist in the canonical arithmetic only");
    if (!fused && materialize(c, side)) return 1;   // stage A alone / the direct variant read a real cost volume
    if (fused) {
        if (c->raw_rows[side] != psm_ctx::RAW_ALL) {
            // producer/consumer kernel on a virtual cost volume: nothing is read from vol[side], so the
            // filtered volume is written straight into it
            {
                Prof p(c, PSM_K_CVF_F);
                launch_cvf_fused(c->stream, c->march, nullptr, fv, c->g[side], W, H, c->Dloc, 0, H, c->g[1 - side].g1, c->d0, 1 + side, next_pc_stamp(c));
            }
            c->raw_rows[side] = psm_ctx::RAW_ALL;   // vol[side] now holds real (filtered) data
            return check_launch(c, "cvf (fused, lazy costs)");
        }
        // producer/consumer kernel reading a materialised cost volume: out of place, all rows in one launch
        if (ensure_spare(c)) return 1;
        float *out = c->spare;
        {
            Prof p(c, PSM_K_CVF_F);
            launch_cvf_fused(c->stream, c->march, fv, out, c->g[side], W, H, c->Dloc, 0, H, c->g[1 - side].g1, c->d0, 0, next_pc_stamp(c));
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
    if (stage_b) {          // (direct variant only: the marching stage B lives in the fused kernel)
        {
            Prof p(c, PSM_K_CVF_B);
            launch_cvf_b_direct(c->stream, c->ab, fv, c->g[side], W, H, c->Dloc);
        }
        if (c->dtype == PSM_U8) launch_f32_to_u8(c->stream, fv, (uint8_t *)c->vol[side], V);
    }
    return check_launch(c, "cvf");
}

// Both volumes per launch: guidance of both images, select-mode fused filter of both volumes, chunk reduction of both - five
// launches per frame instead of twelve.  The default path (costs built on the fly, both sides fresh).
bool can_filter_both(const psm_ctx *c)
{
    return c->opt_variant == 0 && !(c->march.flags & PSM_FLAG_STORE_FILTERED) && scaled_forms_ok(c) &&
           c->raw_rows[0] != psm_ctx::RAW_ALL && c->raw_rows[1] != psm_ctx::RAW_ALL && !c->gf_virtual[0] && !c->gf_virtual[1] &&
           !c->fgf_virtual[0] && !c->fgf_virtual[1];
}

int filter_both(psm_ctx *c)
{
    c->maps_early = nullptr;
    const bool striped = c->march.yend > c->march.ybeg;
    {   // g1 rows this launch reads: everything, or the stripe's rows - 8 .. + 8; guidance rows: the model rows y0 - 4 .. y1 + 2
        const int ya = striped ? (c->march.ybeg - 8 > 0 ? c->march.ybeg - 8 : 0) : 0;
        const int yb = striped ? (c->march.yend + 8 < c->H ? c->march.yend + 8 : c->H) : c->H;
        const int gy0 = striped ? (c->march.ybeg - 4 > 0 ? c->march.ybeg - 4 : 0) : 0;
        const int gy1 = striped ? (c->march.yend + 4 < c->H ? c->march.yend + 4 : c->H) : c->H;
        const bool need_g1 = !c->have_g1 && !(c->g1_y1 > c->g1_y0 && c->g1_y0 <= ya && c->g1_y1 >= yb);
        const bool need_guid = !(c->have_guid[0] && c->have_guid[1]) && !(c->guid_y1 > c->guid_y0 && c->guid_y0 <= gy0 && c->guid_y1 >= gy1);
        if (need_g1 && need_guid && c->dtype == PSM_F32) {
            // image planes AND guidance in one launch, straight from the staged images (psm_cost_construct left the preparation to
            // us): rows [ya, yb) of g1 are written, and the guidance of the same rows (a stripe: 4 rows more either side than it needs)
            const size_t row = (size_t)c->W * 3 * (c->raw_depth == PSM_IMG_F32 ? 4 : 1);
            {
                Prof p(c, PSM_K_GUIDE);
                launch_guidance(c->stream, c->g[0], c->W, c->H, &c->g[1], ya, yb, fma_solve(c), c->raw[0], c->raw[1], row, c->raw_depth == PSM_IMG_F32);
            }
            if (check_launch(c, "prep + guidance")) return 1;
            if (c->ev_free) PSM_HIP(c, hipEventRecord(c->ev_free, c->stream));   // the staged images have been read: their slot may be refilled
            const bool whole = ya == 0 && yb == c->H;
            c->have_g1 = whole;
            c->g1_y0 = c->guid_y0 = ya;
            c->g1_y1 = c->guid_y1 = yb;
            c->have_guid[0] = c->have_guid[1] = whole;
        } else {
            if (need_g1 && (striped ? run_prep(c, c->march.ybeg - 8, c->march.yend + 8) : run_prep(c))) return 1;
            // a row stripe needs the guidance of its model rows only (have_guid stays false: the planes are not whole, any other
            // consumer recomputes them; guid_y0/1 remember what is there for the next frame's check)
            if (!(c->have_guid[0] && c->have_guid[1]) && !(c->guid_y1 > c->guid_y0 && c->guid_y0 <= gy0 && c->guid_y1 >= gy1)) {
                Prof p(c, PSM_K_GUIDE);
                launch_guidance(c->stream, c->g[0], c->W, c->H, &c->g[1], gy0, gy1, fma_solve(c));
                c->guid_y0 = gy0;
                c->guid_y1 = gy1;
                if (gy0 == 0 && gy1 == c->H) c->have_guid[0] = c->have_guid[1] = true;
            }
        }
    }
    const uint8_t *const *p4 = c->dtype == PSM_U8 ? c->p4 : nullptr;
    // Two-phase selection (default from 112 local slices up - measured: -10 % at 1080p x 256, -13 % at 4K x 256, -4 % at
    // 720p x 128, worse at 64 slices and below; PSM_FLAG_TWO_PHASE_ON / _OFF force it for any Dloc >= 2 / disable it): every
    // S-th slice goes through the minima planes -> k_chunk_min -> keys; the other slices then run against that seeded key
    // plane (key form: one key load per voxel, an atomic only where a slice beats the current minimum - rare after the
    // seeding), so they write no planes and need no reduction.  S = pc_seed_stride: 8 since round 6 (5, and 4 from 4 Mpixel up,
    // while the key loads came from the memory side).
    const bool two_phase = !(c->march.flags & PSM_FLAG_TWO_PHASE_OFF) && c->Dloc >= 2 && (c->Dloc >= 112 || (c->march.flags & PSM_FLAG_TWO_PHASE_ON));
    if (two_phase) {
        const int S = pc_seed_stride(c->W, c->march.rows(c->H), c->dtype == PSM_U8);
        const int n1 = (c->Dloc + S - 1) / S, n2 = c->Dloc - n1;
        const PcPlan pl1 = pc_plan(c->W, c->march.rows(c->H), n1, c->march.seg_rows, PC_PLANES | PC_BOTH, 1, c->march.inflight);
        if (ensure_gf_scratch(c, 2 * pl1.scratch_bytes())) return 1;
        {
            Prof p(c, PSM_K_CVF_F);
            launch_cvf_select2(c->stream, c->march, c->g, c->W, c->H, n1, c->d0, c->gf_scratch, next_pc_stamp(c), p4, 1, S);
        }
        {
            Prof p(c, PSM_K_WTA);
            launch_chunk_min2sides(c->stream, c->march, c->W, c->H, n1, c->gf_scratch, c->keys_cur, nullptr);
        }
        if (n2 > 0) {
            Prof p(c, PSM_K_CVF_F);
            launch_cvf_select_keys2(c->stream, c->march, c->g, c->W, c->H, n2, c->d0, c->keys_cur, next_pc_stamp(c), p4, 0, 2, S);
        }
        c->gf_virtual[0] = c->gf_virtual[1] = true;
        return check_launch(c, "cvf (fused, select mode, two phases, both volumes)");
    }
    const PcPlan pl = pc_plan(c->W, c->march.rows(c->H), c->Dloc, c->march.seg_rows, PC_PLANES | PC_BOTH, 1, c->march.inflight);
    if (ensure_gf_scratch(c, 2 * pl.scratch_bytes())) return 1;
    {
        Prof p(c, PSM_K_CVF_F);
        launch_cvf_select2(c->stream, c->march, c->g, c->W, c->H, c->Dloc, c->d0, c->gf_scratch, next_pc_stamp(c), p4);
    }
    {
        // The reduction of the planes is the last thing that touches the keys: when the context holds every slice it writes the
        // maps (the low byte of each key) in the same pass, and psm_disp_select has no kernel left to launch.
        uint8_t *const early = c->Dloc == c->D ? c->maps : nullptr;
        if (early && maps_writable(c)) return 1;   // (still the source of the last frame's download?)
        Prof p(c, PSM_K_WTA);
        launch_chunk_min2sides(c->stream, c->march, c->W, c->H, c->Dloc, c->gf_scratch, c->keys_cur, early);
        c->maps_early = early;
    }
    c->gf_virtual[0] = c->gf_virtual[1] = true;
    return check_launch(c, "cvf (fused, select mode, both volumes)");
}

}  // namespace

extern "C" {

int psm_cost_construct(psm_ctx *c)
{
    if (!c) return 1;
    if (bind(c)) return 1;
    if (adopt_staged_pair(c)) return 1;
    if (!c->have_images) return fail(c, "psm_cost_construct: no image pair uploaded");
    const double t0 = now_us();
    // Lazy cost volume: when the fused filter will consume the costs (marching kernels, PSM_FLAG_MATERIALISE_COSTS not set)
    // they are built inside that kernel and never written to HBM.
    // (8-bit mode: lazy only when the select-mode kernel will consume the costs - its storing form reads a float copy)
    const bool lazy = c->opt_variant == 0 && !(c->march.flags & PSM_FLAG_MATERIALISE_COSTS) &&
                      (c->dtype == PSM_F32 || !(c->march.flags & PSM_FLAG_STORE_FILTERED));
    // CVC::preprocess belongs to this stage (src/DispEst.cpp:232-233).  A row stripe [y0, y1) with lazy costs reads the image
    // planes of rows y0 - 8 .. y1 + 7 only (costs of the model rows y0 - 4 .. y1 + 2, +- 4 for their box sums, and the guidance)
    if (!lazy) PSM_NOT_STRIDED(c, "psm_cost_construct with materialised costs");
    const bool striped = c->march.yend > c->march.ybeg;
    if (lazy && c->dtype == PSM_F32 && c->march.inflight <= 1) {
        // CVC::preprocess is lazy too (round 6): with the cost volume virtual, the first thing that needs the image planes is the
        // guidance precompute of psm_cost_filter - and k_guide_march forms them itself from the staged images (one launch instead of
        // k_prep + k_guide_march; launch_guidance with raw images).  Everything else that reads g1 finds have_g1 false and runs
        // run_prep first, as after a striped frame.  (Not with frames in flight on other streams - PSM_OPT_FRAMES_IN_FLIGHT: the merged
        // launch carries the conversions in all three waves of its workgroups and stretches beside another frame's VALU-bound fused
        // kernel - 450 x 375 x 64, two frames in flight: 0.226 ms against 0.211 with k_prep + k_guide_march; alone it is 0.270 vs 0.279.)
        c->have_g1 = false;
        c->g1_y0 = c->g1_y1 = 0;
        c->have_guid[0] = c->have_guid[1] = false;
        c->guid_y0 = c->guid_y1 = 0;
    } else if (striped && lazy ? run_prep(c, c->march.ybeg - 8, c->march.yend + 8) : run_prep(c)) return 1;
    c->fgf_virtual[0] = c->fgf_virtual[1] = 0;   // a new cost volume replaces whatever was pending
    c->gf_virtual[0] = c->gf_virtual[1] = false;
    c->vol_domain_ok[0] = c->vol_domain_ok[1] = true;   // (uploaded volumes are gone; the costs now follow from the images)
    c->maps_early = nullptr;
    for (int s = 0; s < 2; ++s) {
        if (lazy) {
            c->raw_rows[s] = psm_ctx::RAW_NONE;
        } else if (c->dtype == PSM_U8) {
            Prof p(c, PSM_K_CVC);
            launch_cvc_u8(c->stream, c->p4[s], c->p4[1 - s], (uint8_t *)c->vol[s], c->W, c->H, c->d0, c->Dloc, s);
            c->raw_rows[s] = psm_ctx::RAW_ALL;
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
                                    "(no variant / storing flag, cost volumes not materialised, images / volumes inside the select forms' domain)");
        for (int s = 0; s < 2; ++s)
            if (filter_side(c, s, true)) return 1;
    }
    c->have_maps = false;
    // the minima (and the maps made from them) describe the stripe that is in force NOW, whatever psm_set_rows says later
    c->have_rows = striped;
    c->rows_y0 = striped ? c->march.ybeg : 0;
    c->rows_y1 = striped ? c->march.yend : c->H;
    return end_stage(c, PSM_STAGE_CVF, t0);
}

int psm_cost_filter_side(psm_ctx *c, int side)
{
    if (!c) return 1;
    if (side != PSM_LEFT && side != PSM_RIGHT) return fail(c, "psm_cost_filter_side: bad side %d", side);
    if (!c->have_cost) return fail(c, "psm_cost_filter_side: no cost volume");
    if (!c->have_images) return fail(c, "psm_cost_filter_side: no image pair uploaded (guidance)");
    if (c->march.yend > c->march.ybeg) return fail(c, "psm_cost_filter_side: row stripes (psm_set_rows) go through psm_cost_filter");
    if (bind(c)) return 1;
    const double t0 = now_us();
    if (filter_side(c, side, true)) return 1;
    c->have_maps = false;
    c->have_rows = false;
    c->rows_y0 = 0;
    c->rows_y1 = c->H;
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
    // (the low-resolution models of a stripe would need their own halo arithmetic; a striped host filters whole images here)
    if (c->march.yend > c->march.ybeg) return fail(c, "psm_cost_filter_fgf: row stripes (psm_set_rows) are not supported by the Fast Guided Filter path");
    PSM_NOT_STRIDED(c, "psm_cost_filter_fgf");
    const int ws = c->W / sub, hs = c->H / sub, rad = 8 / sub;
    if (ws <= rad || hs <= rad) return fail(c, "psm_cost_filter_fgf: %dx%d too small for subsample_rate %d", c->W, c->H, sub);
    if (bind(c)) return 1;
    c->maps_early = nullptr;
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
    // PSM_FLAG_FGF_STORE: always write the filtered volume (default: it stays virtual until something other than the WTA reads it)
    const bool keep_virtual = fgf_can_fuse_wta(c->W) && !(c->march.flags & PSM_FLAG_FGF_STORE);
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
    c->have_rows = false;                       // whole-image results
    c->rows_y0 = 0;
    c->rows_y1 = c->H;
    return end_stage(c, PSM_STAGE_CVF, t0);
}

int psm_filter_stage_a(psm_ctx *c, int side)
{
    if (!c) return 1;
    if (side != PSM_LEFT && side != PSM_RIGHT) return fail(c, "psm_filter_stage_a: bad side %d", side);
    if (!c->have_cost || !c->have_images) return fail(c, "psm_filter_stage_a: needs images and a cost volume");
    if (bind(c)) return 1;
    c->maps_early = nullptr;
    if (filter_side(c, side, false)) return 1;
    PSM_HIP(c, hipStreamSynchronize(c->stream));
    return 0;
}

}  // extern "C"
