// psm_ctx.h - the context behind the C ABI of libprimesm_hip.so (include/primesm_hip.h) and the helpers its
// translation units share.  Internal to the library:
//   psm_api_core.cpp    context life cycle, options, uploads / downloads, timers
//   psm_api_filter.cpp  CostConst / CostFilter: what is built lazily, which form of the fused kernel runs, materialisation
//   psm_api_select.cpp  DispSelect: maps, packed minima, row stripes and disparity shards
//   psm_api_pp.cpp      post-processing: L-R check, invalid fill, weighted median
//   psm_api_batch.cpp   several Middlebury-size pairs per launch (psm_compute_batch)
// Takes the place of the reference's oclUtil + CVC_cl / CVF_cl / DispSel_cl host wrappers
// (src/oclUtil.cpp, src/CVC_cl.cpp, src/CVF_cl.cpp, src/DispSel_cl.cpp).
#pragma once
#include "../../include/primesm_hip.h"
#include "psm_kernels.h"

#include <string>
#include <utility>
#include <vector>

namespace psm {

struct KernelTimer {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    double total_ms = 0.0;
    int launches = 0;
};

}  // namespace psm

namespace psm {
// psm_share_streams: the contexts of a batch run on ONE main stream and ONE copy stream each way instead of three streams per
// context (the runtime multiplexes streams onto a few hardware queues; with 8 contexts' 24 streams every asynchronous copy cost
// 0.2 ms of host time).  Owned jointly: the last context to go destroys the streams.
struct StreamSet {
    hipStream_t main = nullptr, up = nullptr, down = nullptr;
    int refs = 0;
};
}  // namespace psm

struct psm_ctx {
    psm_ctx() { for (signed char &p : peer_ok) p = -1; }
    int W = 0, H = 0, D = 0, d0 = 0, d1 = 0, Dloc = 0, dtype = PSM_F32, device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipStream_t copy_stream = nullptr;  // psm_upload_pair_async / psm_download_maps_async: PCIe legs next to the kernels
    hipStream_t down_stream = nullptr;  // ... the D2H leg's stream: copy_stream unless the context shares a StreamSet
    psm::StreamSet *shared = nullptr;   // psm_share_streams
    hipEvent_t ev_up = nullptr, ev_maps = nullptr, ev_down = nullptr, ev_free = nullptr;

    // device memory (DESIGN.md "HBM layout")
    void *raw[2] = {nullptr, nullptr};  // staged copy of the interleaved host images
    void *raw_next[2] = {nullptr, nullptr};   // second image slot (psm_upload_pair_async), allocated on first use
    uint8_t *pin_up = nullptr;          // page-locked staging of the next pair (2 slots x 2 images, used alternately), on first use
    hipEvent_t ev_stage[2] = {nullptr, nullptr};   // ... the H2D copy out of a slot has executed (the host may refill it)
    int stage_slot = 0;
    int next_depth = -1;                // PSM_IMG_* of the pair in raw_next (-1: none pending)
    bool up_recorded = false;           // ev_up has been recorded at least once
    size_t raw_bytes = 0;
    int raw_depth = -1;                 // PSM_IMG_* of the staged pair, -1 = nothing uploaded
    psm::Guidance g[2] = {};
    void *vol[2] = {nullptr, nullptr};  // [Dloc][H][W] float (PSM_F32) or uint8 (PSM_U8)
    float *fvol = nullptr;              // PSM_U8 only: float work volume of one side
    float *spare = nullptr;             // PSM_F32: output volume of the fused filter (ping-pong with vol[side])
    float4 *ab = nullptr;               // [Dloc][H][W] {a0,a1,a2,b}; also box8 output
    long long *keys = nullptr;          // [2][H][W]
    long long *keys_cur = nullptr;      // where the packed minima go: `keys`, or the caller's buffer (psm_set_key_buffer)
    long long *gather = nullptr;        // [gather_ranks][2][H][W], psm_disp_merge_ctx
    int gather_ranks = 0;
    // page-locked bounce buffers of the single-process exchange when two devices cannot reach each other (gather_leg): two slots
    // used alternately - the device -> host copy of leg i + 1 runs while the host -> device copy of leg i is still in flight; a
    // slot is refilled only after the copy out of it has executed (ev_xfer)
    uint8_t *xfer_pin[2] = {nullptr, nullptr};
    size_t xfer_pin_bytes[2] = {0, 0};
    hipEvent_t ev_xfer[2] = {nullptr, nullptr};
    int xfer_slot = 0;
    signed char peer_ok[64];            // hipDeviceCanAccessPeer(this device, d), asked once per device (-1: not asked yet)
    int gather_staged_legs = 0;         // legs that went through the bounce buffers since the context was created (psm_gather_staged_legs: tests)
    uint8_t *maps = nullptr;            // [2][H][W]: maps_own, or the caller's buffer (psm_set_map_buffer)
    uint8_t *maps_own = nullptr;
    uint8_t *maps_early = nullptr;      // the map buffer the single-phase filter already filled from its final keys (k_chunk_min), or null
    // The maps / minima of the current frame cover the rows [rows_y0, rows_y1) only (a psm_set_rows stripe was in force when
    // psm_cost_filter produced them); have_rows false: whole image.  Recorded at filter time - psm_set_rows itself only
    // affects the NEXT filter - and cleared by everything that writes whole maps.
    bool have_rows = false;
    int rows_y0 = 0, rows_y1 = 0;
    uint8_t *valid = nullptr;           // [2][H][W]
    uint8_t *pinned = nullptr;          // [2][H][W] page-locked bounce buffer for map / mask downloads (on first use)
    uint8_t *pinned2 = nullptr;         // second bounce buffer: psm_download_maps_async of frame i while frame i-1 is being read
    int *wm = nullptr;                  // psm_wgt_median scratch: nxt[H][W+1], prog[H], err[1]; allocated on first use
    uint8_t *wm_par = nullptr;          // scratch of its parallel (sweep) form, per side: orig, newv, chgb, rowany (bytes), stamp, 2 active lists, changed list,
                                        // slot_of, the list of all invalid pixels; behind the two sides the per-sweep counters of both maps (one block:
                                        // one fill, one snapshot) - the layout WmPair points into (psm_kernels.h)
    float *wm_wts = nullptr;            // ... the 19 x 19 window weights of every invalid pixel (formed once per call, read by every evaluation)
    size_t wm_wts_n = 0;
    int *wm_pin = nullptr;              // page-locked snapshots of the sweep counters (two slots: a group's counters are read while the next group runs)
    hipEvent_t ev_wm[2] = {nullptr, nullptr};
    int wm_sweeps[2] = {0, 0};          // last call: sweeps until the fixed point (-1: dataflow form), evaluations
    long long wm_evals[2] = {0, 0};
    uint8_t *p4[2] = {nullptr, nullptr};  // PSM_U8 only: {c0,c1,c2,grad} words
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
    unsigned long long *pc_ts = nullptr;  // PSM_OPT_PROFILE 2: {first start, last end} device time stamps per k_cvf_pc launch
    int pc_ts_n = 0;                      // launches stamped since the last reset (slots: PC_TS_SLOTS)
    // psm_compute_batch (this context as the first of a batch): device table of the pairs' plane pointers, its host copy
    // (re-uploaded only when an entry changed) and the event the other contexts' streams wait for
    psm::PcPair *batch_tab = nullptr;
    std::vector<psm::PcPair> batch_host;
    psm::PcPair *batch_pin = nullptr;    // page-locked staging of the table, two slots used alternately (a pageable source would make
    size_t batch_cap = 0;                // the "asynchronous" copy wait for the stream to drain: one frame could not follow the other)
    int batch_slot = 0;
    hipEvent_t ev_tab[2] = {nullptr, nullptr};
    hipEvent_t ev_batch = nullptr;
    // PSM_OPT_GRAPH: the launches of a batch captured once as a hipGraph and replayed while nothing they depend on changes
    hipGraphExec_t batch_graph = nullptr;
    long long graph_sig[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float4 *fgf_mab[2] = {nullptr, nullptr};
    void *fgf = nullptr;                // psm_cost_filter_fgf scratch (small planes), fgf_bytes long
    size_t fgf_bytes = 0;

    bool have_images = false, have_g1 = false, have_cost = false, have_maps = false, have_valid = false;
    bool have_keys = false;             // keys_cur holds the packed minima of the current frame's local slices (both sides)
    bool have_keys_side[2] = {false, false};
    // raw_rows[side]: which rows of the unfiltered cost volume exist in memory.  psm_cost_construct may
    // leave the volume virtual (RAW_NONE): the fused filter builds the costs on the fly from the g1
    // planes.  Anything else that reads the volume materialises it first (materialize()).
    enum { RAW_ALL = 0, RAW_NONE = 1 };
    int raw_rows[2] = {RAW_ALL, RAW_ALL};

    // Domain of the scaled window sums (select forms of the fused kernel carry the 1/64 of both box filters as one 2^-12 at the
    // end: bit-identical to the per-sum scaling only while no intermediate under- or overflows).  8-bit images are always
    // inside; float images (non-zero |I| within 2^-10 .. 2^10) and uploaded cost volumes (2^-60 .. 2^60) are measured on the
    // device when they arrive; outside, psm_cost_filter runs the storing form (the oracle's arithmetic op for op) + k_wta.
    bool img_domain_ok = true, img_next_domain_ok = true, vol_domain_ok[2] = {true, true};
    unsigned *range_dev = nullptr;      // [4][2] exponent ranges (current pair, staged pair, volume L, volume R), on first use
    unsigned *range_pin = nullptr;      // page-locked copy
    bool range_next_pending = false;    // the staged pair's range is still on its way (read when the pair is adopted)

    // options
    int opt_async = 0, opt_variant = 0, opt_profile = 0, opt_graph = 0, opt_gather_staged = 0;
    psm::March march = {0, 4, 0};

    double stage_us[PSM_STAGE_COUNT] = {0, 0, 0, 0};
    psm::KernelTimer timers[PSM_K_COUNT];
    std::vector<hipEvent_t> event_pool;
    std::string err;
};

namespace psm {

int fail(psm_ctx *c, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

#define PSM_HIP(c, call)                                                                         \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) return psm::fail((c), "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

double now_us();
hipEvent_t get_event(psm_ctx *c);
int check_launch(psm_ctx *c, const char *what);
int end_stage(psm_ctx *c, int stage, double t0);
int bind(psm_ctx *c);
void adopt_new_pair(psm_ctx *c, int depth);   // a new image pair is current: nothing derived from the previous one survives
inline size_t velem(const psm_ctx *c) { return c->dtype == PSM_U8 ? 1 : 4; }
// Call before anything on c->stream writes c->maps: the buffer may still be the source of an asynchronous download
// (psm_download_maps_async on the copy stream) - the writer waits for that copy on the device, the host never blocks.
inline int maps_writable(psm_ctx *c)
{
    if (c->ev_down) PSM_HIP(c, hipStreamWaitEvent(c->stream, c->ev_down, 0));
    return 0;
}

// RAII bracket of one kernel launch with hipEvents on the launch stream (PSM_OPT_PROFILE 1)
struct Prof {
    psm_ctx *c;
    int k;
    hipEvent_t a = nullptr, b = nullptr;
    Prof(psm_ctx *c_, int k_) : c(c_), k(k_)
    {
        if (c->opt_profile == 1) {
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

// psm_api_core.cpp
int h2d_rows(psm_ctx *c, void *dst, const void *src, size_t row, size_t stride, int rows);   // host rows -> packed device rows
// psm_api_filter.cpp
int run_prep(psm_ctx *c, int ya = 0, int yb = 0);
int ensure_vol(psm_ctx *c, int side);
int ensure_ab(psm_ctx *c);
int ensure_spare(psm_ctx *c);
int fgf_flush(psm_ctx *c, int side);
int materialize(psm_ctx *c, int side);
int adopt_staged_pair(psm_ctx *c);               // the pair psm_upload_pair_async staged becomes the current one
int ensure_gf_scratch(psm_ctx *c, size_t bytes);
// psm_api_core.cpp: measure the exponent range of n floats on `stream` into slot (0..3) of the context's range buffers
int range_enqueue(psm_ctx *c, hipStream_t stream, int slot, const float *p0, size_t n0, const float *p1, size_t n1);
bool range_inside(const psm_ctx *c, int slot, int lo_exp, int hi_exp);     // after the stream has been synchronised
// PSM_FLAG_FMA_SOLVE applies to float mode only (8-bit contexts never carry the bit: psm_set_option)
inline bool fma_solve(const psm_ctx *c) { return c->dtype == PSM_F32 && (c->march.flags & PSM_FLAG_FMA_SOLVE) != 0; }
// psm_create_shard_strided: the slices of such a context exist in the select forms of the fused kernel only
inline bool strided(const psm_ctx *c) { return c->march.dstep != 1; }
#define PSM_NOT_STRIDED(c, what)                                                                                               \
    do {                                                                                                                        \
        if (psm::strided(c)) return psm::fail((c), "%s: a strided disparity shard (psm_create_shard_strided) runs the default select path only", (what)); \
    } while (0)
inline bool scaled_forms_ok(const psm_ctx *c) { return c->img_domain_ok && c->vol_domain_ok[0] && c->vol_domain_ok[1]; }
constexpr int PSM_IMG_EXP = 10, PSM_VOL_EXP = 60;
constexpr size_t PSM_COPY_KERNEL_MAX = (size_t)2 << 20;     // asynchronous PCIe legs up to this size go through k_copy16 instead of the copy engines
unsigned long long *next_pc_stamp(psm_ctx *c);   // slot of the next k_cvf_pc launch (NULL unless PSM_OPT_PROFILE 2)

// psm_api_select.cpp
int copy_maps_out(psm_ctx *c, const uint8_t *dev, uint8_t *lmap, uint8_t *rmap, size_t stride);

}  // namespace psm
