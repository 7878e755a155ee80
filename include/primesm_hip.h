/*
 * primesm_hip.h - C ABI of libprimesm_hip.so: the MI355X (gfx950) implementation of the
 * PRiMEStereoMatch DispEst hot path (CVC cost build -> CVF guided-image-filter aggregation
 * -> DispSel WTA) that takes the place of the reference's OpenCL side.
 *
 * Every entry point names the reference interface it replaces (paths relative to the
 * reference repository).  Conventions are those of the reference's `_cl` classes:
 *   - every function returns 0 on success and non-zero on failure (CVC_cl::buildCV,
 *     src/CVC_cl.cpp:185-210); the message is available from psm_last_error();
 *   - calls are synchronous on return unless PSM_OPT_ASYNC is set (the reference issues
 *     clFinish after each launch, src/CVC_cl.cpp:193);
 *   - the caller owns host memory, the context owns all device memory
 *     (DispEst owns memoryObjects[12], src/DispEst.cpp:88-128,159-160);
 *   - a context is used from one thread at a time (the reference calls from a single
 *     worker thread, src/main.cpp:42,64-73).
 * Plain pointers and sizes only; no C++/torch types cross this boundary.  The library is
 * meant to be dlopen()ed by the host program (hipUtil in primestereomatch_amd/host takes
 * the place of oclUtil) the way the reference defers .cl compilation to run time
 * (src/oclUtil.cpp:438-496).
 */
#ifndef PRIMESM_HIP_H
#define PRIMESM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSM_ABI_VERSION 1

typedef struct psm_ctx psm_ctx;

/* element type of the cost volume ("float mode" / "8-bit char mode") */
enum { PSM_F32 = 0, PSM_U8 = 1 };
/* element type of host images handed to psm_upload_pair.  PSM_IMG_F32 images are used as they are (the reference hands
 * DispEst images already scaled by 1/255, src/StereoMatch.cpp:195-198).  The default (select) forms of psm_cost_filter carry the
 * 1/64 of the box filters as one exact power of two at the end, which is bit-identical to the reference's per-sum scaling while
 * no intermediate under- or overflows: float images whose non-zero magnitudes leave 2^-10 .. 2^10 (measured on the device when
 * they arrive) make psm_cost_filter run its storing form - the reference's arithmetic op for op, ~25 % slower - automatically. */
enum { PSM_IMG_U8 = 0, PSM_IMG_F32 = 1 };
/* volume side: the reference's buffers CV_LCV / CV_RCV (include/ComFunc.h:65) */
enum { PSM_LEFT = 0, PSM_RIGHT = 1 };
/* stages, in the order StereoMatch::compute times them (src/StereoMatch.cpp:225-242) */
enum { PSM_STAGE_CVC = 0, PSM_STAGE_CVF = 1, PSM_STAGE_DISPSEL = 2, PSM_STAGE_PP = 3,
       PSM_STAGE_COUNT = 4 };
/* kernels whose device time can be queried with psm_kernel_time_ms() */
enum { PSM_K_PREP = 0, PSM_K_CVC = 1, PSM_K_GUIDE = 2, PSM_K_CVF_A = 3, PSM_K_CVF_B = 4,
       PSM_K_WTA = 5, PSM_K_MERGE = 6, PSM_K_BOX = 7, PSM_K_LRC = 8, PSM_K_CVF_F = 9, PSM_K_FGF = 10, PSM_K_WMF = 11,
       PSM_K_COUNT = 12 };
/* options for psm_set_option */
enum {
    PSM_OPT_ASYNC = 0,          /* 1: stage calls only enqueue; use psm_synchronize()        */
    PSM_OPT_KERNEL_VARIANT = 1, /* 0: marching kernels (default), 1: direct per-voxel kernels */
    PSM_OPT_PROFILE = 2,        /* 1: bracket every kernel launch with hipEvents (psm_kernel_time_ms); 2: the fused filter
                                   kernel stamps its own start / end instead (psm_filter_launch_times) */
    PSM_OPT_SEG_ROWS = 3,       /* rows per y-segment of the marching kernels (0 = auto)      */
    PSM_OPT_WAVES = 4,          /* waves (disparity slices) per workgroup: 1,2,4,8            */
    PSM_OPT_FLAGS = 5,          /* PSM_FLAG_* bits below; no flag but PSM_FLAG_F32_TOL / PSM_FLAG_FMA_SOLVE changes any result */
    PSM_OPT_GRAPH = 6,          /* retired: 1 is refused by the product library (hipGraph replay of a batch's launches measured slower
                                   than the launches themselves on this runtime; experiment builds keep it), 0 is accepted */
    PSM_OPT_GATHER_STAGED = 7,  /* 1 (on the ROOT context): psm_gather_rows_ctx / psm_disp_merge_ctx move every stripe / shard through
                                   page-locked host memory instead of device / peer copies - what they do on their own between two
                                   devices for which hipDeviceCanAccessPeer says no; the option forces that path (test hook) */
    PSM_OPT_FRAMES_IN_FLIGHT = 8 /* F >= 1 (default 1): this context is one of F contexts of the same geometry whose frames are
                                   filtered at the same time, each on its own stream (the frame loop of src/main.cpp:64-73 with F
                                   frames queued).  A hint for the planner of the fused launches only - how the work is cut into
                                   segments and chunks; no result changes */
};

/* PSM_OPT_FLAGS bits.  The default (0) is the product path: cost volumes and filtered volumes stay virtual, the fused
 * kernel runs CVC + CVF + WTA in one pass (two phases from 112 local slices up).  The flags select the forms that
 * materialise a volume - what a host that reads volumes gets anyway, on demand - and test hooks. */
enum psm_flag {
    PSM_FLAG_MATERIALISE_COSTS = 128,   /* psm_cost_construct always writes the cost volumes (default: they stay virtual
                                           and the fused filter builds the costs on the fly; any other reader
                                           materialises them first) */
    PSM_FLAG_FGF_STORE = 4096,          /* psm_cost_filter_fgf always writes the filtered volumes (default: they stay
                                           virtual - low-resolution models - and the WTA consumes those directly) */
    PSM_FLAG_STORE_FILTERED = 8192,     /* psm_cost_filter always writes the filtered volumes (storing form of the fused
                                           kernel + a separate WTA; default: select forms, packed per-pixel minima) */
    PSM_FLAG_TWO_PHASE_ON = 1048576,    /* force / disable the two-phase selection of psm_cost_filter (default: on from */
    PSM_FLAG_TWO_PHASE_OFF = 2097152,   /* 112 local slices - every 8th (short stripes, large 8-bit images: 5th) slice through minima planes, the rest against
                                           the seeded key plane) */
    PSM_FLAG_WMF_DATAFLOW = 4194304,    /* psm_wgt_median runs its row-dataflow form only */
    PSM_FLAG_WMF_TWO_SWEEPS = 8388608,  /* ... at most 2 sweeps of its parallel form (test hook for the fall-back) */
    PSM_FLAG_WMF_NO_CACHE = 16777216,   /* ... without the cache of window weights (1.5 KB of device memory per invalid pixel):
                                           every evaluation forms its weights itself; same maps */
    PSM_FLAG_F32_TOL = 33554432,        /* float mode, default (select) path of psm_cost_filter: the TOLERANCE form of the fused kernel -
                                           level 1 of its horizontal window sums in fp32 instead of fp64 (fewer four-cycle
                                           instructions).  The ONLY flag that changes results: filtered costs within 1e-4 of
                                           the default form's (measured: <= 4e-5, no disparity changed on the test pairs), not
                                           bit-identical.  The storing form (psm_download_volume ...) stays exact; 8-bit mode
                                           ignores the flag.  Off by default. */
    PSM_FLAG_FMA_SOLVE = 67108864,      /* float mode: the 3x3 solve of the guided filter (src/CVF.cpp:129-147) with the fused
                                           multiply-adds GCC's default -ffp-contract=fast forms on an FMA target (the ARM boards
                                           the reference ran on): every x*y - z*w of the minors and of DET and every s + x*y of
                                           the three accumulations is one fma.  Bit-identical to the oracle's reading
                                           PSMO_VAR_FMA_SOLVE of those lines (maps, minima and filtered volumes); within 4e-4 of
                                           the default (the canon: the same lines compiled without contraction, e.g. x86-64).
                                           Applies to psm_cost_filter (select and storing forms); psm_compute_batch, the FGF
                                           variant and 8-bit mode refuse / ignore it.  Off by default. */
    PSM_FLAGS_ALL = 128 | 4096 | 8192 | 1048576 | 2097152 | 4194304 | 8388608 | 16777216 | 33554432 | 67108864
};

/* Number of usable HIP devices; 0 if none.  Replaces openCLdevicepoll()
 * (src/oclUtil.cpp:18-135) whose result StereoMatch receives as gotOCLDev
 * (src/main.cpp:29,37). */
int psm_device_count(void);

/* Create a context for W x H images and disparities [0, max_disp): device selection,
 * stream, the device buffers of enum buff_id (8 image/gradient planes, 2 volumes, 2 maps;
 * include/ComFunc.h:65, src/DispEst.cpp:88-128) and the CVF intermediates
 * (src/CVF_cl.cpp:115-159).  Replaces createContext/createCommandQueue/clCreateBuffer and
 * the three `_cl` constructors (src/DispEst.cpp:57-140).
 * dtype: PSM_F32 | PSM_U8.  8 <= W,H; 1 <= max_disp <= 256 (maps are 8-bit,
 * src/DispSel.cpp:105). */
int psm_create(psm_ctx **out, int width, int height, int max_disp, int dtype, int device);

/* As psm_create, for one rank of a disparity-sharded job: this context holds only the
 * slices d in [d_begin, d_end) of both volumes.  0 <= d_begin < d_end <=
 * max_disp. */
int psm_create_shard(psm_ctx **out, int width, int height, int max_disp, int d_begin, int d_end,
                     int dtype, int device);
/* The same job cut the other way (round 6): this context holds the slices d_first, d_first + d_step, d_first + 2 d_step, ... <
 * max_disp of both volumes - rank g of G owns d = g (mod G).  A rank's slices then span the whole disparity range, so the
 * slices that seed its key plane bound EVERY pixel's minimum and the two-phase selection works on a shard as it does on the
 * whole volume (a contiguous shard's seeds bound little: most pixels' minima lie in other ranks' slices).  The merge
 * (psm_disp_merge / psm_disp_merge_ctx: a signed minimum of packed keys) does not care how the slices were dealt
 * (DispSel::CVSelect, src/DispSel.cpp:96-104, is a minimum over d in any order with ties to the lowest d).  Such a context
 * runs the default select path only: CostConst / CostFilter / psm_disp_select_partial; everything that reads or writes a
 * volume refuses it.  0 <= d_first < max_disp, d_step >= 1 (1: the contiguous shard [d_first, max_disp)). */
int psm_create_shard_strided(psm_ctx **out, int width, int height, int max_disp, int d_first, int d_step,
                             int dtype, int device);

/* Releases everything the context owns (DispEst::~DispEst, src/DispEst.cpp:143-162). */
void psm_destroy(psm_ctx *ctx);

/* Last error text of this context (ctx == NULL: last creation error).  Replaces
 * checkSuccess/errorNumberToString (include/oclUtil.h:63-71, src/oclUtil.cpp:582-683). */
const char *psm_last_error(const psm_ctx *ctx);

int psm_set_option(psm_ctx *ctx, int option, int value);
/* Run all work of this context on an existing hipStream_t (NULL = the context's own). */
int psm_set_stream(psm_ctx *ctx, void *hip_stream);
int psm_synchronize(psm_ctx *ctx);
/* Give back the device / page-locked memory the context allocated ON FIRST USE and that holds no state between calls: the
 * weighted median's sweep scratch and weight cache (1.5 KB per invalid pixel), the minima planes of the fused filter, the gather /
 * bounce buffers of the single-process exchange, the 8-bit storing path's float work volume.  Synchronises the context first;
 * everything is allocated again by the call that needs it.  Maps, minima, volumes and images are untouched.  (The reference
 * allocates its CVF intermediates per call, src/CVF_cl.cpp:115-159; this library keeps them - this is how a long-running host
 * gets the memory back without destroying the context.) */
int psm_release_scratch(psm_ctx *ctx);

/* Copy a stereo pair to the device and planarise it.  Replaces the host half of
 * CVC_cl::buildCV (split + 8x map/memcpy, src/CVC_cl.cpp:95-160) and
 * DispEst::setInputImages (src/DispEst.cpp:164-170).  l, r: H rows of W interleaved
 * `channels`(=3, cv::imread order B,G,R) pixels, row pitch stride_bytes.
 * depth PSM_IMG_U8: values are scaled by 1/255.0f on the device exactly as
 * convertTo(CV_32F, 1/255.0f) does (src/StereoMatch.cpp:195-196); PSM_IMG_F32: used as is. */
int psm_upload_pair(psm_ctx *ctx, const void *l, const void *r, int channels, size_t stride_bytes,
                    int depth);

/* DispEst::CostConst_GPU (src/DispEst.cpp:272-276) == CVC_cl::buildCV: gray + x-gradient
 * of both images (CVC::preprocess arithmetic, src/CVC.cpp:41-46 - no +0.5) and both cost
 * volumes (CVC::buildCV_left/right arithmetic, src/CVC.cpp:122-179).  In the default float path all of it is lazy: the cost
 * volumes stay virtual (the fused filter builds the costs on the fly) and - since round 6 - so does the image preparation (the
 * guidance launch of psm_cost_filter forms gray and gradient itself); the call then only adopts the pair and resets the frame's
 * state.  Whatever reads the image planes or a volume earlier gets them prepared / materialised on demand. */
int psm_cost_construct(psm_ctx *ctx);

/* DispEst::CostFilter_GPU (src/DispEst.cpp:299-308) == CVF_cl::preprocess + filterCV for
 * the left then the right volume, with the arithmetic of CVF::preprocess and
 * GuidedFilter_cv (src/CVF.cpp:44-165).  Filters both volumes in place. */
int psm_cost_filter(psm_ctx *ctx);

/* One half of psm_cost_filter: preprocess + filter of volume `side` only (the reference runs the two
 * halves back to back, src/DispEst.cpp:302-305).  Lets a sharded host start exchanging the left
 * minima while the right volume is still being filtered. */
int psm_cost_filter_side(psm_ctx *ctx, int side);

/* DispEst::CostFilter_FGF (src/DispEst.cpp:281-296): the Fast Guided Filter variant of the aggregation,
 * FastGuidedFilterColor(I, GIF_R_WIN, GIF_EPS, subsample_rate) per slice (src/fastguidedfilter.cpp:124-209),
 * left then right volume, in place.  subsample_rate in {2, 4, 8} (the reference's default is 4,
 * src/DispEst.cpp:19); PSM_F32 contexts only; width/subsample_rate and height/subsample_rate must exceed
 * the blur radius 8/subsample_rate.  The reference has this on the CPU only ('m' mode has no FGF kernel). */
int psm_cost_filter_fgf(psm_ctx *ctx, int subsample_rate);

/* DispEst::DispSelect_GPU (src/DispEst.cpp:323-328) == DispSel_cl::CVSelect
 * (src/DispSel_cl.cpp:69-140) with DispSel::CVSelect arithmetic (src/DispSel.cpp:83-109).
 * lmap/rmap: H rows of W bytes, row pitch `stride` (cv::Mat lDisMap/rDisMap, CV_8UC1).
 * Either may be NULL: the maps then stay on the device (psm_download_maps).
 * Only valid on an unsharded context. */
int psm_disp_select(psm_ctx *ctx, uint8_t *lmap, uint8_t *rmap, size_t stride);

/* Sharded DispSel, step 1: local argmin over this context's slices with the global
 * semantics (d = 0 never a candidate, strict '<', lowest d wins ties).  Writes one packed
 * 64-bit key per pixel and side, keys[side][y][x] = (ordered(cost) << 32 | d) ^ (1<<63), so
 * that a signed 64-bit minimum over ranks selects (min cost, then lowest d).
 * dev_keys: DEVICE pointer to 2*H*W int64 (e.g. a torch tensor handed to RCCL), or NULL to
 * use the context's own buffer (psm_partial_keys). */
int psm_disp_select_partial(psm_ctx *ctx, void *dev_keys);
/* The same for one side only: dev_keys_side = DEVICE pointer to H*W int64 (NULL: the context's buffer). */
int psm_disp_select_partial_side(psm_ctx *ctx, int side, void *dev_keys_side);
/* Let the context write its packed minima (both sides, 2*H*W int64) straight into a caller-owned DEVICE buffer - e.g. the
 * torch tensor a collective is about to reduce - instead of its own (NULL: back to its own).  Call before
 * psm_cost_filter of the frame whose minima should land there; psm_disp_select_partial(ctx, NULL) and psm_disp_merge
 * then use that buffer.  Saves the device-to-device copy of the keys per frame on a sharded host. */
int psm_set_key_buffer(psm_ctx *ctx, void *dev_keys);
/* Device pointer / size of the context's current key buffer. */
int psm_partial_keys(psm_ctx *ctx, void **dev_keys, size_t *bytes);
/* Sharded DispSel, step 2: dev_keys_all = DEVICE pointer to nranks consecutive key buffers
 * (the all-gather result, rank-major).  Produces the two final maps as psm_disp_select. */
int psm_disp_merge(psm_ctx *ctx, const void *dev_keys_all, int nranks, uint8_t *lmap,
                   uint8_t *rmap, size_t stride);

/* Single-process form of the exchange step: `shards` are nshards contexts of one job (on the
 * same or on different devices) whose psm_disp_select_partial has run; their key planes are
 * copied (device copy; peer copy across devices; through page-locked host memory when the two
 * devices have no peer access) into `root`'s gather buffer and merged there.  This is
 * what a C++ host that drives several GPUs from one process uses instead of RCCL, and what
 * the single-GPU "logical shard" tests use. */
int psm_disp_merge_ctx(psm_ctx *root, psm_ctx *const *shards, int nshards, uint8_t *lmap,
                       uint8_t *rmap, size_t stride);

int psm_download_maps(psm_ctx *ctx, uint8_t *lmap, uint8_t *rmap, size_t stride);

/* ---- frame loop (src/main.cpp:64-73: one pair after the other; the reference's stage timers include the copies,
 * src/StereoMatch.cpp:227-237): the PCIe legs next to the kernels ----
 * psm_upload_pair_async: as psm_upload_pair, but for the NEXT frame: the images are copied to page-locked staging memory
 * before the call returns (the caller's buffers are free again) and travel on the context's copy stream into a second image
 * slot while the current frame is being computed; the next psm_cost_construct adopts that pair (its kernels wait for the
 * copy on the device) - so it is called right after psm_cost_construct of the current frame:
 *   psm_cost_construct(i); psm_upload_pair_async(pair i+1); psm_cost_filter(i); psm_disp_select(i, NULL, NULL, 0);
 *   psm_download_maps_async(); psm_download_maps_wait(...)   <- typically one frame later: the maps of frame i-1
 * psm_download_maps_async starts the D2H copy of the current maps (after the kernels that produce them, before any
 * later kernel overwrites them) and returns; psm_download_maps_wait blocks until they have arrived and hands them over
 * (lmap / rmap as psm_download_maps).  One upload and one download may be in flight. */
int psm_upload_pair_async(psm_ctx *ctx, const void *l, const void *r, int channels, size_t stride_bytes, int depth);
int psm_download_maps_async(psm_ctx *ctx);
int psm_download_maps_wait(psm_ctx *ctx, uint8_t *lmap, uint8_t *rmap, size_t stride);

/* "next" row: PP lrCheck on the device (src/PP.cpp:17-50) on the maps of the last
 * psm_disp_select/psm_disp_merge.  lvalid/rvalid: H x W bytes (0/1), pitch `stride`; either
 * may be NULL (results stay on the device). */
int psm_lr_check(psm_ctx *ctx, uint8_t *lvalid, uint8_t *rvalid, size_t stride);

/* "next" row: PP fillInv on the device (src/PP.cpp:52-143): every pixel the last psm_lr_check
 * marked invalid takes the smaller disparity of its nearest valid left/right neighbours in the row.
 * Modifies the device maps in place; lmap/rmap (optional) receive them. */
int psm_fill_invalid(psm_ctx *ctx, uint8_t *lmap, uint8_t *rmap, size_t stride);

/* "next" row: the plain weighted-median post-filter, wgtMedian (src/PP.cpp:145-247; constants MED_SZ 19, SIG_CLR 0.1,
 * SIG_DIS 9, include/PP.h:12-14), on the device maps of the last psm_disp_select/psm_disp_merge, for the pixels the
 * last psm_lr_check marked invalid (the sequence of PP::processDM: lrCheck, fillInv, wgtMedian, src/PP.cpp:405-410).
 * Left map with the left image and the squared distances (:169-175), right map with the right image and the
 * square-rooted ones (:216-224).  Same result as the reference's single-threaded form: the map is filtered in place in
 * raster order, a filtered pixel sees the filtered pixels before it.  Run as parallel sweeps to the fixed point of that
 * recursion (every sweep evaluates all pixels whose earlier window taps changed; it stops, at the reference's map, when
 * a sweep changes nothing); falls back to a row-dataflow pipeline with the reference's own dependency chain if 96
 * sweeps do not reach it (PSM_OPT_FLAGS 4194304: dataflow form only).
 * Needs W, H >= 9 (the reference's modulo wrap is undefined below that).  lmap/rmap (optional) receive the maps.
 * Device memory: from 8192 invalid pixels per map the 19 x 19 window weights of every invalid pixel are formed once and kept
 * for the sweeps - 1.5 KB per invalid pixel (0.6 GB per 1080p map at 20 % invalid), held by the context until psm_destroy;
 * above 12 GB (or half of the device's free memory) for the pair, or when the allocation fails, the evaluations form their
 * weights themselves (slower, same maps).
 * Always synchronises with the host, PSM_OPT_ASYNC or not: the number of sweeps depends on the data (the host reads the
 * device's per-sweep counters), so the call returns with the filtered maps complete. */
int psm_wgt_median(psm_ctx *ctx, uint8_t *lmap, uint8_t *rmap, size_t stride);
/* What the last psm_wgt_median did, per map {left, right}: sweeps until the fixed point (-1: dataflow form) and pixel
 * evaluations in total.  Either pointer may be NULL.  The maps are a function of the input alone (the unique fixed point of the
 * in-place recursion); these two numbers are not - a changed pixel is visible to evaluations still running in its sweep, so how
 * many evaluations (and, on dense maps, sweeps) a call needs can differ from run to run. */
int psm_wgt_median_stats(psm_ctx *ctx, int sweeps[2], long long evals[2]);

/* ---- second sharding axis: row stripes (SURVEY.md 8e asks for shards of the path; the filter's vertical support is
 * bounded - 8 rows of costs either side - so a stripe of output rows needs nothing from another stripe) ----
 * psm_set_rows restricts psm_cost_filter (select form) and psm_disp_select* of this context to the output rows
 * [y_begin, y_end) of the whole image - all D slices (or this context's slices) of both volumes, identical values to
 * the unrestricted run; the image pair is uploaded whole (borders reflect at the true image border).  Rows outside the
 * stripe of the maps / minima are undefined.  (0, 0) or (0, H): whole image again.  Takes effect with the next psm_cost_filter.
 * With G contexts / ranks on stripes [H*g/G, H*(g+1)/G) no minima are exchanged at all: the only exchange is the
 * gather of the finished map rows (2*W*H bytes in total) - psm_gather_rows_ctx in one process, an all-gather of the
 * stripes between ranks (bench.py).  The post-processing entry points refuse stripe-only maps. */
int psm_set_rows(psm_ctx *ctx, int y_begin, int y_end);
/* The disparity maps [2][H][W] (uint8) of this context live in dev_maps (device memory of this context's GPU, at least
 * 2*W*H + 4 bytes) from now on; NULL: the context's own buffer again.  whole != 0: the buffer already holds both
 * complete maps of the current frame (the caller gathered the stripes into it). */
int psm_set_map_buffer(psm_ctx *ctx, void *dev_maps, int whole);
/* One process, several contexts (one per GPU or logical stripes on one GPU): copies the stripe rows of every context's
 * maps into root's maps (root may be one of them); checks that the stripes tile [0, H) and that every context has run
 * psm_disp_select for the frame.  lmap/rmap (optional) receive the whole maps. */
int psm_gather_rows_ctx(psm_ctx *root, psm_ctx *const *stripes, int nstripes, uint8_t *lmap, uint8_t *rmap, size_t stride);
/* How many legs of psm_gather_rows_ctx / psm_disp_merge_ctx with this root went through host memory so far (devices without
 * peer access, or PSM_OPT_GATHER_STAGED) instead of a device / peer copy. */
int psm_gather_staged_legs(const psm_ctx *root);

/* ---- several pairs per launch: the reference's use on Middlebury-size data is a loop over pairs / datasets
 * (src/main.cpp:64-73, src/StereoMatch.cpp:556-607) - one 450 x 375 x 64 pair is 1.7 rounds of the chip's resident workgroups
 * behind four launches at their latency floor ----
 * psm_compute_batch runs DispEst::CostConst_GPU + CostFilter_GPU + DispSelect_GPU (src/DispEst.cpp:272-276,299-308,323-328) for
 * the n contexts ctxs[0..n) - same width, height, max_disp, slice range, dtype, device and options, each holding its own pair
 * (psm_upload_pair / psm_upload_pair_async) - in SHARED launches: one guidance kernel (which also prepares the images; 8-bit mode: a preparation launch before it), one fused
 * CVC + CVF + WTA grid over all pairs (two in the two-phase form), one reduction.  Afterwards every context is exactly where
 * psm_cost_construct + psm_cost_filter + psm_disp_select(ctx, NULL, NULL, 0) would have left it - same maps bit for bit
 * (psm_download_maps, psm_download_maps_async), same packed minima on a disparity shard, post-processing and volume readers
 * as usual.  The launches run on ctxs[0]'s stream, ordered after everything already queued on the other contexts' streams
 * and before anything queued on them later; synchronous on return unless ctxs[0] has PSM_OPT_ASYNC.  Default select path
 * only: contexts with a row stripe, the storing flags or the direct kernel variant are refused.  Stage timers of every context
 * receive the batch's wall time. */
int psm_compute_batch(psm_ctx *const *ctxs, int n);
/* The contexts of a batch (one device) run on ONE compute stream and one copy stream each way from now on, instead of three
 * streams per context - what a frame loop over batches wants: the runtime multiplexes streams onto a few hardware queues, and
 * with 8 contexts' 24 streams every asynchronous copy cost 0.2 ms of HOST time (measured).  Call once, before the first frame;
 * the streams live until the last of the contexts is destroyed.  psm_set_stream is refused afterwards. */
int psm_share_streams(psm_ctx *const *ctxs, int n);

/* ---- debug / bench entry points (no counterpart in the reference) ---- */
/* Replace the device maps and validity masks (any may be NULL = keep) - lets the post-processing stages run on maps
 * that did not come from this context's WTA.  H rows of W bytes, pitch `stride`; map values must be < max_disp. */
int psm_upload_maps(psm_ctx *ctx, const uint8_t *lmap, const uint8_t *rmap, const uint8_t *lvalid, const uint8_t *rvalid,
                    size_t stride);

/* Copy slices [d0,d1) (global disparity numbers) of a volume to/from dense host memory
 * [d1-d0][H][W]; element type = the context's dtype. */
int psm_download_volume(psm_ctx *ctx, int side, int d0, int d1, void *host);
/* Uploaded float costs may have any scale: slices whose non-zero magnitudes leave 2^-60 .. 2^60 (measured on the device) make
 * the following psm_cost_filter run its storing form (see PSM_IMG_F32 above) - same results as the reference arithmetic at any
 * scale, never a silent difference between the two forms. */
int psm_upload_volume(psm_ctx *ctx, int side, int d0, int d1, const void *host);
/* After psm_cost_filter: the a0,a1,a2,b intermediates of the LAST filtered side (right) are
 * still in the scratch buffer; psm_filter_stage_a(side) runs only the first half of the
 * guided filter for `side` and leaves them there.  host: [d1-d0][H][W][4] floats. */
int psm_filter_stage_a(psm_ctx *ctx, int side);
int psm_download_ab(psm_ctx *ctx, int d0, int d1, float *host);
/* guidance planes of `side` after psm_cost_filter/psm_filter_stage_a: host receives
 * [14][H][W] floats: I0,I1,I2,grad, mI0,mI1,mI2,invDET, A00,A01,A02,A11,A12,A22.  (PSM_U8 contexts: `grad` is the 8-bit
 * gradient of assets/cvc.cl's preprocessing as a float - the float x-gradient is not kept in that mode.) */
int psm_download_guidance(psm_ctx *ctx, int side, float *host);
/* The north-star kernel in isolation: cv::boxFilter(Size(8,8)) semantics applied to every
 * slice of volume `side`, result left in the scratch buffer; host (optional) receives
 * [Dlocal][H][W] floats. */
int psm_box8_volume(psm_ctx *ctx, int side, float *host);

/* Wall time of the last call of each stage in microseconds (the reference's
 * cvc_time/cvf_time/dispsel_time, src/StereoMatch.cpp:227-241). */
int psm_stage_time_us(psm_ctx *ctx, int stage, double *us);
/* With PSM_OPT_PROFILE=1: accumulated device time (hipEvent pairs on the launch stream)
 * and launch count of a kernel class since the last psm_reset_kernel_times(). */
int psm_kernel_time_ms(psm_ctx *ctx, int kernel, double *total_ms, int *launches);
int psm_reset_kernel_times(psm_ctx *ctx);
/* With PSM_OPT_PROFILE=2: duration in ms (first workgroup start to last workgroup end, device constant-rate clock) and form
 * (1 = minima planes, 2 = key plane, 0 = storing) of every launch of the fused filter kernel since the last call, in launch
 * order; at most max_launches (the library keeps 4096).  Costs two 64-bit atomics per workgroup and no events or
 * synchronisation between kernels, so it can stay on inside a timed region.  Resets the record. */
int psm_filter_launch_times(psm_ctx *ctx, double *ms, int *form, int max_launches, int *n_launches);

/* geometry queries */
int psm_get_info(const psm_ctx *ctx, int *width, int *height, int *max_disp, int *d_begin,
                 int *d_end, int *dtype, int *device);

#ifdef __cplusplus
}
#endif
#endif
