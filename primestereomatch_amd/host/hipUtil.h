// hipUtil.h - run-time loader of libprimesm_hip.so.  Takes the place of the reference's oclUtil
// (include/oclUtil.h:47-152, src/oclUtil.cpp): where oclUtil polls OpenCL devices, creates the
// context/queue and JIT-builds the .cl programs, hipUtil dlopen()s the prebuilt HIP library and
// binds the C ABI of include/primesm_hip.h, so the host program has no link-time HIP dependency.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>

#include "../../include/primesm_hip.h"

namespace psm {

struct HipApi {
    // one pointer per entry point of include/primesm_hip.h that the DispEst mirror uses
    int (*device_count)(void) = nullptr;
    int (*create_shard)(psm_ctx **, int, int, int, int, int, int, int) = nullptr;
    void (*destroy)(psm_ctx *) = nullptr;
    const char *(*last_error)(const psm_ctx *) = nullptr;
    int (*set_option)(psm_ctx *, int, int) = nullptr;
    int (*upload_pair)(psm_ctx *, const void *, const void *, int, size_t, int) = nullptr;
    int (*upload_pair_async)(psm_ctx *, const void *, const void *, int, size_t, int) = nullptr;
    int (*download_maps_async)(psm_ctx *) = nullptr;
    int (*download_maps_wait)(psm_ctx *, uint8_t *, uint8_t *, size_t) = nullptr;
    int (*cost_construct)(psm_ctx *) = nullptr;
    int (*cost_filter)(psm_ctx *) = nullptr;
    int (*cost_filter_fgf)(psm_ctx *, int) = nullptr;
    int (*disp_select)(psm_ctx *, uint8_t *, uint8_t *, size_t) = nullptr;
    int (*disp_select_partial)(psm_ctx *, void *) = nullptr;
    int (*disp_merge_ctx)(psm_ctx *, psm_ctx *const *, int, uint8_t *, uint8_t *, size_t) = nullptr;
    int (*set_rows)(psm_ctx *, int, int) = nullptr;
    int (*gather_rows_ctx)(psm_ctx *, psm_ctx *const *, int, uint8_t *, uint8_t *, size_t) = nullptr;
    int (*lr_check)(psm_ctx *, uint8_t *, uint8_t *, size_t) = nullptr;
    int (*fill_invalid)(psm_ctx *, uint8_t *, uint8_t *, size_t) = nullptr;
    int (*wgt_median)(psm_ctx *, uint8_t *, uint8_t *, size_t) = nullptr;
    int (*stage_time_us)(psm_ctx *, int, double *) = nullptr;
    int (*compute_batch)(psm_ctx *const *, int) = nullptr;
    int (*download_maps)(psm_ctx *, uint8_t *, uint8_t *, size_t) = nullptr;
};

class hipUtil {
public:
    // Loads the library once per process.  path == nullptr: $PRIMESM_HIP_LIB, then
    // libprimesm_hip.so next to the executable's ../lib, then the dynamic linker's search path.
    // Returns false (and keeps error()) if the library or a symbol is missing.
    static bool load(const char *path = nullptr);
    static bool loaded();
    static const HipApi &api();
    static const std::string &error();
    // Replaces openCLdevicepoll() (src/oclUtil.cpp:18-135): number of usable devices, 0 if the
    // library cannot be loaded or no GPU is present.
    static int hipDevicePoll();
};

}  // namespace psm
