#include "hipUtil.h"

#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>

namespace psm {
namespace {
void *g_handle = nullptr;
HipApi g_api;
std::string g_error;

template <typename F>
bool bind(F &fn, const char *name)
{
    fn = reinterpret_cast<F>(dlsym(g_handle, name));
    if (!fn) {
        g_error = std::string("hipUtil: symbol ") + name + " missing from libprimesm_hip.so";
        return false;
    }
    return true;
}
}  // namespace

bool hipUtil::load(const char *path)
{
    if (g_handle) return true;
    const char *cands[3] = {path, std::getenv("PRIMESM_HIP_LIB"), "libprimesm_hip.so"};
    for (const char *c : cands) {
        if (!c || !*c) continue;
        g_handle = dlopen(c, RTLD_NOW | RTLD_LOCAL);
        if (g_handle) break;
        g_error = std::string("hipUtil: dlopen(") + c + ") failed: " + dlerror();
    }
    if (!g_handle) {
        fprintf(stderr, "%s\n", g_error.c_str());
        return false;
    }
    bool ok = bind(g_api.device_count, "psm_device_count") && bind(g_api.create_shard, "psm_create_shard") &&
              bind(g_api.destroy, "psm_destroy") && bind(g_api.last_error, "psm_last_error") &&
              bind(g_api.set_option, "psm_set_option") && bind(g_api.upload_pair, "psm_upload_pair") &&
              bind(g_api.upload_pair_async, "psm_upload_pair_async") && bind(g_api.download_maps_async, "psm_download_maps_async") &&
              bind(g_api.download_maps_wait, "psm_download_maps_wait") &&
              bind(g_api.cost_construct, "psm_cost_construct") && bind(g_api.cost_filter, "psm_cost_filter") &&
              bind(g_api.cost_filter_fgf, "psm_cost_filter_fgf") &&
              bind(g_api.disp_select, "psm_disp_select") && bind(g_api.disp_select_partial, "psm_disp_select_partial") &&
              bind(g_api.disp_merge_ctx, "psm_disp_merge_ctx") && bind(g_api.set_rows, "psm_set_rows") && bind(g_api.gather_rows_ctx, "psm_gather_rows_ctx") &&
              bind(g_api.lr_check, "psm_lr_check") && bind(g_api.fill_invalid, "psm_fill_invalid") &&
              bind(g_api.wgt_median, "psm_wgt_median") &&
              bind(g_api.stage_time_us, "psm_stage_time_us") && bind(g_api.compute_batch, "psm_compute_batch") &&
              bind(g_api.download_maps, "psm_download_maps");
    if (!ok) {
        fprintf(stderr, "%s\n", g_error.c_str());
        dlclose(g_handle);
        g_handle = nullptr;
    }
    return ok;
}

bool hipUtil::loaded() { return g_handle != nullptr; }
const HipApi &hipUtil::api() { return g_api; }
const std::string &hipUtil::error() { return g_error; }

int hipUtil::hipDevicePoll()
{
    if (!load()) return 0;
    return g_api.device_count();
}

}  // namespace psm
