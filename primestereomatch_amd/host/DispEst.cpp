#include "DispEst.h"

#include <cassert>
#include <cstdio>
#include <cstdlib>

namespace psm {

DispEst::DispEst(Mat l, Mat r, const int d, int t, bool ocl, int ndev, int dtype)
    : lImg(l), rImg(r), maxDis(d), threads(t), useOCL(ocl)
{
    hei = lImg.rows;
    wid = lImg.cols;
    if (lImg.type() != rImg.type()) {  // src/DispEst.cpp:21-29
        printf("DE: Error - Left & Right images are of different types.\n");
        exit(1);
    }
    lDisMap = Mat::zeros(hei, wid, 1, PSM_8U);
    rDisMap = Mat::zeros(hei, wid, 1, PSM_8U);
    lValid = Mat::zeros(hei, wid, 1, PSM_8U);
    rValid = Mat::zeros(hei, wid, 1, PSM_8U);
    if (!useOCL) return;
    if (!hipUtil::load()) {
        useOCL = false;
        return;
    }
    const HipApi &api = hipUtil::api();
    int have = api.device_count();
    if (ndev < 1) ndev = 1;
    if (ndev > have) ndev = have;
    if (ndev > maxDis) ndev = maxDis;
    for (int g = 0; g < ndev; ++g) {
        psm_ctx *c = nullptr;
        int d0 = (int)((long)maxDis * g / ndev), d1 = (int)((long)maxDis * (g + 1) / ndev);
        if (api.create_shard(&c, wid, hei, maxDis, d0, d1, dtype, g) != 0) {
            fprintf(stderr, "DispEst: %s\n", api.last_error(nullptr));
            for (psm_ctx *p : ctx) api.destroy(p);
            ctx.clear();
            useOCL = false;
            return;
        }
        if (ndev > 1) api.set_option(c, PSM_OPT_ASYNC, 1);  // shards run concurrently
        ctx.push_back(c);
    }
    setInputImages(lImg, rImg);
}

DispEst::~DispEst(void)
{
    if (hipUtil::loaded())
        for (psm_ctx *p : ctx) hipUtil::api().destroy(p);
}

int DispEst::setInputImages(Mat leftImg, Mat rightImg)
{
    assert(leftImg.type() == rightImg.type());  // src/DispEst.cpp:166
    lImg = leftImg;
    rImg = rightImg;
    int rc = 0;
    for (psm_ctx *c : ctx)
        rc |= hipUtil::api().upload_pair(c, lImg.data, rImg.data, lImg.channels, lImg.step,
                                         lImg.depth == PSM_32F ? PSM_IMG_F32 : PSM_IMG_U8);
    return rc;
}

int DispEst::setThreads(unsigned int newThreads)
{
    if (newThreads > MAX_CPU_THREADS) return -1;  // src/DispEst.cpp:172-179
    threads = newThreads;
    return 0;
}

int DispEst::CostConst_GPU()
{
    if (ctx.empty()) return 1;
    int rc = 0;
    for (psm_ctx *c : ctx) rc |= hipUtil::api().cost_construct(c);
    return rc;
}

int DispEst::CostFilter_GPU()
{
    if (ctx.empty()) return 1;
    int rc = 0;
    for (psm_ctx *c : ctx) rc |= hipUtil::api().cost_filter(c);
    return rc;
}

int DispEst::CostFilter_FGF_GPU()
{
    if (ctx.empty()) return 1;
    int rc = 0;
    for (psm_ctx *c : ctx) rc |= hipUtil::api().cost_filter_fgf(c, (int)subsample_rate);
    return rc;
}

int DispEst::DispSelect_GPU()
{
    if (ctx.empty()) return 1;
    const HipApi &api = hipUtil::api();
    if (ctx.size() == 1) return api.disp_select(ctx[0], lDisMap.data, rDisMap.data, lDisMap.step);
    int rc = 0;
    for (psm_ctx *c : ctx) rc |= api.disp_select_partial(c, nullptr);
    rc |= api.disp_merge_ctx(ctx[0], ctx.data(), (int)ctx.size(), lDisMap.data, rDisMap.data, lDisMap.step);
    return rc;
}

int DispEst::PostProcess_GPU()
{
    if (ctx.empty()) return 1;
    return hipUtil::api().lr_check(ctx[0], lValid.data, rValid.data, lValid.step);
}

int DispEst::FillInvalid_GPU()
{
    if (ctx.empty()) return 1;
    return hipUtil::api().fill_invalid(ctx[0], lDisMap.data, rDisMap.data, lDisMap.step);
}

int DispEst::WgtMedian_GPU()
{
    if (ctx.empty()) return 1;
    return hipUtil::api().wgt_median(ctx[0], lDisMap.data, rDisMap.data, lDisMap.step);
}

double DispEst::stageTimeUs(int stage) const
{
    double us = 0;
    if (!ctx.empty()) hipUtil::api().stage_time_us(ctx[0], stage, &us);
    return us;
}

}  // namespace psm
