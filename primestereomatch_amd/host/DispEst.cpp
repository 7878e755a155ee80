#include "DispEst.h"

#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <cstring>

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
    // PSM_HOST_LOGICAL_STRIPES=1: more stripes than devices are allowed, the contexts then share devices (tests on one GPU)
    const char *logical = std::getenv("PSM_HOST_LOGICAL_STRIPES");
    const bool share = logical && *logical && *logical != '0' && have > 0;
    if (ndev > have && !share) ndev = have;
    if (ndev > hei) ndev = hei;
    // several devices: one context per device on a stripe of the image rows - all disparities of both volumes, so the WTA
    // finishes on the device and only finished map rows are gathered (psm_gather_rows_ctx).  Stripes of R = ceil(H / ndev)
    // rows, aligned at multiples of R (the same bounds as the rank-per-GPU host, primestereomatch_amd/stripes.py: a gathered
    // [ndev][2][R][W] buffer is the image); devices past the last non-empty stripe stay unused.
    const int R = (hei + ndev - 1) / ndev;
    while (ndev > 1 && (ndev - 1) * R >= hei) --ndev;
    for (int g = 0; g < ndev; ++g) {
        psm_ctx *c = nullptr;
        const int ya = g * R, yb = ya + R < hei ? ya + R : hei;
        y0s.push_back(ya);
        y1s.push_back(yb);
        if (api.create_shard(&c, wid, hei, maxDis, 0, maxDis, dtype, g % have) != 0 ||
            (ndev > 1 && api.set_rows(c, ya, yb) != 0)) {
            fprintf(stderr, "DispEst: %s\n", api.last_error(c));
            if (c) api.destroy(c);
            for (psm_ctx *p : ctx) api.destroy(p);
            ctx.clear();
            useOCL = false;
            return;
        }
        if (ndev > 1) api.set_option(c, PSM_OPT_ASYNC, 1);  // the stripes run concurrently
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
    if (whole_on_first && ctx.size() > 1) rc |= hipUtil::api().set_rows(ctx[0], y0s[0], y1s[0]);   // back to stripes
    whole_on_first = false;
    for (psm_ctx *c : ctx) rc |= hipUtil::api().cost_filter(c);
    return rc;
}

int DispEst::CostFilter_FGF_GPU()
{
    if (ctx.empty()) return 1;
    // the Fast Guided Filter path has no row stripes (its low-resolution models would need their own halo arithmetic, and
    // it is 3x cheaper than the full filter anyway): the first device filters the whole image, the others sit this frame out
    int rc = 0;
    if (ctx.size() > 1) rc |= hipUtil::api().set_rows(ctx[0], 0, 0);
    whole_on_first = ctx.size() > 1;
    rc |= hipUtil::api().cost_filter_fgf(ctx[0], (int)subsample_rate);
    return rc;
}

int DispEst::DispSelect_GPU()
{
    if (ctx.empty()) return 1;
    const HipApi &api = hipUtil::api();
    if (ctx.size() == 1 || whole_on_first) return api.disp_select(ctx[0], lDisMap.data, rDisMap.data, lDisMap.step);
    int rc = 0;
    for (psm_ctx *c : ctx) rc |= api.disp_select(c, nullptr, nullptr, 0);
    rc |= api.gather_rows_ctx(ctx[0], ctx.data(), (int)ctx.size(), lDisMap.data, rDisMap.data, lDisMap.step);
    return rc;
}

int DispEst::setOption(int option, int value)
{
    if (ctx.empty()) return 1;
    int rc = 0;
    for (psm_ctx *c : ctx) rc |= hipUtil::api().set_option(c, option, value);
    return rc;
}

int DispEst::LRCheck_GPU()
{
    if (ctx.empty()) return 1;
    return hipUtil::api().lr_check(ctx[0], lValid.data, rValid.data, lValid.step);
}

int DispEst::PostProcess_GPU()
{   // The reference's call (src/DispEst.cpp:338-344) runs PP::processDM, whose LIVE body is the CPU JointWMF (out of scope) with
    // lrCheck / fillInv / wgtMedian commented out (src/PP.cpp:405-412).  What the accelerator side contributes to this stage is
    // the L-R check: lValid / rValid are filled, lDisMap / rDisMap stay the raw WTA maps (a caller that reads them afterwards
    // gets what DispSelect_GPU left).  The commented-out sequence as a whole: ProcessDM_GPU().
    return LRCheck_GPU();
}

int DispEst::ProcessDM_GPU()
{   // the sequence PP::processDM's source spells out but does not run (src/PP.cpp:405-410): lrCheck, fillInv, wgtMedian - all on
    // the device; lDisMap / rDisMap are REPLACED by the filled, weighted-median-filtered maps
    if (ctx.empty()) return 1;
    const HipApi &api = hipUtil::api();
    int rc = api.lr_check(ctx[0], lValid.data, rValid.data, lValid.step);
    if (!rc) rc = api.fill_invalid(ctx[0], nullptr, nullptr, 0);
    if (!rc) rc = api.wgt_median(ctx[0], lDisMap.data, rDisMap.data, lDisMap.step);
    return rc;
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

int DispEst::computeFrame(const Mat *nextL, const Mat *nextR, bool have_prev)
{
    if (ctx.size() != 1) return 1;           // (a multi-device host gathers stripes: use the stage calls)
    const HipApi &api = hipUtil::api();
    psm_ctx *c = ctx[0];
    int rc = api.cost_construct(c);
    if (nextL && nextR)
        rc |= api.upload_pair_async(c, nextL->data, nextR->data, nextL->channels, nextL->step,
                                    nextL->depth == PSM_32F ? PSM_IMG_F32 : PSM_IMG_U8);
    rc |= api.cost_filter(c);
    rc |= api.disp_select(c, nullptr, nullptr, 0);
    if (have_prev) rc |= api.download_maps_wait(c, lDisMap.data, rDisMap.data, lDisMap.step);
    rc |= api.download_maps_async(c);
    return rc;
}

int DispEst::finishFrames()
{
    if (ctx.size() != 1) return 1;
    return hipUtil::api().download_maps_wait(ctx[0], lDisMap.data, rDisMap.data, lDisMap.step);
}

int DispEst::computeBatch(DispEst *const *des, int n)
{
    if (!des || n < 1) return 1;
    std::vector<psm_ctx *> cs;
    for (int i = 0; i < n; ++i) {
        if (!des[i] || des[i]->ctx.size() != 1) return 1;      // (a multi-device object shards ONE pair; batches are per device)
        cs.push_back(des[i]->ctx[0]);
    }
    const HipApi &api = hipUtil::api();
    int rc = api.compute_batch(cs.data(), n);
    for (int i = 0; i < n && !rc; ++i)
        rc |= api.download_maps(cs[i], des[i]->lDisMap.data, des[i]->rDisMap.data, des[i]->lDisMap.step);
    return rc;
}

double DispEst::stageTimeUs(int stage) const
{
    double us = 0;
    if (!ctx.empty()) hipUtil::api().stage_time_us(ctx[0], stage, &us);
    return us;
}

// ---- FrameRing ----
FrameRing::FrameRing(Mat l, Mat r, int d, int frames, int dtype)
{
    if (frames < 1) return;
    for (int i = 0; i < frames; ++i) {
        DispEst *de = new DispEst(l, r, d, MAX_CPU_THREADS, true, 1, dtype);
        if (!de->ok() || de->setOption(PSM_OPT_ASYNC, 1) || de->setOption(PSM_OPT_FRAMES_IN_FLIGHT, frames)) {
            delete de;
            for (DispEst *o : ring) delete o;
            ring.clear();
            return;
        }
        ring.push_back(de);
    }
    busy.assign(frames, 0);
}

FrameRing::~FrameRing()
{
    for (DispEst *o : ring) delete o;
}

int FrameRing::setOption(int option, int value)
{
    int rc = 0;
    for (DispEst *o : ring) rc |= o->setOption(option, value);
    return rc;
}

int FrameRing::deliver(int i, Mat *outL, Mat *outR)
{   // the maps of object i's last frame have been on their way since its push: wait for them, hand them over
    DispEst *de = ring[i];
    if (hipUtil::api().download_maps_wait(de->ctx[0], de->lDisMap.data, de->rDisMap.data, de->lDisMap.step)) return -1;
    busy[i] = 0;
    const size_t row = (size_t)de->wid;
    for (int y = 0; y < de->hei; ++y) {
        if (outL && outL->data) memcpy(outL->ptr<unsigned char>(y), de->lDisMap.ptr<unsigned char>(y), row);
        if (outR && outR->data) memcpy(outR->ptr<unsigned char>(y), de->rDisMap.ptr<unsigned char>(y), row);
    }
    return 1;
}

int FrameRing::push(const Mat &l, const Mat &r, Mat *outL, Mat *outR)
{
    if (ring.empty()) return -1;
    const HipApi &api = hipUtil::api();
    const int i = (int)(pushed % (long long)ring.size());
    int got = 0;
    if (busy[i]) {
        got = deliver(i, outL, outR);
        if (got < 0) return got;
        ++flushed;
    }
    DispEst *de = ring[i];
    psm_ctx *c = de->ctx[0];
    if (api.upload_pair_async(c, l.data, r.data, l.channels, l.step, l.depth == PSM_32F ? PSM_IMG_F32 : PSM_IMG_U8)) return -1;
    if (api.cost_construct(c) || api.cost_filter(c) || api.disp_select(c, nullptr, nullptr, 0) || api.download_maps_async(c)) return -1;
    busy[i] = 1;
    ++pushed;
    return got;
}

int FrameRing::flush(Mat *outL, Mat *outR)
{
    if (ring.empty()) return -1;
    while (flushed < pushed) {
        const int i = (int)(flushed % (long long)ring.size());
        ++flushed;
        if (busy[i]) return deliver(i, outL, outR);
    }
    return 0;
}

}  // namespace psm
