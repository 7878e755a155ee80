// psm_demo - headless counterpart of the reference's StereoMatch::compute accelerator branch
// (src/StereoMatch.cpp:193-262): raw B,G,R uint8 pair in, four timed stages, raw uint8 maps out.
//   psm_demo <left.raw> <right.raw> <W> <H> <maxDis> <out_prefix> [ndev] [f32|u8] [float_input] [fgf_rate] [pp] [frames] [batch] [ring]
// ring > 0: additionally push that many frames of the pair through a psm::FrameRing of two objects (two frames in flight, each
// object told PSM_OPT_FRAMES_IN_FLIGHT = 2), check every delivered frame's maps against the single-pair run, dump <out>_ldisp_ring.raw
// batch > 1: additionally run that many copies of the pair as ONE batch (DispEst::computeBatch -> psm_compute_batch: the
// reference's loop over pairs as shared launches), check every copy's maps against the single-pair run, dump <out>_ldisp_batch.raw
// frames > 0: additionally run that many frames of the pair through DispEst::computeFrame (asynchronous upload of the next
// pair / download of the previous maps: the frame loop of src/main.cpp:64-73), dump its last maps as <out>_ldisp_loop.raw and
// print the time per frame
// fgf_rate 0 (default): CostFilter_GPU; 2/4/8: CostFilter_FGF_GPU with that subsample rate
// pp 1: after the left-right check also fillInv + wgtMedian (the stages of PP::processDM, src/PP.cpp:405-410); the
//       post-processed maps go to <out_prefix>_ldisp_pp.raw / _rdisp_pp.raw
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "DispEst.h"

static bool slurp(const char *path, std::vector<unsigned char> &buf, size_t n)
{
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    buf.resize(n);
    size_t got = fread(buf.data(), 1, n, f);
    fclose(f);
    return got == n;
}
static bool dump(const std::string &path, const unsigned char *p, size_t n)
{
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return false;
    size_t put = fwrite(p, 1, n, f);
    fclose(f);
    return put == n;
}

int main(int argc, char **argv)
{
    if (argc < 7) {
        fprintf(stderr, "usage: %s left.raw right.raw W H maxDis out_prefix [ndev] [f32|u8] [float_input] [fgf_rate] [pp] [frames] [batch] [ring]\n", argv[0]);
        return 2;
    }
    const int W = atoi(argv[3]), H = atoi(argv[4]), D = atoi(argv[5]);
    const std::string out = argv[6];
    const int ndev = argc > 7 ? atoi(argv[7]) : 1;
    const int dtype = (argc > 8 && !strcmp(argv[8], "u8")) ? PSM_U8 : PSM_F32;
    const bool float_input = argc > 9 && atoi(argv[9]) != 0;
    const int fgf_rate = argc > 10 ? atoi(argv[10]) : 0;
    const bool pp = argc > 11 && atoi(argv[11]) != 0;
    const int frames = argc > 12 ? atoi(argv[12]) : 0;
    const int batch = argc > 13 ? atoi(argv[13]) : 0;
    const int nring = argc > 14 ? atoi(argv[14]) : 0;
    std::vector<unsigned char> lraw, rraw;
    if (!slurp(argv[1], lraw, (size_t)W * H * 3) || !slurp(argv[2], rraw, (size_t)W * H * 3)) {
        fprintf(stderr, "psm_demo: cannot read the input pair\n");
        return 2;
    }
    int gotDev = psm::hipUtil::hipDevicePoll();  // src/main.cpp:29 openCLdevicepoll()
    if (gotDev <= 0) {
        fprintf(stderr, "psm_demo: no HIP device / library (%s)\n", psm::hipUtil::error().c_str());
        return 3;
    }
    psm::Mat l(H, W, 3, psm::PSM_8U, lraw.data()), r(H, W, 3, psm::PSM_8U, rraw.data());
    std::vector<float> lf, rf;
    if (float_input) {  // src/StereoMatch.cpp:195-196 convertTo(CV_32F, 1/255.0f)
        lf.resize(lraw.size());
        rf.resize(rraw.size());
        const float alpha = 1 / 255.0f;
        for (size_t i = 0; i < lraw.size(); ++i) {
            lf[i] = (float)lraw[i] * alpha;
            rf[i] = (float)rraw[i] * alpha;
        }
        l = psm::Mat(H, W, 3, psm::PSM_32F, lf.data());
        r = psm::Mat(H, W, 3, psm::PSM_32F, rf.data());
    }
    psm::DispEst SMDE(l, r, D, 8, gotDev > 0, ndev, dtype);
    if (!SMDE.ok()) return 4;
    SMDE.setInputImages(l, r);
    SMDE.setThreads(8);
    SMDE.setSubsampleRate(fgf_rate ? fgf_rate : 4);
    int rc = 0;
    rc |= SMDE.CostConst_GPU();
    rc |= fgf_rate ? SMDE.CostFilter_FGF_GPU() : SMDE.CostFilter_GPU();
    rc |= SMDE.DispSelect_GPU();
    rc |= SMDE.PostProcess_GPU();        // (L-R validity only: the maps stay the raw WTA maps; ProcessDM_GPU below is the commented-out sequence of processDM)
    if (rc) return 5;
    printf("STEREO GIF Module Times:\nCVC Time:\t %4.2f ms\nCVF Time:\t %4.2f ms\nDispSel Time:\t %4.2f ms\nPP Time:\t %4.2f ms\n",
           SMDE.stageTimeUs(PSM_STAGE_CVC) / 1000, SMDE.stageTimeUs(PSM_STAGE_CVF) / 1000,
           SMDE.stageTimeUs(PSM_STAGE_DISPSEL) / 1000, SMDE.stageTimeUs(PSM_STAGE_PP) / 1000);
    bool ok = dump(out + "_ldisp.raw", SMDE.lDisMap.data, (size_t)W * H) && dump(out + "_rdisp.raw", SMDE.rDisMap.data, (size_t)W * H) &&
              dump(out + "_lvalid.raw", SMDE.lValid.data, (size_t)W * H) && dump(out + "_rvalid.raw", SMDE.rValid.data, (size_t)W * H);
    if (ok && pp) {
        if (SMDE.ProcessDM_GPU()) return 5;        // lrCheck + fillInv + wgtMedian (src/PP.cpp:405-410, commented out in the reference)
        ok = dump(out + "_ldisp_pp.raw", SMDE.lDisMap.data, (size_t)W * H) && dump(out + "_rdisp_pp.raw", SMDE.rDisMap.data, (size_t)W * H);
    }
    if (ok && frames > 0 && ndev == 1 && !fgf_rate) {
        SMDE.setInputImages(l, r);
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < frames; ++i)
            if (SMDE.computeFrame(i + 1 < frames ? &l : nullptr, i + 1 < frames ? &r : nullptr, i > 0)) return 5;
        if (SMDE.finishFrames()) return 5;
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        printf("Frame loop:	 %d frames, %4.3f ms per frame (H2D of every pair and D2H of every pair of maps included)\n", frames, ms / frames);
        ok = dump(out + "_ldisp_loop.raw", SMDE.lDisMap.data, (size_t)W * H) && dump(out + "_rdisp_loop.raw", SMDE.rDisMap.data, (size_t)W * H);
    }
    if (ok && batch > 1 && ndev == 1 && !fgf_rate) {
        std::vector<std::vector<uint8_t>> keep(2);
        keep[0].assign(SMDE.lDisMap.data, SMDE.lDisMap.data + (size_t)W * H);      // (frames > 0: the loop's maps = the same pair's)
        keep[1].assign(SMDE.rDisMap.data, SMDE.rDisMap.data + (size_t)W * H);
        std::vector<psm::DispEst *> des;
        for (int b = 0; b < batch; ++b) des.push_back(new psm::DispEst(l, r, D, 8, true, 1, dtype));
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = psm::DispEst::computeBatch(des.data(), batch);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        bool same = rc == 0;
        for (int b = 0; b < batch && same && !pp; ++b)
            same = !memcmp(des[b]->lDisMap.data, keep[0].data(), (size_t)W * H) && !memcmp(des[b]->rDisMap.data, keep[1].data(), (size_t)W * H);
        printf("Batch:\t %d pairs in one set of launches, %4.3f ms (first call, maps downloaded), maps %s\n", batch, ms, same ? "equal the single-pair run's" : "DIFFER");
        ok = same && dump(out + "_ldisp_batch.raw", des[batch - 1]->lDisMap.data, (size_t)W * H) && dump(out + "_rdisp_batch.raw", des[batch - 1]->rDisMap.data, (size_t)W * H);
        for (auto *d : des) delete d;
    }
    if (ok && nring > 0 && ndev == 1 && !fgf_rate && !pp) {
        std::vector<uint8_t> kl(SMDE.lDisMap.data, SMDE.lDisMap.data + (size_t)W * H), kr(SMDE.rDisMap.data, SMDE.rDisMap.data + (size_t)W * H);
        psm::FrameRing ring(l, r, D, 2, dtype);
        if (!ring.ok()) return 5;
        psm::Mat ol = psm::Mat::zeros(H, W, 1, psm::PSM_8U), orr = psm::Mat::zeros(H, W, 1, psm::PSM_8U);
        int delivered = 0;
        bool same = true;
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < nring + 2 && same; ++i) {
            const int got = i < nring ? ring.push(l, r, &ol, &orr) : ring.flush(&ol, &orr);
            if (got < 0) return 5;
            if (got == 1) {
                ++delivered;
                same = !memcmp(ol.data, kl.data(), (size_t)W * H) && !memcmp(orr.data, kr.data(), (size_t)W * H);
            }
        }
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        same = same && delivered == nring && ring.flush(&ol, &orr) == 0;
        printf("Frame ring:\t %d frames through 2 objects, %4.3f ms per frame (H2D and D2H of every frame included), %d delivered, maps %s\n", nring,
               ms / nring, delivered, same ? "equal the single-pair run's" : "DIFFER");
        ok = same && dump(out + "_ldisp_ring.raw", ol.data, (size_t)W * H) && dump(out + "_rdisp_ring.raw", orr.data, (size_t)W * H);
    }
    return ok ? 0 : 6;
}
