"""Headless counterpart of StereoMatch::compute's accelerator branch (src/StereoMatch.cpp:193-311):
set the pair, run the four timed stages through the DispEst mirror, scale the maps for display and
score the left map against ground truth exactly as the reference does.  ("next" row 2 of SURVEY.md 8f.)
"""
from __future__ import annotations

import numpy as np

from . import capi
from .dispest import DispEst

MASK_NONE, MASK_NONOCC, MASK_DISC = 0, 1, 2   # include/StereoMatch.h


def error_vs_ground_truth(lDisMap, gt, mask, maxDis, scale_factor, error_threshold=4):
    """src/StereoMatch.cpp:248,275-309.  Returns (%BP, avg_err, num_bad_pixels, error map)."""
    # SMDE->lDisMap.convertTo(lDispMap, CV_8U, scale_factor)
    lDispMap = np.clip(lDisMap.astype(np.int32) * int(scale_factor), 0, 255).astype(np.uint8)
    e = np.abs(lDispMap.astype(np.int32) - np.asarray(gt, np.uint8).astype(np.int32))   # cv::absdiff
    e[:, :maxDis + 1] = 0                                    # eDispMap(Rect(0,0,maxDis+1,rows)) = 0
    unit = 127 // maxDis                                     # CHAR_MAX/maxDis, integer division
    e[e <= error_threshold * unit] = 0                       # THRESH_TOZERO
    if mask is not None:                                     # eDispMap.mul(errMask, 1/255.f)
        m = np.asarray(mask, np.uint8).astype(np.float64)
        e = np.rint(e * m * float(np.float32(1 / 255.0))).astype(np.int32)
    e = np.clip(e, 0, 255).astype(np.uint8)
    avg_err = float(e.mean()) / unit if unit else 0.0
    bad = int(np.count_nonzero(e))
    return 100.0 * bad / e.size, avg_err, bad, e


def compute(l_bgr, r_bgr, maxDis=64, gt=None, mask=None, scale_factor=4, error_threshold=4, threads=8,
            dtype="f32", post_process=True, verbose=False, subsample_rate=0, process_dm=False):
    """One frame of STEREO_GIF on the accelerator path.  l_bgr/r_bgr: H x W x 3 uint8 (imread order).
    subsample_rate 0: full guided filter (CostFilter_GPU, the reference's 'm' branch); 2/4/8: the Fast Guided
    Filter variant (CostFilter_FGF, the snapshot's live branch, src/StereoMatch.cpp:213) on the device.
    process_dm: after the L-R check run the rest of PP::processDM's plain sequence on the device - fillInv, then wgtMedian
    on the pixels the check rejected (src/PP.cpp:405-410; lrCheck and fillInv are commented out in the snapshot's live
    processDM, wgtMedian is its dead-code predecessor of JointWMF) - the maps before it stay in lDisMap_raw / rDisMap_raw."""
    out = {}
    lFrame = np.ascontiguousarray(l_bgr)
    rFrame = np.ascontiguousarray(r_bgr)
    if dtype == "f32":
        # lFrame.convertTo(lFrame, CV_32F, 1/255.0f) (src/StereoMatch.cpp:193-197)
        lFrame = lFrame.astype(np.float32) * np.float32(1 / 255.0)
        rFrame = rFrame.astype(np.float32) * np.float32(1 / 255.0)
    with DispEst(lFrame, rFrame, maxDis, threads, True, dtype=dtype) as SMDE:
        SMDE.setInputImages(lFrame, rFrame)
        SMDE.setThreads(threads)
        SMDE.setSubsampleRate(subsample_rate or 4)
        SMDE.CostConst_GPU()
        if subsample_rate:
            SMDE.CostFilter_FGF_GPU()
        else:
            SMDE.CostFilter_GPU()
        SMDE.DispSelect_GPU()
        if post_process or process_dm:
            SMDE.LRCheck_GPU()
            out["lValid"], out["rValid"] = SMDE.lValid.copy(), SMDE.rValid.copy()
        if process_dm:
            out["lDisMap_raw"], out["rDisMap_raw"] = SMDE.lDisMap.copy(), SMDE.rDisMap.copy()
            SMDE.FillInv_GPU()
            SMDE.WgtMedian_GPU()
        out["cvc_ms"] = SMDE.stage_time_us(capi.PSM_STAGE_CVC) / 1000
        out["cvf_ms"] = SMDE.stage_time_us(capi.PSM_STAGE_CVF) / 1000
        out["dispsel_ms"] = SMDE.stage_time_us(capi.PSM_STAGE_DISPSEL) / 1000
        out["pp_ms"] = SMDE.stage_time_us(capi.PSM_STAGE_PP) / 1000
        out["lDisMap"], out["rDisMap"] = SMDE.lDisMap.copy(), SMDE.rDisMap.copy()
    out["lDispMap"] = np.clip(out["lDisMap"].astype(np.int32) * scale_factor, 0, 255).astype(np.uint8)
    if gt is not None:
        bp, avg, bad, emap = error_vs_ground_truth(out["lDisMap"], gt, mask, maxDis, scale_factor, error_threshold)
        out.update({"bp_percent": bp, "avg_err": avg, "bad_pixels": bad})
    if verbose:
        print("STEREO GIF Module Times:")
        print("CVC Time:\t %4.2f ms" % out["cvc_ms"])
        print("CVF Time:\t %4.2f ms" % out["cvf_ms"])
        print("DispSel Time:\t %4.2f ms" % out["dispsel_ms"])
        print("PP Time:\t %4.2f ms" % out["pp_ms"])
        if gt is not None:
            print("%%BP = %.2f%% \t Avg Err = %.2f" % (out["bp_percent"], out["avg_err"]))
    return out
