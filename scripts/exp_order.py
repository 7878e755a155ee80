"""Experiment (round 5; experiment build: make -C primestereomatch_amd/csrc exp; PRIMESM_HIP_LIB=.../libprimesm_hip_exp.so):
does the order in which an XCD walks its (pair, chunk) items matter for SHORT launches (a 32-slice disparity shard)?
    PSM_PC_ORDER=K  XCDs own whole (column group, segment) pairs and walk them K pairs interleaved (1: pair-major)
                    (this knob and PSM_PC_SIDESX - both sides of a pair adjacent in one grid - lived in the kernel for the round-5
                    experiments only: git history; the product's planner knobs PSM_PC_DC / _SLOTS / _S / _SPREAD remain)
    PSM_PC_DC, PSM_PC_SLOTS, seg_rows: the planner's other choices
Prints ms per frame (best of 3 x 40 frames) per setting; maps / keys are compared with the default setting's."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import primestereomatch_amd as P
from primestereomatch_amd import capi, synth

W, H, D = 1920, 1080, 256
l, r, _ = synth.make_pair(W, H, D, seed=0)


def run(d0, d1, env, seg=0, flags=0, steps=40):
    for k in ("PSM_PC_ORDER", "PSM_PC_DC", "PSM_PC_SLOTS", "PSM_PC_S", "PSM_PC_SPREAD", "PSM_PC_SIDESX"):
        os.environ.pop(k, None)
    os.environ.update({k: str(v) for k, v in env.items()})
    with P.DispEst(l, r, D, 8, True, d_range=(d0, d1)) as de:
        de.set_option(capi.PSM_OPT_ASYNC, 1)
        if seg:
            de.set_option(capi.PSM_OPT_SEG_ROWS, seg)
        if flags:
            de.set_option(capi.PSM_OPT_FLAGS, flags)
        shard = d1 - d0 < D

        def step():
            de.CostConst_GPU(); de.CostFilter_GPU()
            de.DispSelect_partial() if shard else de.DispSelect_device()
        for _ in range(5):
            step()
        de.synchronize()
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            de.synchronize()
            best = min(best, 1e3 * (time.perf_counter() - t0) / steps)
        de.set_option(capi.PSM_OPT_PROFILE, 2)
        step(); de.synchronize(); de.filter_launch_times()
        step(); de.synchronize()
        lt = de.filter_launch_times()
        de.set_option(capi.PSM_OPT_ASYNC, 0)
        if shard:
            ptr, nbytes = de.partial_keys()
            import ctypes as C
            hip = C.CDLL("libamdhip64.so")
            buf = np.empty(nbytes // 8, np.int64)
            hip.hipMemcpy(buf.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), C.c_size_t(nbytes), 2)
            sig = buf
        else:
            sig = np.concatenate([m.ravel() for m in de.download_maps()])
    return best, [round(ms, 4) for ms, _ in lt], sig


if __name__ == "__main__":
    base = {}
    QUICK = len(sys.argv) > 1 and sys.argv[1] == "quick"
    for name, d0, d1, steps in (("disp8", 0, 32, 40), ("c4", 0, 256, 12)):
        settings = [({}, 0, 0)]
        if QUICK:
            settings += [({}, 0, capi.PSM_FLAG_TWO_PHASE_OFF)] if name == "c4" else [({"PSM_PC_DC": 4}, 0, 0), ({}, 0, capi.PSM_FLAG_TWO_PHASE_ON), ({"PSM_PC_S": 4}, 0, capi.PSM_FLAG_TWO_PHASE_ON)]
        elif name == "disp8":
            settings += [({"PSM_PC_ORDER": k}, 0, 0) for k in (1, 2, 3, 9)]
            settings += [({"PSM_PC_DC": dc}, 0, 0) for dc in (1, 4)]
            settings += [({"PSM_PC_DC": 1, "PSM_PC_ORDER": 1}, 0, 0)]
            settings += [({}, s, 0) for s in (120, 135, 180, 216, 270, 360, 540)]
            settings += [({"PSM_PC_SLOTS": s}, 0, 0) for s in (64, 128)]
            settings += [({"PSM_PC_S": s}, 0, capi.PSM_FLAG_TWO_PHASE_ON) for s in (4, 8, 16, 32)]
        else:
            settings += [({"PSM_PC_ORDER": k}, 0, 0) for k in (1, 2)]
        for env, seg, flags in settings:
            ms, lt, sig = run(d0, d1, env, seg, flags, steps)
            if name not in base:
                base[name] = sig
            print(f"{name}: env {env} seg {seg} flags {flags}: {ms:.4f} ms per frame; launches {lt}; same result: {bool(np.array_equal(sig, base[name]))}", flush=True)

