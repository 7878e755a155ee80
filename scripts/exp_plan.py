"""Experiment (round 5, experiment build): planner choices (DC, segment rows) of short launches, one and two frames in flight.
    python scripts/exp_plan.py   (PRIMESM_HIP_LIB = the experiment library)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import primestereomatch_amd as P
from primestereomatch_amd import capi, synth


def run(W, H, D, l, r, d0, d1, y0, y1, env, seg, F, steps=40, dtype="f32", hint=True):
    for k in ("PSM_PC_ORDER", "PSM_PC_DC", "PSM_PC_SLOTS", "PSM_PC_S", "PSM_PC_SPREAD", "PSM_PC_NARROW", "PSM_PC_MODEL", "PSM_PC_KDIV", "PSM_PC_DC1PEN", "PSM_PC_CONC"):
        os.environ.pop(k, None)
    os.environ.update({k: str(v) for k, v in env.items()})
    ctxs = []
    for _ in range(F):
        de = P.DispEst(l, r, D, 8, True, d_range=(d0, d1), dtype=dtype)
        de.set_option(capi.PSM_OPT_ASYNC, 1)
        if F > 1 and hint and hasattr(capi, "PSM_OPT_FRAMES_IN_FLIGHT"):
            try:
                de.set_option(capi.PSM_OPT_FRAMES_IN_FLIGHT, F)
            except capi.PsmError:
                pass                                     # (an older library in an A/B run)
        if seg:
            de.set_option(capi.PSM_OPT_SEG_ROWS, seg)
        if y1 > y0:
            de.set_rows(y0, y1)
        ctxs.append(de)
    shard = d1 - d0 < D

    def step(i):
        de = ctxs[i % F]
        de.CostConst_GPU(); de.CostFilter_GPU()
        de.DispSelect_partial() if shard else de.DispSelect_device()

    def sync():
        for de in ctxs:
            de.synchronize()
    for i in range(2 * F + 3):
        step(i)
    sync()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        sync()
        best = min(best, 1e3 * (time.perf_counter() - t0) / steps)
    for de in ctxs:
        de.close()
    return best


if __name__ == "__main__":
    W, H, D = 1920, 1080, 256
    l, r, _ = synth.make_pair(W, H, D, seed=0)
    for name, d0, d1, y0, y1 in (("disp8", 0, 32, 0, 0), ("rows8", 0, 256, 0, 135)):
        for F in (1, 2):
            for dc in ((0, 1, 2, 4, 8) if name == "disp8" else (0,)):
                for seg in ((0, 135, 180, 216, 270, 360) if name == "disp8" else (0, 45, 68, 135)):
                    env = {"PSM_PC_DC": dc} if dc else {}
                    ms = run(W, H, D, l, r, d0, d1, y0, y1, env, seg, F)
                    print(f"{name}: F={F} DC={dc} seg={seg}: {ms:.4f} ms per frame", flush=True)
    for cfg, (W, H, D), dt in (("c3", (1280, 720, 128), "f32"), ("c2", (450, 375, 64), "f32"), ("c1", (450, 375, 64), "u8")):
        l, r, _ = synth.make_pair(W, H, D, seed=0)
        for F in (1, 2):
            for dc in (0, 1, 2, 4):
                for seg in ((0,) if cfg == "c3" else (0, 94, 125, 188, 375)):
                    env = {"PSM_PC_DC": dc} if dc else {}
                    ms = run(W, H, D, l, r, 0, D, 0, 0, env, seg, F, dtype=dt)
                    print(f"{cfg}: F={F} DC={dc} seg={seg}: {ms:.4f} ms per frame", flush=True)

