"""Experiment (round 6, experiment build): the planner's cut of the two launches of the two-phase selection, per form - segment rows
of the plane launch (PSM_PC_SEGP) and of the key launch (PSM_PC_SEGK), seed stride (PSM_PC_S), slices per chunk (PSM_PC_DC) - at the
sizes below the headline.      PRIMESM_HIP_LIB=.../libprimesm_hip_exp.so python scripts/exp_plan6.py [c3|c4|c5|rows8]"""
import os
import sys
import time

sys.path.insert(0, '.')
import primestereomatch_amd as P
from primestereomatch_amd import capi, synth

KNOBS = ("PSM_PC_DC", "PSM_PC_S", "PSM_PC_SEGP", "PSM_PC_SEGK", "PSM_PC_SPREAD", "PSM_PC_SLOTS")


def run(l, r, D, env, steps, rows=None, dtype="f32"):
    for k in KNOBS:
        os.environ.pop(k, None)
    os.environ.update({k: str(v) for k, v in env.items() if v})
    de = P.DispEst(l, r, D, 8, True, dtype=dtype)
    de.set_option(capi.PSM_OPT_ASYNC, 1)
    if rows:
        de.set_rows(*rows)
    for _ in range(4):
        de.CostConst_GPU(); de.CostFilter_GPU(); de.DispSelect_device()
    de.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(steps):
            de.CostConst_GPU(); de.CostFilter_GPU(); de.DispSelect_device()
        de.synchronize()
        best = min(best, 1e3 * (time.perf_counter() - t0) / steps)
    de.close()
    return best


CFG = {"c3": (1280, 720, 128, 40, None), "c4": (1920, 1080, 256, 12, None), "c5": (3840, 2160, 256, 4, None), "rows8": (1920, 1080, 256, 40, (0, 135))}
if __name__ == "__main__":
    for cfg in (sys.argv[1:] or ["c3"]):
        W, H, D, steps, rows = CFG[cfg]
        R = rows[1] - rows[0] if rows else H
        l, r, _ = synth.make_pair(W, H, D, seed=0)
        base = run(l, r, D, {}, steps, rows)
        print(f"{cfg}: planner's own cut: {base:.4f} ms", flush=True)
        segs = [0] + sorted({-(-R // k) for k in (1, 2, 3, 4, 5, 6, 8, 10)}, reverse=True)
        best = (base, {})
        for S in (0, 3, 4, 6, 8):
            for sk in segs:
                env = {"PSM_PC_S": S, "PSM_PC_SEGK": sk}
                ms = run(l, r, D, env, steps, rows)
                print(f"{cfg}: S={S} segK={sk}: {ms:.4f}", flush=True)
                if ms < best[0]:
                    best = (ms, env)
        e0 = dict(best[1])
        for dc in (0, 1, 2, 4):
            for sp in segs:
                env = dict(e0, PSM_PC_DC=dc, PSM_PC_SEGP=sp)
                ms = run(l, r, D, env, steps, rows)
                print(f"{cfg}: {e0} DC={dc} segP={sp}: {ms:.4f}", flush=True)
                if ms < best[0]:
                    best = (ms, env)
        print(f"{cfg}: BEST {best[0]:.4f} ms ({100 * (best[0] / base - 1):+.1f} %) with {best[1]}; same again: {run(l, r, D, best[1], steps, rows):.4f}, planner again: {run(l, r, D, {}, steps, rows):.4f}", flush=True)
