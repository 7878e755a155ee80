"""Round-5 experiment (experiment build): the narrow (1+1 waves, 50 columns) against the wide (2+2 waves, 107 columns) layout of
k_cvf_pc on image widths where the narrow one needs fewer waves per row - single pair, two frames in flight, batch of 8 - with the
planner's chunk depth / segment count forced to a few alternatives."""
import os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'scripts')
from exp_plan import run
import primestereomatch_amd as P
from primestereomatch_amd import capi, synth
from primestereomatch_amd.dispest import compute_batch, share_streams


def batch(W, H, D, dt, env, seg, B=8):
    for k in ("PSM_PC_DC", "PSM_PC_SLOTS", "PSM_PC_NARROW", "PSM_PC_MODEL", "PSM_PC_KDIV", "PSM_PC_DC1PEN", "PSM_PC_CONC"):
        os.environ.pop(k, None)
    os.environ.update({k: str(v) for k, v in env.items()})
    des = [P.DispEst(*synth.make_pair(W, H, D, seed=b)[:2], D, dtype=dt) for b in range(B)]
    for de in des:
        de.set_option(capi.PSM_OPT_ASYNC, 1)
        if seg:
            de.set_option(capi.PSM_OPT_SEG_ROWS, seg)
    share_streams(des)
    for _ in range(5):
        compute_batch(des)
    des[0].synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(25):
            compute_batch(des)
        des[0].synchronize()
        best = min(best, 1e3 * (time.perf_counter() - t0) / 25 / B)
    for de in des:
        de.close()
    return best



if __name__ == "__main__":
    for cfg, (W, H, D), dt in (("c2", (450, 375, 64), "f32"), ("w340", (340, 256, 64), "f32"), ("w150", (150, 120, 32), "f32"), ("w250", (250, 200, 48), "f32")):
        l, r, _ = synth.make_pair(W, H, D, seed=0)
        for nw in (2, 1):
            line = []
            for dc in (0, 2, 4, 8):
                for k in ((0, 2, 3, 4, 5) if H > 300 else (0, 1, 2, 3, 4)):
                    if (dc == 0) != (k == 0):
                        continue
                    env = {"PSM_PC_NARROW": nw}
                    if dc:
                        env["PSM_PC_DC"] = dc
                    seg = -(-H // k) if k else 0
                    line.append(f"dc{dc}k{k} {batch(W, H, D, dt, env, seg):.4f}")
            print(cfg, "narrow" if nw == 1 else "wide", "batch-8:", "  ".join(line), flush=True)
            line = []
            for dc, k in ((0, 0), (2, 4), (2, 5), (2, 3), (4, 3), (1, 3), (1, 2)):
                env = {"PSM_PC_NARROW": nw}
                if dc:
                    env["PSM_PC_DC"] = dc
                seg = -(-H // k) if k else 0
                line.append(f"dc{dc}k{k} {run(W, H, D, l, r, 0, D, 0, 0, env, seg, 1, steps=30):.4f}/{run(W, H, D, l, r, 0, D, 0, 0, env, seg, 2, steps=30):.4f}")
            print(cfg, "narrow" if nw == 1 else "wide", "F=1/F=2:", "  ".join(line), flush=True)
