"""Round-5 experiment (experiment build): the planner's cost model - rounds of resident workgroups (the product's) against a flow
model (work / slots + half an item) - over single pairs, two frames in flight and batches; and dc = 1 in batches.

Kept as the record of what produced profiles/r05/exp_plan_model.txt: PSM_PC_MODEL = 3 (a rounds / flow blend), PSM_PC_CONC (pairs in
flight, now the product option PSM_OPT_FRAMES_IN_FLIGHT) and PSM_PC_DC1PEN were knobs of that session's experiment build only; today's
`make exp` knows PSM_PC_MODEL 1 (flow) / 2 (rounds), PSM_PC_KDIV, PSM_PC_NARROW, PSM_PC_DC - the other settings fall back to the
product's rule."""
import os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'scripts')
import exp_plan
from exp_plan import run
from exp_narrow import batch  # noqa (exp_narrow's module-level sweep is skipped: see __name__ guard)
from primestereomatch_amd import synth

KN = ("PSM_PC_MODEL", "PSM_PC_KDIV", "PSM_PC_DC1PEN", "PSM_PC_CONC")


def clear():
    for k in KN:
        os.environ.pop(k, None)


which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "small"):
    for cfg, (W, H, D), dt in (("c2", (450, 375, 64), "f32"), ("c1", (450, 375, 64), "u8"), ("c1x", (384, 288, 64), "u8"), ("w340", (340, 256, 64), "f32"),
                               ("w150", (150, 120, 32), "f32"), ("w250", (250, 200, 48), "f32"), ("w640", (640, 480, 128), "f32")):
        l, r, _ = synth.make_pair(W, H, D, seed=0)
        for name, env in (("rounds", {}), ("flow", {"PSM_PC_MODEL": 1}), ("blend", {"PSM_PC_MODEL": 3}), ("blend k32", {"PSM_PC_MODEL": 3, "PSM_PC_KDIV": 32}),
                          ("blend k32 c2", {"PSM_PC_MODEL": 3, "PSM_PC_KDIV": 32, "PSM_PC_CONC": 2})):
            clear()
            os.environ.update({k: str(v) for k, v in env.items()})
            exp_plan.KEEP = KN
            f1 = run(W, H, D, l, r, 0, D, 0, 0, env, 0, 1, steps=30, dtype=dt)
            f2 = run(W, H, D, l, r, 0, D, 0, 0, env, 0, 2, steps=30, dtype=dt)
            b8 = batch(W, H, D, dt, env, 0)
            print(f"{cfg} {name:11s} F=1 {f1:.4f}  F=2 {f2:.4f}  batch-8 {b8:.4f}", flush=True)
if which in ("all", "big"):
    for cfg, (W, H, D), dt, B in (("c3", (1920, 1080, 64), "f32", 2), ("c4", (1920, 1080, 256), "f32", 2), ("c5", (1920, 1080, 256), "u8", 2), ("w1280", (1280, 720, 128), "f32", 2)):
        l, r, _ = synth.make_pair(W, H, D, seed=0)
        for name, env in (("rounds", {}), ("blend", {"PSM_PC_MODEL": 3}), ("blend c2", {"PSM_PC_MODEL": 3, "PSM_PC_CONC": 2})):
            f1 = run(W, H, D, l, r, 0, D, 0, 0, env, 0, 1, steps=12, dtype=dt)
            f2 = run(W, H, D, l, r, 0, D, 0, 0, env, 0, 2, steps=12, dtype=dt) if B else 0
            b = batch(W, H, D, dt, env, 0, B=B) if B else 0
            sh = run(W, H, D, l, r, 0, D // 8, 0, 0, env, 0, 1, steps=20, dtype=dt)
            st = run(W, H, D, l, r, 0, D, 0, H // 8, env, 0, 1, steps=20, dtype=dt)
            print(f"{cfg} {name:8s} F=1 {f1:.4f}  F=2 {f2:.4f}  batch-{B} {b:.4f}  D-shard/8 {sh:.4f}  row stripe/8 {st:.4f}", flush=True)
