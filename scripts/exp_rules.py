"""Round-5 experiment (experiment build): where do pc_plan's two round-5 rules end?  (a) the narrow layout when the wave counts tie or
the narrow one needs a few more; (b) the flow model between 640 x 480 and 1280 x 720."""
import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'scripts')
from exp_plan import run
from exp_narrow import batch
from primestereomatch_amd import synth

which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "narrow"):
    for cfg, (W, H, D), dt, B in (("c1x 384 (16 v 16)", (384, 288, 64), "u8", 8), ("w200 (8 v 8)", (200, 160, 48), "f32", 8), ("w300 (12 v 12)", (300, 240, 64), "f32", 8),
                                  ("w500 (20 v 20)", (500, 375, 64), "f32", 8), ("w640 (26 v 24)", (640, 480, 128), "f32", 4), ("w1280 (52 v 48)", (1280, 720, 128), "f32", 2),
                                  ("c4 1920 (78 v 72)", (1920, 1080, 256), "f32", 0)):
        l, r, _ = synth.make_pair(W, H, D, seed=0)
        for nw in (2, 1):
            env = {"PSM_PC_NARROW": nw}
            f1 = run(W, H, D, l, r, 0, D, 0, 0, env, 0, 1, steps=20, dtype=dt)
            f2 = run(W, H, D, l, r, 0, D, 0, 0, env, 0, 2, steps=20, dtype=dt)
            b = batch(W, H, D, dt, env, 0, B=B) if B else 0
            print(f"{cfg:20s} {'narrow' if nw == 1 else 'wide  '} F=1 {f1:.4f}  F=2 {f2:.4f}  batch-{B} {b:.4f}", flush=True)
if which in ("all", "flow"):
    for cfg, (W, H, D), dt, B in (("800x600x128", (800, 600, 128), "f32", 4), ("1024x768x128", (1024, 768, 128), "f32", 2), ("960x540x96", (960, 540, 96), "f32", 4),
                                  ("720x576x64", (720, 576, 64), "f32", 8)):
        l, r, _ = synth.make_pair(W, H, D, seed=0)
        for name, mk in (("rounds", 2), ("flow", 1)):
            env = {"PSM_PC_MODEL": mk}
            f2 = run(W, H, D, l, r, 0, D, 0, 0, env, 0, 2, steps=20, dtype=dt)
            b = batch(W, H, D, dt, env, 0, B=B)
            print(f"{cfg:14s} {name:6s} F=2 {f2:.4f}  batch-{B} {b:.4f}", flush=True)
