"""Experiment (round 5): with two frames in flight, which segment count does a launch want?  (the planner's cost model assumes the
launch runs alone: rounds + 1/2)"""
import os, sys, time
import numpy as np
sys.path.insert(0, '.')
sys.path.insert(0, 'scripts')
from exp_plan import run
from primestereomatch_amd import synth
for cfg, (W, H, D), dt, d1, y1, segs in (("c3", (1280, 720, 128), "f32", 128, 0, (0, 720, 360, 240, 180, 144, 120)),
                                          ("c2", (450, 375, 64), "f32", 64, 0, (0, 375, 188)),
                                          ("c1x", (384, 288, 64), "u8", 64, 0, (0, 288, 144, 96)),
                                          ("rows8", (1920, 1080, 256), "f32", 256, 135, (0, 135, 68)),
                                          ("disp8", (1920, 1080, 256), "f32", 32, 0, (0, 1080, 540, 360, 270, 216))):
    l, r, _ = synth.make_pair(W, H, D, seed=0)
    for F in (1, 2):
        for seg in segs:
            ms = run(W, H, D, l, r, 0, d1, 0, y1, {}, seg, F, dtype=dt)
            print(f"{cfg}: F={F} seg={seg}: {ms:.4f} ms per frame", flush=True)
