"""Experiment (round 5): slices per chunk (DC) of single-phase plane launches on disparity shards of large images."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'scripts')
from exp_plan import run
from primestereomatch_amd import synth
for name, (W, H, D), d1 in (("1080p 1/8", (1920, 1080, 256), 32), ("1080p 1/4", (1920, 1080, 256), 64), ("1080p 1/16", (1920, 1080, 256), 16),
                            ("4K 1/8", (3840, 2160, 256), 32), ("720p 1/4", (1280, 720, 128), 32), ("720p 1/2", (1280, 720, 128), 64)):
    l, r, _ = synth.make_pair(W, H, D, seed=0)
    for rep in range(2):
        for dc in (0, 2, 4, 8):
            ms = run(W, H, D, l, r, 0, d1, 0, 0, {"PSM_PC_DC": dc} if dc else {}, 0, 1, steps=30 if W < 3000 else 10)
            print(f"{name}: DC={dc or 'auto'}: {ms:.4f} ms per frame", flush=True)
