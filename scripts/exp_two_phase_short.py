"""Experiment (round 5): the two-phase selection on SHORT launches (a 32- / 64-slice disparity shard, 720p x 128) - segment rows and
seed stride; launch times of the two phases from the kernel's own stamps."""
import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'scripts')
import numpy as np
from exp_order import run
from primestereomatch_amd import capi
ON = capi.PSM_FLAG_TWO_PHASE_ON
for d1 in (32, 64):
    for S in (4, 5, 8):
        for seg in (0, 135, 180, 270, 360, 540, 1080):
            ms, lt, _ = run(0, d1, {"PSM_PC_S": S}, seg, ON, 30)
            print(f"disp 0..{d1}: S={S} seg={seg}: {ms:.4f} ms per frame; launches {lt}", flush=True)
