"""Small driver for rocprofv3 counter passes: a few launches of every hot kernel at a BASELINE size.
    python scripts/prof_run.py [config] [seg_rows] [waves]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import primestereomatch_amd as P  # noqa: E402
from primestereomatch_amd import capi, synth  # noqa: E402

cfg = {"c4": (1920, 1080, 256), "c3": (1280, 720, 128), "c2": (450, 375, 64), "c5": (3840, 2160, 256)}[sys.argv[1] if len(sys.argv) > 1 else "c4"]
W, H, D = cfg
l, r, _ = synth.make_pair(W, H, D, seed=0)
dr = tuple(int(v) for v in os.environ["PSM_DRANGE"].split(",")) if os.environ.get("PSM_DRANGE") else None   # disparity shard
de = P.DispEst(l, r, D, dtype=os.environ.get("PSM_DTYPE", "f32"), d_range=dr)
if len(sys.argv) > 2:
    de.set_option(capi.PSM_OPT_SEG_ROWS, int(sys.argv[2]))
if len(sys.argv) > 3:
    de.set_option(capi.PSM_OPT_WAVES, int(sys.argv[3]))
if os.environ.get("PSM_FLAGS"):
    de.set_option(capi.PSM_OPT_FLAGS, int(os.environ["PSM_FLAGS"]))
for _ in range(2):
    de.CostConst_GPU()
    if os.environ.get("PSM_BOX"):
        de.box8_volume(0, download=False)
    de.CostFilter_GPU()
    if dr is None:
        de.DispSelect_device()
de.synchronize()
de.close()
