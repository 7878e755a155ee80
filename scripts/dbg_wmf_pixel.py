"""Which pixels of psm_wgt_median differ from the oracle on the long-list test input, per form (debug aid).
    python scripts/dbg_wmf_pixel.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import primestereomatch_amd as P
from primestereomatch_amd import capi
from oracle import psm_oracle_py as O
from test_gpu_pp_ocv import _wm_inputs

H, W, D = 110, 230, 96
l, lm, rm, lv, rv = _wm_inputs(H, W, D, seed=77, frac_invalid=0.55)
r = np.roll(l, 5, axis=1)
er = O.wgt_median(O.u8_to_f32(r), rm, rv, D, right=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "wmf_pixel_in.npz"), r=r, rm=rm, rv=rv, er=er)
for name, fl in [("sweeps", 0), ("nocache", 16777216), ("dataflow", 4194304)]:
    with P.DispEst(l, r, D) as de:
        de.set_option(capi.PSM_OPT_FLAGS, fl)
        de.upload_maps(lm, rm, lv, rv)
        de.WgtMedian_GPU()
        gr = de.rDisMap.copy()
    bad = np.argwhere(gr != er)
    print(name, "mismatches", len(bad), [(int(y), int(x), int(gr[y, x]), int(er[y, x])) for y, x in bad[:5]], flush=True)
    np.save(os.path.join(ROOT, "gpurun_out", f"wmf_pixel_{name}.npy"), gr)
