"""Experiment (not part of the suite): per-launch times of the fused kernel's forms from its own time stamps
(PSM_OPT_PROFILE 2) for a list of cases  W,H,D,d0,d1,flags,seg_rows  given on the command line.
    python scripts/exp_forms.py 1920,1080,256,0,32,0,0 1920,1080,256,0,256,2097152,0"""
import sys

import numpy as np

sys.path.insert(0, '.')
import primestereomatch_amd as P
from primestereomatch_amd import capi, synth

for case in sys.argv[1:]:
    W, H, D, d0, d1, flags, seg = (int(v) for v in case.split(","))
    l, r, _ = synth.make_pair(W, H, D, seed=0)
    with P.DispEst(l, r, D, d_range=(d0, d1)) as de:
        de.set_option(capi.PSM_OPT_PROFILE, 2)
        if flags:
            de.set_option(capi.PSM_OPT_FLAGS, flags)
        if seg:
            de.set_option(capi.PSM_OPT_SEG_ROWS, seg)
        res = []
        for it in range(6):
            de.CostConst_GPU(); de.CostFilter_GPU(); de.synchronize()
            t = de.filter_launch_times()
            if it >= 2:
                res.append(t)
        forms = [f for _, f in res[0]]
        ms = np.array([[m for m, _ in t] for t in res]).mean(0)
        n = d1 - d0
        print(case, "forms", forms, "ms", np.round(ms, 4).tolist(), "us/slice pair (all launches)", round(1e3 * ms.sum() / n, 2), flush=True)
