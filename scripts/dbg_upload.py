"""Host-side cost of the blocking entries (psm_upload_pair, psm_download_maps) by image width: python scripts/dbg_upload.py"""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import primestereomatch_amd as P
from primestereomatch_amd import synth

for W, H in [(450, 375), (448, 375), (452, 375), (451, 375), (384, 288), (1280, 720), (1920, 1080), (1919, 1080)]:
    l, r, _ = synth.make_pair(W, H, 16, seed=0)
    with P.DispEst(l, r, 16) as de:
        de.CostConst_GPU(); de.CostFilter_GPU(); de.DispSelect_GPU()
        up, dn = [], []
        for _ in range(5):
            de.synchronize(); t = time.perf_counter(); de.setInputImages(l, r); up.append(1e3 * (time.perf_counter() - t))
            de.CostConst_GPU(); de.CostFilter_GPU(); de.DispSelect_device(); de.synchronize()
            t = time.perf_counter(); de.download_maps(); dn.append(1e3 * (time.perf_counter() - t))
        print(f"{W}x{H}: upload pair {min(up):.3f} ms (median {np.median(up):.3f}), download maps {min(dn):.3f} ms", flush=True)
