"""A/B helper (round 5): ms per frame of the Middlebury-size configurations, one / two frames in flight, and a batch of 8, for the library
PRIMESM_HIP_LIB points at."""
import sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'scripts')
from exp_plan import run
import primestereomatch_amd as P
from primestereomatch_amd import capi, synth
from primestereomatch_amd.dispest import compute_batch, share_streams
for cfg, (W, H, D), dt in (("c2", (450, 375, 64), "f32"), ("c1", (450, 375, 64), "u8"), ("c1x", (384, 288, 64), "u8"), ("w340", (340, 256, 64), "f32"), ("w150", (150, 120, 32), "f32"), ("w640", (640, 480, 128), "f32")):
    l, r, _ = synth.make_pair(W, H, D, seed=0)
    out = []
    for F, seg, hint in ((1, 0, True), (2, 0, True), (2, H, False), (2, 0, False)):
        out.append(run(W, H, D, l, r, 0, D, 0, 0, {}, seg, F, dtype=dt, hint=hint))
    des = [P.DispEst(*synth.make_pair(W, H, D, seed=b)[:2], D, dtype=dt) for b in range(8)]
    for de in des:
        de.set_option(capi.PSM_OPT_ASYNC, 1)
    share_streams(des)
    for _ in range(5):
        compute_batch(des)
    des[0].synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(30):
            compute_batch(des)
        des[0].synchronize()
        best = min(best, 1e3 * (time.perf_counter() - t0) / 30 / 8)
    for de in des:
        de.close()
    print(f"{cfg}: F=1 {out[0]:.4f}  F=2 {out[1]:.4f}  F=2 one segment {out[2]:.4f}  F=2 no hint {out[3]:.4f}  batch-8 per pair {best:.4f}", flush=True)
