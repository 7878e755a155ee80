"""Where does the host spend its time in a batched frame loop?  python scripts/dbg_batch_loop.py [B]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import primestereomatch_amd as P
from primestereomatch_amd import capi, synth
from primestereomatch_amd.dispest import compute_batch, share_streams
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
W, H, D = 450, 375, 64
pairs = [synth.make_pair(W, H, D, seed=b)[:2] for b in range(B)]
des = [P.DispEst(l, r, D) for l, r in pairs]
if len(sys.argv) > 2: share_streams(des)
des[0].set_option(capi.PSM_OPT_ASYNC, 1)
acc = {}
def T(name, fn, *a):
    t = time.perf_counter(); r = fn(*a); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t; return r
def frame(i, last, up=True, down=True):
    T("compute", compute_batch, des)
    for o, (l, r) in zip(des, pairs):
        if up and not last: T("upload_async", o.setInputImages_async, l, r)
    for o in des:
        if down and i > 0: T("down_wait", o.download_maps_wait)
        if down: T("down_async", o.download_maps_async)
for mode in ("both", "up only", "down only", "none"):
    up, down = mode in ("both", "up only"), mode in ("both", "down only")
    for i in range(3): frame(i, False, up, down)
    if down:
        for o in des: o.download_maps_wait()
    des[0].synchronize(); compute_batch(des); des[0].synchronize()
    acc.clear(); n = 20
    t0 = time.perf_counter()
    for i in range(n): frame(i, i + 1 == n, up, down)
    if down:
        for o in des: T("down_wait", o.download_maps_wait)
    T("final sync", des[0].synchronize)
    tot = time.perf_counter() - t0
    print(f"B={B} {mode:10s} {1e3*tot/n:.3f} ms/frame | host per frame:", {k: round(1e3 * v / n, 3) for k, v in acc.items()})
