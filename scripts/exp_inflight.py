"""Experiment (round 5): several frames in flight - F contexts of the same geometry, each on its own stream, steps dealt to them
round-robin - against one frame at a time.  A frame's launches are prep / guidance / fused (1 or 2) / reduction / merge: short
kernels that cannot fill the chip, and the tail of every fused launch (the last round of workgroups is partly empty), run
beside the NEXT frame's fused kernel instead of alone.
    python scripts/exp_inflight.py [case ...]      case = name,W,H,D,d0,d1,y0,y1,dtype
Prints one line per (case, F, priority mode): ms per frame; every context's maps are compared with context 0's.
"""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import primestereomatch_amd as P
from primestereomatch_amd import capi, synth

hip = C.CDLL("libamdhip64.so")
lo, hi = C.c_int(), C.c_int()
hip.hipDeviceGetStreamPriorityRange(C.byref(lo), C.byref(hi))
print("stream priority range (least, greatest):", lo.value, hi.value, flush=True)


def mk_stream(prio):
    s = C.c_void_p()
    rc = hip.hipStreamCreateWithPriority(C.byref(s), 1, prio)   # 1 = hipStreamNonBlocking
    assert rc == 0, rc
    return s


CASES = sys.argv[1:] or [
    "c4,1920,1080,256,0,256,0,0,f32",
    "c4_disp8,1920,1080,256,0,32,0,0,f32",
    "c4_rows8,1920,1080,256,0,256,0,135,f32",
    "c3,1280,720,128,0,128,0,0,f32",
    "c2,450,375,64,0,64,0,0,f32",
    "c1,450,375,64,0,64,0,0,u8",
]
STEPS = 40
for case in CASES:
    name, W, H, D, d0, d1, y0, y1, dt = case.split(",")
    W, H, D, d0, d1, y0, y1 = (int(v) for v in (W, H, D, d0, d1, y0, y1))
    l, r, _ = synth.make_pair(W, H, D, seed=0)
    steps = STEPS if W * H * D < 3e8 else 20
    for F, mode in ((1, "own"), (2, "own"), (2, "prio"), (3, "own"), (3, "prio"), (4, "own")):
        ctxs = []
        for k in range(F):
            de = P.DispEst(l, r, D, 8, True, d_range=(d0, d1), dtype=dt)
            de.set_option(capi.PSM_OPT_ASYNC, 1)
            if y1 > y0:
                de.set_rows(y0, y1)
            ctxs.append(de)
        streams = []
        if mode == "prio":     # frame k of the ring: the older frame gets the higher priority (hi is numerically lower)
            for k, de in enumerate(ctxs):
                s = mk_stream(hi.value if k == 0 else lo.value)
                streams.append(s)
                de.set_stream(s.value)
        shard = (d1 - d0) < D

        def step(i):
            de = ctxs[i % F]
            de.CostConst_GPU()
            de.CostFilter_GPU()
            if shard:
                de.DispSelect_partial()
            else:
                de.DispSelect_device()

        def sync():
            for de in ctxs:
                de.synchronize()

        for i in range(2 * F + 2):
            step(i)
        sync()
        best = None
        for rep in range(3):
            t0 = time.perf_counter()
            for i in range(steps):
                step(i)
            sync()
            ms = 1e3 * (time.perf_counter() - t0) / steps
            best = ms if best is None else min(best, ms)
        ok = True
        if not shard:
            ref = [m.copy() for m in ctxs[0].download_maps()]
            for de in ctxs[1:]:
                got = de.download_maps()
                ok = ok and all(np.array_equal(a[y0:y1 or H], b[y0:y1 or H]) for a, b in zip(ref, got))
        print(f"{name}: F={F} {mode}: {best:.4f} ms per frame, maps equal: {ok}", flush=True)
        for de in ctxs:
            de.close()
        for s in streams:
            hip.hipStreamDestroy(s)
