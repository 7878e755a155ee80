"""Per-workgroup trace of the last k_cvf_pc launch (library built with -DPSM_PC_TIMING=1, PRIMESM_HIP_LIB points at it):
how many workgroups are resident over the launch, how long they run, when each XCD goes idle.
    python scripts/dbg_pc_trace.py W,H,D,d0,d1,flags,seg_rows [out.npz]"""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, '.')
import primestereomatch_amd as P
from primestereomatch_amd import capi, synth

lib = capi.load()
W, H, D, d0, d1, flags, seg = (int(v) for v in sys.argv[1].split(","))
l, r, _ = synth.make_pair(W, H, D, seed=0)
N = 1 << 16
with P.DispEst(l, r, D, d_range=(d0, d1)) as de:
    if flags:
        de.set_option(capi.PSM_OPT_FLAGS, flags)
    if seg:
        de.set_option(capi.PSM_OPT_SEG_ROWS, seg)
    for it in range(3):
        de.CostConst_GPU(); de.CostFilter_GPU(); de.synchronize()
    out = (C.c_ulonglong * (3 * N))()
    sm = (C.c_ubyte * (4 * N))()
    lib.psm_debug_pc_trace(out, None, sm, N)
t = np.frombuffer(out, dtype=np.uint64).reshape(N, 3)
simd = np.frombuffer(sm, dtype=np.uint8).reshape(N, 4)[t[:, 0] > 0]
t = t[t[:, 0] > 0]
from collections import Counter
print("SIMD of waves (A0, A1, B0, B1) -> workgroups:", sorted(Counter(map(tuple, simd.tolist())).items(), key=lambda kv: -kv[1])[:12])
st = (t[:, 0] - t[:, 0].min()).astype(np.float64) * 0.01      # us (100 MHz)
en = (t[:, 1] - t[:, 0].min()).astype(np.float64) * 0.01
xcc = (t[:, 2] >> np.uint64(32)).astype(int) & 15
hw = t[:, 2].astype(np.uint64) & np.uint64(0xffffffff)
cu = ((hw >> np.uint64(8)) & np.uint64(15)).astype(int); se = ((hw >> np.uint64(13)) & np.uint64(7)).astype(int)
dur = en - st
T = en.max()
print(f"{sys.argv[1]}: {len(t)} workgroups (traced), launch {T:.1f} us; duration mean {dur.mean():.1f} min {dur.min():.1f} max {dur.max():.1f} us")
edges = np.linspace(0, T, 41)
act = [(np.minimum(en, b) - np.maximum(st, a)).clip(0).sum() / (b - a) for a, b in zip(edges[:-1], edges[1:])]
print("resident workgroups per 1/40 of the launch:", " ".join(f"{a:.0f}" for a in act))
print("integral of residency / (1024 x launch):", round(dur.sum() / (1024 * T), 3))
for x in range(8):
    m = xcc == x
    if m.any():
        print(f"  xcc {x}: {m.sum()} wgs, first start {st[m].min():.1f}, last start {st[m].max():.1f}, last end {en[m].max():.1f}, mean dur {dur[m].mean():.1f}, cus {len(set(zip(se[m], cu[m])))}")
# duration by start decile
o = np.argsort(st)
for i in range(10):
    s = o[i * len(o) // 10:(i + 1) * len(o) // 10]
    print(f"  start decile {i}: start {st[s].mean():8.1f} us, mean dur {dur[s].mean():7.1f}")
if len(sys.argv) > 2:
    np.savez_compressed(sys.argv[2], st=st, en=en, xcc=xcc, cu=cu, se=se, simd=simd)
