"""Shader clock seen by the workgroups of the last k_cvf_pc launch (library built with -DPSM_PC_TIMING=1): cycles of the
shader-clock counter per microsecond of the constant 100 MHz counter, by start time within the launch.
    python scripts/dbg_pc_clock.py W,H,D,d0,d1,flags,seg_rows,frames,sync ..."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, '.')
import primestereomatch_amd as P
from primestereomatch_amd import capi, synth

lib = capi.load()
N = 1 << 16
for case in sys.argv[1:]:
    W, H, D, d0, d1, flags, seg, frames, sync = (int(v) for v in case.split(","))
    l, r, _ = synth.make_pair(W, H, D, seed=0)
    with P.DispEst(l, r, D, d_range=(d0, d1)) as de:
        if flags:
            de.set_option(capi.PSM_OPT_FLAGS, flags)
        if seg:
            de.set_option(capi.PSM_OPT_SEG_ROWS, seg)
        lib.psm_debug_pc_trace(None, None, None, 0)           # (clears the stamps of earlier, larger launches)
        for it in range(frames):
            de.CostConst_GPU(); de.CostFilter_GPU()
            if sync:
                de.synchronize()
        de.synchronize()
        out = (C.c_ulonglong * (3 * N))(); clk = (C.c_ulonglong * (2 * N))()
        lib.psm_debug_pc_trace(out, clk, None, N)
    t = np.frombuffer(out, dtype=np.uint64).reshape(N, 3); c = np.frombuffer(clk, dtype=np.uint64).reshape(N, 2)
    m = t[:, 0] > 0
    t, c = t[m], c[m]
    st = (t[:, 0] - t[:, 0].min()).astype(np.float64) * 0.01; en = (t[:, 1] - t[:, 0].min()).astype(np.float64) * 0.01
    mhz = (c[:, 1] - c[:, 0]).astype(np.float64) / (en - st)
    o = np.argsort(st)
    parts = [o[i * len(o) // 8:(i + 1) * len(o) // 8] for i in range(8)]
    print(case, f"launch {en.max():.0f} us; MHz by start octile:", " ".join(f"{mhz[p].mean():.0f}" for p in parts),
          "| WG us by octile:", " ".join(f"{(en - st)[p].mean():.0f}" for p in parts), flush=True)
