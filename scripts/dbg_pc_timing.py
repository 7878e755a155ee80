"""Per-role cycle counters of k_cvf_pc's plane form (a library built with -DPSM_PC_TIMING=1; PRIMESM_HIP_LIB points at it):
how busy the two producer and two consumer waves of a workgroup are between barriers.
    python scripts/dbg_pc_timing.py [W,H,D,d0,d1,flags,seg_rows ...]"""
import ctypes as C
import sys

sys.path.insert(0, '.')
import primestereomatch_amd as P
from primestereomatch_amd import capi, synth

lib = capi.load()
cases = sys.argv[1:] or [f"1920,1080,256,0,256,{capi.PSM_FLAG_TWO_PHASE_OFF},0"]
for case in cases:
    W, H, D, d0, d1, flags, seg = (int(v) for v in case.split(","))
    l, r, _ = synth.make_pair(W, H, D, seed=0)
    with P.DispEst(l, r, D, d_range=(d0, d1)) as de:
        if flags:
            de.set_option(capi.PSM_OPT_FLAGS, flags)      # (the counters are kept by the plane form)
        if seg:
            de.set_option(capi.PSM_OPT_SEG_ROWS, seg)
        for it in range(3):
            out = (C.c_ulonglong * 8)()
            lib.psm_debug_pc_cycles(out)                    # clear
            de.CostConst_GPU(); de.CostFilter_GPU(); de.synchronize()
            lib.psm_debug_pc_cycles(out)
        v = list(out)
        print(case, "k_cvf_pc, plane form (2 A + 2 B waves)")
        for i in range(4):
            print(f"  wave role {i}: work {v[i]/1e6:10.1f}  wait {v[4+i]/1e6:10.1f}  busy {100*v[i]/max(v[i]+v[4+i],1):5.1f} %")
