import ctypes as C, os, sys
sys.path.insert(0, '.')
import primestereomatch_amd as P
from primestereomatch_amd import capi, synth
lib = capi.load()
W, H, D = 1920, 1080, 256
l, r, _ = synth.make_pair(W, H, D, seed=0)
for name, flags, fn in (("pc select (2 A + 2 B waves)", 65536, lib.psm_debug_pc_cycles), ("q2", 16384 + 65536, lib.psm_debug_q2_cycles)):
    with P.DispEst(l, r, D) as de:
        de.set_option(capi.PSM_OPT_FLAGS, flags)
        for it in range(2):
            de.CostConst_GPU(); de.CostFilter_GPU(); de.synchronize()
            out = (C.c_ulonglong * 8)()
            fn(out)
            v = list(out)
            if it == 1:
                print(name)
                for i in range(4):
                    print(f"  wave role {i}: work {v[i]/1e6:10.1f}  wait {v[4+i]/1e6:10.1f}  busy {100*v[i]/max(v[i]+v[4+i],1):5.1f} %")
