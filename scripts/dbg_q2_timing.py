import ctypes as C, os, sys
sys.path.insert(0, '.')
import primestereomatch_amd as P
from primestereomatch_amd import capi, synth
lib = capi.load()
W, H, D = 1920, 1080, 256
l, r, _ = synth.make_pair(W, H, D, seed=0)
with P.DispEst(l, r, D) as de:
    de.set_option(capi.PSM_OPT_FLAGS, 16384 + 65536)
    for it in range(2):
        de.CostConst_GPU(); de.CostFilter_GPU(); de.synchronize()
        out = (C.c_ulonglong * 8)()
        lib.psm_debug_q2_cycles(out)
        v = list(out)
        if it == 1:
            tot = [v[i] + v[4 + i] for i in range(4)]
            print("role   work(Mcyc)  wait(Mcyc)  work%")
            for i, n in enumerate(("A1", "A2", "B1", "B2")):
                print(f"{n}   {v[i]/1e6:10.1f}  {v[4+i]/1e6:10.1f}  {100*v[i]/max(tot[i],1):5.1f}")
