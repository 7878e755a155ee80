import sys, numpy as np
sys.path.insert(0, '.')
import primestereomatch_amd as P
from primestereomatch_amd import capi, synth
def run(l, r, D, flags, d_range=None):
    with P.DispEst(l, r, D, d_range=d_range) as de:
        de.set_option(capi.PSM_OPT_FLAGS, flags)
        de.CostConst_GPU(); de.CostFilter_GPU()
        if d_range is None:
            de.DispSelect_GPU(); return de.lDisMap.copy(), de.rDisMap.copy()
        de.DispSelect_partial(); de.DispSelect_merge(de.partial_keys()[0], 1); return de.lDisMap.copy(), de.rDisMap.copy()
for (W,H,D) in [(294,10,34),(294,40,34),(294,10,8),(120,10,8),(294,10,2),(1920,1080,256),(228,64,16),(114,64,16),(115,64,16),(300,64,16)]:
    l, r, _ = synth.make_pair(W, H, D, seed=1)
    dr = (0, 4) if W == 1920 else None
    a = run(l, r, D, 0, dr); b = run(l, r, D, 16384, dr)
    for s, (x, y) in enumerate(zip(a, b)):
        bad = np.argwhere(x != y)
        print(W, H, D, 'side', s, 'mismatch', len(bad), 'rows', sorted(set(bad[:,0]))[:12], 'cols', sorted(set(bad[:,1]))[:16], '...', sorted(set(bad[:,1]))[-5:] if len(bad) else '')
