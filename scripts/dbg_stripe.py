import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import primestereomatch_amd as P
from primestereomatch_amd import synth, capi
import psm_oracle_py as O
W, H, D, seed = 100, 12, 42, 946963188
l, r, _ = synth.make_pair(W, H, D, seed=seed & 0xffff)
for dtype in ("f32", "u8"):
    ref = (O.pipeline_u8 if dtype == "u8" else O.pipeline_f32)(l, r, D, threads=4)
    for flags in (0, 1048576):
        for (y0, y1) in [(0, 1), (1, 5), (5, 12), (0, 12), (3, 4), (11, 12), (0, 2), (2, 12)]:
            with P.DispEst(l, r, D, dtype=dtype) as c:
                c.set_option(capi.PSM_OPT_FLAGS, flags)
                c.set_rows(y0, y1)
                c.CostConst_GPU(); c.CostFilter_GPU(); c.DispSelect_GPU()
                okl = np.array_equal(c.lDisMap[y0:y1], ref["ldisp"][y0:y1]); okr = np.array_equal(c.rDisMap[y0:y1], ref["rdisp"][y0:y1])
                if not (okl and okr):
                    bad = np.argwhere(c.lDisMap[y0:y1] != ref["ldisp"][y0:y1])
                    print(dtype, flags, (y0, y1), "L" if not okl else "", "R" if not okr else "", "bad L px:", len(bad), bad[:5].tolist())
print("done")
