"""Randomised soak of the whole C ABI against the CPU oracle (not part of the test suite: minutes of GPU time).
    python scripts/soak.py [seconds] [seed]
Every case: random geometry / disparity range / dtype / tuning flags / segment rows, then one of: whole image, disparity
shards, row stripes, both; maps (and sometimes the filtered volumes, the L-R check, fill and the weighted median) must be
bit-identical to the oracle.  Prints one line per failure and a summary."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import primestereomatch_amd as P          # noqa: E402
from primestereomatch_amd import capi, synth   # noqa: E402
import psm_oracle_py as O                 # noqa: E402

FLAGS = [0, 0, 0, 0, 1048576, 1048576, 2097152, 128, 8192, 8192 | 128, 1048576 | 128]


def one(rng, idx):
    W = int(rng.choice([8, 9, 57, 96, 107, 108, 114, 128, 213, 214, 215, 300])) if rng.random() < 0.5 else int(rng.integers(8, 340))
    H = int(rng.choice([8, 9, 11, 12, 15, 16, 17, 31, 33, 63, 64, 65, 71])) if rng.random() < 0.5 else int(rng.integers(8, 100))
    D = int(rng.integers(1, min(W, 256) + 1)) if rng.random() < 0.2 else int(rng.integers(1, min(W, 40) + 1))
    dtype = "u8" if rng.random() < 0.25 else "f32"
    flags = int(rng.choice(FLAGS))
    if dtype == "u8" and flags & 8192:
        flags = 0
    seg = int(rng.integers(8, 48)) if rng.random() < 0.4 else -1
    l, r, _ = synth.make_pair(W, H, D, seed=int(rng.integers(0, 1 << 16)))
    if rng.random() < 0.3:
        a, b = sorted(int(v) for v in rng.integers(0, H, 2)); c, d = sorted(int(v) for v in rng.integers(0, W, 2))
        l[a:b + 1, c:d + 1] = int(rng.integers(0, 256)); r[a:b + 1, c:d + 1] = l[a, c]
    ref = (O.pipeline_u8 if dtype == "u8" else O.pipeline_f32)(l, r, D, threads=8, want_volumes=(dtype == "f32" and D <= 24))
    mode = rng.choice(["whole", "whole", "shards", "stripes", "both"])
    desc = f"case {idx}: {W}x{H} D={D} {dtype} flags={flags} seg={seg} mode={mode}"

    def setup(c):
        c.set_option(capi.PSM_OPT_FLAGS, flags)
        if seg > 0:
            c.set_option(capi.PSM_OPT_SEG_ROWS, seg)

    ok = True
    if mode == "whole":
        with P.DispEst(l, r, D, dtype=dtype) as de:
            setup(de)
            de.CostConst_GPU(); de.CostFilter_GPU(); de.DispSelect_GPU()
            ok &= np.array_equal(de.lDisMap, ref["ldisp"]) and np.array_equal(de.rDisMap, ref["rdisp"])
            if "lvol" in ref and rng.random() < 0.5:
                ok &= np.array_equal(de.download_volume(0), ref["lvol"]) and np.array_equal(de.download_volume(1), ref["rvol"])
            if rng.random() < 0.5:
                de.LRCheck_GPU()
                lv, rv = O.lr_check(ref["ldisp"], ref["rdisp"])
                ok &= np.array_equal(de.lValid, lv) and np.array_equal(de.rValid, rv)
                de.FillInv_GPU()
                lf, rf = O.fill_inv(ref["ldisp"], lv), O.fill_inv(ref["rdisp"], rv)
                ok &= np.array_equal(de.lDisMap, lf) and np.array_equal(de.rDisMap, rf)
                if W >= 9 and H >= 9 and W * H <= 12000 and rng.random() < 0.6:
                    wm = int(rng.choice([0, 0, 4194304, 8388608]))
                    de.set_option(capi.PSM_OPT_FLAGS, (flags & ~(4194304 | 8388608)) | wm)
                    de.WgtMedian_GPU()
                    ok &= np.array_equal(de.lDisMap, O.wgt_median(O.u8_to_f32(l), lf, lv, D, right=False))
                    ok &= np.array_equal(de.rDisMap, O.wgt_median(O.u8_to_f32(r), rf, rv, D, right=True))
        return ok, desc
    ycuts = [0, H]
    dcuts = [0, D]
    if mode in ("stripes", "both"):
        ycuts = sorted(set([0, H] + [int(v) for v in rng.integers(1, H, size=int(rng.integers(1, 4)))]))
    if mode in ("shards", "both") and D >= 2:
        dcuts = sorted(set([0, D] + [int(v) for v in rng.integers(1, D, size=int(rng.integers(1, 3)))]))
    if flags & (8192 | 128) and len(ycuts) > 2:
        flags = 0             # stripes need the default select form
    outl, outr = np.zeros_like(ref["ldisp"]), np.zeros_like(ref["rdisp"])
    for y0, y1 in zip(ycuts[:-1], ycuts[1:]):
        shards = [P.DispEst(l, r, D, dtype=dtype, d_range=(d0, d1)) for d0, d1 in zip(dcuts[:-1], dcuts[1:])]
        try:
            for s in shards:
                s.set_option(capi.PSM_OPT_FLAGS, flags)
                if seg > 0:
                    s.set_option(capi.PSM_OPT_SEG_ROWS, seg)
                if len(ycuts) > 2:
                    s.set_rows(y0, y1)
                s.CostConst_GPU(); s.CostFilter_GPU()
            if len(shards) == 1:
                shards[0].DispSelect_GPU()
            else:
                for s in shards:
                    s.DispSelect_partial()
                shards[0].DispSelect_merge_ctx(shards)
            outl[y0:y1], outr[y0:y1] = shards[0].lDisMap[y0:y1], shards[0].rDisMap[y0:y1]
        finally:
            for s in shards:
                s.close()
    ok = np.array_equal(outl, ref["ldisp"]) and np.array_equal(outr, ref["rdisp"])
    return ok, desc + f" ycuts={ycuts} dcuts={dcuts}"


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    O.build()
    rng = np.random.default_rng(seed)
    t0 = time.time()
    n = bad = 0
    while time.time() - t0 < secs:
        try:
            ok, desc = one(rng, n)
        except Exception as e:      # an error return is a failure too
            ok, desc = False, f"case {n}: EXCEPTION {type(e).__name__}: {str(e)[:300]}"
        n += 1
        if not ok:
            bad += 1
            print("FAIL", desc, flush=True)
    print(f"soak: {n} cases in {time.time() - t0:.0f} s, {bad} failures (seed {seed})")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
