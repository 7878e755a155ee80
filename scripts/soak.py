"""Randomised soak of the whole C ABI against the CPU oracle (not part of the test suite: minutes of GPU time).
    python scripts/soak.py [seconds] [seed]
Every case: random geometry / disparity range / dtype / tuning flags / segment rows, then one of: whole image, disparity
shards, row stripes, both, or a BATCH of 2-6 different pairs through psm_compute_batch (round 4; sometimes with float images,
shared streams, a second frame staged asynchronously), or (round 5) a FrameRing of 2-3 contexts over a short stream of
pairs; one float case in five runs the FMA reading of the solve (PSM_FLAG_FMA_SOLVE) against the oracle's same reading, merges
sometimes go through the host-staged exchange leg, scratch is sometimes released between calls; maps (and sometimes the filtered volumes, the L-R check, fill and the weighted median) must be
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
from primestereomatch_amd.dispest import compute_batch, share_streams   # noqa: E402
import psm_oracle_py as O                 # noqa: E402

FLAGS = [0, 0, 0, 0, 1048576, 1048576, 2097152, 128, 8192, 8192 | 128, 1048576 | 128]
FMA_SOLVE = 67108864


def one(rng, idx):
    # (widths around the layout boundaries of pc_plan: narrow layout for 1-50, 108-150, 215-250, 322-350, 429-450)
    W = int(rng.choice([8, 9, 49, 50, 51, 57, 96, 107, 108, 114, 128, 149, 150, 151, 213, 214, 215, 250, 251, 300, 321, 322, 350, 351, 428, 429, 450])) if rng.random() < 0.5 else int(rng.integers(8, 460))
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
    mode = rng.choice(["whole", "whole", "shards", "stripes", "both", "batch", "ring"])
    fma = dtype == "f32" and mode not in ("batch", "ring") and rng.random() < 0.2
    if fma:
        flags |= FMA_SOLVE
        with O.variant(O.VAR_FMA_SOLVE):
            ref = O.pipeline_f32(l, r, D, threads=8, want_volumes=(D <= 24))
    else:
        ref = (O.pipeline_u8 if dtype == "u8" else O.pipeline_f32)(l, r, D, threads=8, want_volumes=(dtype == "f32" and D <= 24))
    inflight = int(rng.integers(2, 5)) if rng.random() < 0.25 else 1     # the planner's hint: another cut of the launches, same bits
    desc = f"case {idx}: {W}x{H} D={D} {dtype} flags={flags} seg={seg} mode={mode} inflight={inflight}"

    def setup(c):
        c.set_option(capi.PSM_OPT_FLAGS, flags)
        if inflight > 1:
            c.set_option(capi.PSM_OPT_FRAMES_IN_FLIGHT, inflight)
        if seg > 0:
            c.set_option(capi.PSM_OPT_SEG_ROWS, seg)

    ok = True
    if mode == "ring":
        # a FrameRing of 2-3 contexts over a short stream of different pairs: every frame's maps, in order, against its own oracle run
        F = int(rng.integers(2, 4))
        n = int(rng.integers(F, F + 4))
        pairs = [(l, r)] + [synth.make_pair(W, H, D, seed=int(rng.integers(0, 1 << 16)))[:2] for _ in range(n - 1)]
        refs = [ref] + [(O.pipeline_u8 if dtype == "u8" else O.pipeline_f32)(a, b, D, threads=8) for a, b in pairs[1:]]
        got = []
        with P.FrameRing(l, r, D, frames=F, dtype=dtype) as ring:
            for c in ring.ctx:
                c.set_option(capi.PSM_OPT_FLAGS, flags if flags in (0, 1048576, 2097152) else 0)
            for a, b in pairs:
                out = ring.push(a, b)
                if out is not None:
                    got.append(out)
            got += ring.flush()
        ok = len(got) == n and all(np.array_equal(g[0], e["ldisp"]) and np.array_equal(g[1], e["rdisp"]) for g, e in zip(got, refs))
        return ok, desc + f" F={F} frames={n}"
    if mode == "batch":
        # B different pairs of this geometry in shared launches; every pair against its own oracle run.  Only the select
        # forms batch (flags 0 / two-phase on / off).
        fl = flags if flags in (0, 1048576, 2097152) else 0
        B = int(rng.integers(2, 7))
        as_float = dtype == "f32" and rng.random() < 0.3          # images as the reference hands them over (x 1/255)
        pairs = [(l, r)] + [synth.make_pair(W, H, D, seed=int(rng.integers(0, 1 << 16)))[:2] for _ in range(B - 1)]
        conv = (lambda a: O.u8_to_f32(a)) if as_float else (lambda a: a)
        des = [P.DispEst(conv(a), conv(b), D, dtype=dtype) for a, b in pairs]
        try:
            for de in des:
                de.set_option(capi.PSM_OPT_FLAGS, fl)
                if seg > 0:
                    de.set_option(capi.PSM_OPT_SEG_ROWS, seg)
            if rng.random() < 0.5:
                share_streams(des)
            refs = [ref] + [(O.pipeline_u8 if dtype == "u8" else O.pipeline_f32)(a, b, D, threads=8) for a, b in pairs[1:]]
            frames = 2 if rng.random() < 0.4 else 1
            for f in range(frames):
                compute_batch(des)
                if f + 1 < frames:                                  # next frame: the pairs rotate by one, staged asynchronously
                    for k, de in enumerate(des):
                        a, b = pairs[(k + 1) % B]
                        de.setInputImages_async(conv(a), conv(b))
            rot = frames - 1
            for k, de in enumerate(des):
                lm, rm = de.download_maps()
                e = refs[(k + rot) % B]
                ok &= np.array_equal(lm, e["ldisp"]) and np.array_equal(rm, e["rdisp"])
        finally:
            for de in des:
                de.close()
        return ok, desc + f" B={B} flags={fl} float={as_float}"
    if mode == "whole":
        with P.DispEst(l, r, D, dtype=dtype) as de:
            setup(de)
            de.CostConst_GPU(); de.CostFilter_GPU(); de.DispSelect_GPU()
            ok &= np.array_equal(de.lDisMap, ref["ldisp"]) and np.array_equal(de.rDisMap, ref["rdisp"])
            if rng.random() < 0.2:
                de.release_scratch()             # (maps, minima and volumes are state, not scratch: everything below still holds)
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
        flags &= FMA_SOLVE    # stripes need the default select form (either arithmetic reading)
    outl, outr = np.zeros_like(ref["ldisp"]), np.zeros_like(ref["rdisp"])
    for y0, y1 in zip(ycuts[:-1], ycuts[1:]):
        # (round 6) sometimes the same number of shards dealt the other way: shard g of G holds d = g (mod G) - strided ownership needs
        # the select path, so not with the storing / materialising flags
        G = len(dcuts) - 1
        if G > 1 and G <= D and not (flags & (8192 | 128)) and rng.random() < 0.4:
            shards = [P.DispEst(l, r, D, dtype=dtype, d_stride=(g, G)) for g in range(G)]
        else:
            shards = [P.DispEst(l, r, D, dtype=dtype, d_range=(d0, d1)) for d0, d1 in zip(dcuts[:-1], dcuts[1:])]
        try:
            for s in shards:
                s.set_option(capi.PSM_OPT_FLAGS, flags)
                if inflight > 1:
                    s.set_option(capi.PSM_OPT_FRAMES_IN_FLIGHT, inflight)
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
                if rng.random() < 0.3:
                    shards[0].set_option(capi.PSM_OPT_GATHER_STAGED, 1)     # the exchange leg through page-locked host memory
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
