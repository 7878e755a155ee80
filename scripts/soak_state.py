"""Randomised soak of the CONTEXT STATE MACHINE (not part of the test suite): one context lives through a random sequence of
new image pairs, row stripes, tuning flags and readers of intermediate results; after every filter + select the maps of the
current stripe must equal the oracle's for the current pair.
    python scripts/soak_state.py [seconds] [seed]"""
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

SAFE_FLAGS = [0, 0, 0, 1048576, 2097152]     # compatible with row stripes
ANY_FLAGS = SAFE_FLAGS + [128, 8192, 8192 | 128]


def episode(rng, idx):
    W = int(rng.integers(8, 260)); H = int(rng.integers(8, 80))
    D = int(rng.integers(2, min(W, 140) + 1)) if rng.random() < 0.3 else int(rng.integers(2, min(W, 24) + 1))
    dtype = "u8" if rng.random() < 0.25 else "f32"
    pairs = [synth.make_pair(W, H, D, seed=int(rng.integers(0, 1 << 16)))[:2] for _ in range(3)]
    pipe = O.pipeline_u8 if dtype == "u8" else O.pipeline_f32
    refs = [pipe(l, r, D, threads=8, want_volumes=(dtype == "f32" and D <= 16)) for l, r in pairs]
    log = [f"episode {idx}: {W}x{H} D={D} {dtype}"]
    cur, pend = 0, None
    y0, y1 = 0, H
    with P.DispEst(pairs[0][0], pairs[0][1], D, dtype=dtype) as de:
        for step in range(int(rng.integers(4, 12))):
            op = rng.choice(["frame", "frame", "frame", "images", "images_async", "rows", "flags", "volume", "pp", "frame_async"])
            if op == "images":
                cur = int(rng.integers(0, 3))
                pend = None                     # a blocking upload supersedes a staged pair
                de.setInputImages(*pairs[cur]); log.append(f"images {cur}")
                continue
            if op == "images_async":            # staged for the NEXT CostConst (psm_upload_pair_async)
                pend = int(rng.integers(0, 3))
                de.setInputImages_async(*pairs[pend]); log.append(f"images_async {pend}")
                continue
            if op == "rows":
                if rng.random() < 0.4:
                    y0, y1 = 0, H
                else:
                    y0 = int(rng.integers(0, H)); y1 = int(rng.integers(y0 + 1, H + 1))
                de.set_rows(y0, y1); log.append(f"rows {y0},{y1}")
                if (y0, y1) != (0, H):          # stripes need the default select form: back to a compatible flag set
                    de.set_option(capi.PSM_OPT_FLAGS, int(rng.choice(SAFE_FLAGS)))
                continue
            if op == "flags":
                fl = int(rng.choice(SAFE_FLAGS if (y0, y1) != (0, H) else ANY_FLAGS))
                if dtype == "u8" and fl == 8192:
                    fl = 0
                de.set_option(capi.PSM_OPT_FLAGS, fl); log.append(f"flags {fl}")
                continue
            if pend is not None:                # CostConst adopts the staged pair
                cur, pend = pend, None
            if op == "frame_async":             # maps through psm_download_maps_async / _wait
                de.CostConst_GPU(); de.CostFilter_GPU(); de.DispSelect_device()
                de.download_maps_async(); de.download_maps_wait()
            else:
                de.CostConst_GPU(); de.CostFilter_GPU(); de.DispSelect_GPU()
            ref = refs[cur]
            log.append(f"frame pair {cur} rows {y0},{y1}")
            if not (np.array_equal(de.lDisMap[y0:y1], ref["ldisp"][y0:y1]) and np.array_equal(de.rDisMap[y0:y1], ref["rdisp"][y0:y1])):
                return False, " | ".join(log)
            if op == "volume" and "lvol" in ref:
                d = int(rng.integers(0, D))
                log.append(f"volume slice {d}")
                if not np.array_equal(de.download_volume(int(rng.integers(0, 2)) * 0, d, d + 1)[0], ref["lvol"][d]):
                    return False, " | ".join(log)
            if op == "pp" and (y0, y1) == (0, H):
                de.LRCheck_GPU()
                lv, rv = O.lr_check(ref["ldisp"], ref["rdisp"])
                de.FillInv_GPU()
                log.append("pp")
                if not (np.array_equal(de.lValid, lv) and np.array_equal(de.lDisMap, O.fill_inv(ref["ldisp"], lv))):
                    return False, " | ".join(log)
    return True, log[0]


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    O.build()
    rng = np.random.default_rng(seed)
    t0 = time.time(); n = bad = 0
    while time.time() - t0 < secs:
        try:
            ok, desc = episode(rng, n)
        except Exception as e:
            ok, desc = False, f"episode {n}: EXCEPTION {type(e).__name__}: {str(e)[:300]}"
        n += 1
        if not ok:
            bad += 1
            print("FAIL", desc, flush=True)
    print(f"soak_state: {n} episodes in {time.time() - t0:.0f} s, {bad} failures (seed {seed})")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
