"""Generate tests/golden fixtures.  Runs in the BUILD container only (reads the data files
under /root/reference/data, which do not exist on the GPU box).  Fixtures are data: the
Middlebury input pairs / ground truth / masks the reference ships (data/Cones, data/Teddy),
re-encoded as numpy arrays in cv::imread order (BGR), plus outputs of the CPU oracle.

  python scripts/make_fixtures.py
"""
import hashlib
import json
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import psm_oracle_py as O  # noqa: E402

REF = "/root/reference/data"
OUT = os.path.join(ROOT, "tests", "golden")


def bgr(path):
    return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1])


def gray(path):
    return np.ascontiguousarray(np.asarray(Image.open(path).convert("L")))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    os.makedirs(OUT, exist_ok=True)
    O.build(force=True)
    manifest = {}
    for name in ("Cones", "Teddy"):
        d = os.path.join(REF, name)
        # loader mapping: src/StereoMatch.cpp:532-541 (im2 = left, im6 = right, disp2 = left GT)
        l, r = bgr(d + "/im2.png"), bgr(d + "/im6.png")
        gt_l, gt_r = gray(d + "/disp2.png"), gray(d + "/disp6.png")
        occl = gray(d + "/occl.png")
        occl = (occl > 0).astype(np.uint8) * 255
        np.savez_compressed(os.path.join(OUT, f"{name.lower()}_pair.npz"), l_bgr=l, r_bgr=r,
                            gt_l=gt_l, gt_r=gt_r, occl=occl)
        D = 64
        res = O.pipeline_f32(l, r, D, threads=8, want_volumes=True, want_raw=True)
        res8 = O.pipeline_u8(l, r, D, threads=8, want_volumes=True, want_raw=True)
        # full outputs are large; keep maps + a few filtered slices / crops + SHA-256 of the rest
        ys, xs = slice(100, 132), slice(200, 248)
        np.savez_compressed(
            os.path.join(OUT, f"{name.lower()}_oracle_d64.npz"),
            ldisp=res["ldisp"], rdisp=res["rdisp"], ldisp_u8mode=res8["ldisp"],
            rdisp_u8mode=res8["rdisp"],
            raw_l_d17=res["raw_l"][17],
            lvol_d17=res["lvol"][17], rvol_d17=res["rvol"][17],
            lvol_crop=res["lvol"][:, ys, xs],
            raw8_l_d17=res8["raw_l"][17], lvol8_d17=res8["lvol"][17])
        manifest[name] = {
            "shape": list(l.shape), "D": D,
            "sha256": {k: sha(res[k]) for k in ("ldisp", "rdisp", "lvol", "rvol", "raw_l", "raw_r")},
            "sha256_u8mode": {k: sha(res8[k]) for k in ("ldisp", "rdisp", "lvol", "rvol", "raw_l", "raw_r")},
        }
        bad, avg = O.eval_bad_pixels(res["ldisp"], gt_l, occl, D, 4, 4)
        manifest[name]["bad_pixels_thr4_nonocc"] = bad
        manifest[name]["avg_err"] = avg
        print(name, l.shape, "bad(thr4,nonocc)=%.2f%%" % (100.0 * bad / gt_l.size), "avg", avg)
        # Fast Guided Filter variant (CostFilter_FGF), the three subsample rates the reference sweeps
        fgf = {}
        manifest[name]["fgf"] = {}
        for s in (2, 4, 8):
            rf = O.pipeline_fgf(l, r, D, s=s, threads=8, want_volumes=True)
            fgf[f"ldisp_s{s}"], fgf[f"rdisp_s{s}"] = rf["ldisp"], rf["rdisp"]
            if s == 4:
                fgf["lvol_d17_s4"] = rf["lvol"][17]
            badf, avgf = O.eval_bad_pixels(rf["ldisp"], gt_l, occl, D, 4, 4)
            manifest[name]["fgf"][str(s)] = {"sha256": {k: sha(rf[k]) for k in ("ldisp", "rdisp", "lvol", "rvol")},
                                            "bad_pixels_thr4_nonocc": badf, "avg_err": avgf}
            print(name, "fgf s=%d bad=%.2f%%" % (s, 100.0 * badf / gt_l.size))
        np.savez_compressed(os.path.join(OUT, f"{name.lower()}_oracle_fgf.npz"), **fgf)
    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
