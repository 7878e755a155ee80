"""Timing of psm_wgt_median on synthetic validity patterns (debug aid)."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import primestereomatch_amd as P

FLAGS = int(os.environ.get("WM_FLAGS", "0"))
def run(name, H, W, D, lv):
    rng = np.random.default_rng(1)
    base = rng.integers(0, 256, (H // 8 + 1, W // 8 + 1, 3))
    img = np.clip(np.kron(base, np.ones((8, 8, 1)))[:H, :W] + rng.integers(-3, 4, (H, W, 3)), 0, 255).astype(np.uint8)
    lm = rng.integers(1, D, (H, W)).astype(np.uint8)
    with P.DispEst(img, img, D) as de:
        de.set_option(P.capi.PSM_OPT_FLAGS, FLAGS)
        for rep in range(2):
            de.upload_maps(lm, lm, lv, lv)
            t0 = time.perf_counter()
            de.WgtMedian_GPU()
            dt = time.perf_counter() - t0
        n = int((lv == 0).sum())
        print(de.wgt_median_stats(), end=" ")
        print(f"{name:28s} {W}x{H} D={D}: {n} invalid/side, both sides {dt*1e3:8.2f} ms -> {dt*1e6/(2*max(n,1)):7.2f} us per filtered pixel")

H, W, D = 375, 450, 64
rng = np.random.default_rng(0)
if len(sys.argv) > 1 and sys.argv[1] == "hd20":      # one case, for a kernel trace
    H, W, D = 1080, 1920, 256
    run("1080p iid 20%", H, W, D, (rng.random((H, W)) > 0.2).astype(np.uint8))
    sys.exit(0)
run("none invalid", H, W, D, np.ones((H, W), np.uint8))
run("iid 5%", H, W, D, (rng.random((H, W)) > 0.05).astype(np.uint8))
run("iid 43%", H, W, D, (rng.random((H, W)) > 0.43).astype(np.uint8))
v = np.ones((H, W), np.uint8); v[:, 100:140] = 0
run("one band of 40 columns", H, W, D, v)
v = np.ones((H, W), np.uint8); v[100:140, :] = 0
run("40 full rows", H, W, D, v)
v = np.ones((H, W), np.uint8); v[5, :] = 0
run("one full row", H, W, D, v)
v = np.ones((H, W), np.uint8); v[:, 7] = 0
run("one full column", H, W, D, v)
run("all invalid", H, W, D, np.zeros((H, W), np.uint8))
if len(sys.argv) > 1:
    H, W, D = 1080, 1920, 256
    run("1080p iid 20%", H, W, D, (rng.random((H, W)) > 0.2).astype(np.uint8))
    run("1080p iid 43%", H, W, D, (rng.random((H, W)) > 0.43).astype(np.uint8))
    v = np.ones((H, W), np.uint8); v[:, 300:420] = 0; v[:, 900:960] = 0; v[200:260, :] = 0
    run("1080p bands", H, W, D, v)
