"""CPU-only tests of the Fast Guided Filter restatement (oracle/psm_oracle.c psmo_fgf_*; reference
src/fastguidedfilter.cpp, src/DispEst.cpp:281-296): an independent numpy statement, analytic known
answers and the committed golden vectors (tests/golden/*_oracle_fgf.npz, scripts/make_fixtures.py)."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN


def r101(k, n):
    k = np.abs(k)
    return np.where(k >= n, 2 * (n - 1) - k, k)


def np_blur(p, k):
    """cv::blur(Size(k,k)), REFLECT_101, fp64 window sums as one 2-D gather (different order)."""
    H, W = p.shape
    r = k // 2
    ys = r101(np.arange(H)[:, None] + np.arange(-r, r + 1)[None, :], H)
    xs = r101(np.arange(W)[:, None] + np.arange(-r, r + 1)[None, :], W)
    pd = p.astype(np.float64)
    return (pd[:, xs].sum(axis=2)[ys, :].sum(axis=1) * (1.0 / (k * k))).astype(np.float32)


def nn_idx(ssize, s):
    d = ssize // s
    return np.minimum(np.floor(np.arange(d) * (1.0 / (d / ssize))).astype(np.int64), ssize - 1)


def np_upsample(a, H, W):
    """bilinear resize with half-pixel centres, edge clamped (fp64; the oracle works in fp32)."""
    hs, ws = a.shape

    def axis(ssize, dsize):
        f = (np.arange(dsize) + 0.5) * (ssize / dsize) - 0.5
        i = np.floor(f).astype(np.int64)
        f = f - i
        f = np.where((i < 0) | (i >= ssize - 1), 0.0, f)
        i = np.clip(i, 0, ssize - 1)
        return i, np.minimum(i + 1, ssize - 1), f

    y0, y1, fy = axis(hs, H)
    x0, x1, fx = axis(ws, W)
    ad = a.astype(np.float64)
    rows = ad[:, x0] * (1 - fx) + ad[:, x1] * fx
    return rows[y0, :] * (1 - fy)[:, None] + rows[y1, :] * fy[:, None]


def np_fgf(img, p, s):
    H, W, _ = img.shape
    k = 2 * (8 // s) + 1
    yi, xi = nn_idx(H, s), nn_idx(W, s)
    I = [img[:, :, c][np.ix_(yi, xi)] for c in range(3)]
    ps = p[np.ix_(yi, xi)]
    mI = [np_blur(I[c], k) for c in range(3)]
    mp = np_blur(ps, k)
    S = np.empty(ps.shape + (3, 3), np.float64)
    for c in range(3):
        for cp in range(c, 3):
            S[:, :, c, cp] = S[:, :, cp, c] = np_blur(I[c] * I[cp], k) - mI[c] * mI[cp]
        S[:, :, c, c] += np.float32(1e-4)
    cov = np.stack([np_blur(I[c] * ps, k) - mI[c] * mp for c in range(3)], -1).astype(np.float64)
    a = np.linalg.solve(S, cov[..., None])[..., 0]
    b = mp - sum(a[:, :, c] * mI[c] for c in range(3))
    ma = [np_blur(a[:, :, c].astype(np.float32), k) for c in range(3)]
    mb = np_blur(b.astype(np.float32), k)
    return sum(np_upsample(ma[c], H, W) * img[:, :, c] for c in range(3)) + np_upsample(mb, H, W)


@pytest.mark.parametrize("s", [2, 4, 8])
@pytest.mark.parametrize("shape", [(48, 64), (45, 70)])
def test_fgf_vs_numpy(oracle, s, shape):
    rng = np.random.default_rng(11 + s)
    H, W = shape
    img = rng.random((H, W, 3), dtype=np.float32)
    p = (rng.random((H, W), dtype=np.float32) * 2.0).astype(np.float32)
    setup = oracle.fgf_setup(img, s)
    assert setup.shape == (12, H // s, W // s)
    yi, xi = nn_idx(H, s), nn_idx(W, s)
    for c in range(3):
        assert np.array_equal(setup[c], img[:, :, c][np.ix_(yi, xi)])          # INTER_NN subsampling
        assert np.allclose(setup[3 + c], np_blur(setup[c], 2 * (8 // s) + 1), atol=1e-7)
    q = oracle.fgf_filter(img, setup, p, s)
    qn = np_fgf(img, p, s)
    # fp32 adjugate solve vs fp64 LU: a few 1e-4 on random (well-conditioned) guidance
    assert np.allclose(q, qn, atol=2e-3), np.abs(q - qn).max()
    # a constant cost slice is a fixed point (a = 0, b = p)
    qc = oracle.fgf_filter(img, setup, np.full((H, W), 0.25, np.float32), s)
    assert np.allclose(qc, 0.25, atol=2e-5)


def test_fgf_constant_guidance_is_blur_then_upsample(oracle):
    """With a constant guidance image the covariance vanishes: a = 0 exactly, b = blur(p_small), and q is the
    bilinear upsampling of blur(blur(p_small)) - checked against the numpy statement to fp32 rounding."""
    H, W, s = 40, 56, 4
    img = np.full((H, W, 3), 0.5, np.float32)
    rng = np.random.default_rng(5)
    p = rng.random((H, W), dtype=np.float32)
    q = oracle.fgf_filter(img, oracle.fgf_setup(img, s), p, s)
    ps = p[np.ix_(nn_idx(H, s), nn_idx(W, s))]
    ref = np_upsample(np_blur(np_blur(ps, 5), 5), H, W)
    assert np.allclose(q, ref, atol=5e-6)


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("name", ["cones", "teddy"])
def test_fgf_golden(oracle, golden, name):
    pair = golden(f"{name}_pair.npz")
    gold = golden(f"{name}_oracle_fgf.npz")
    man = json.load(open(os.path.join(GOLDEN, "manifest.json")))[name.capitalize()]["fgf"]
    for s in (2, 4, 8):
        res = oracle.pipeline_fgf(pair["l_bgr"], pair["r_bgr"], 64, s=s, threads=8, want_volumes=True)
        assert np.array_equal(res["ldisp"], gold[f"ldisp_s{s}"]) and np.array_equal(res["rdisp"], gold[f"rdisp_s{s}"])
        for k in ("ldisp", "rdisp", "lvol", "rvol"):
            assert _sha(res[k]) == man[str(s)]["sha256"][k], (s, k)
        bad, _ = oracle.eval_bad_pixels(res["ldisp"], pair["gt_l"], pair["occl"], 64, 4, 4)
        assert bad == man[str(s)]["bad_pixels_thr4_nonocc"]
        if s == 4:
            assert np.array_equal(res["lvol"][17], gold["lvol_d17_s4"])
    # quality ordering the reference's subsample sweep would show: finer subsampling -> fewer bad pixels
    assert man["2"]["bad_pixels_thr4_nonocc"] < man["4"]["bad_pixels_thr4_nonocc"] < man["8"]["bad_pixels_thr4_nonocc"]


def test_fgf_threads_invariant_and_bad_rate(oracle):
    from primestereomatch_amd import synth
    l, r, _ = synth.make_pair(96, 64, 16, seed=9)
    a = oracle.pipeline_fgf(l, r, 16, s=4, threads=1, want_volumes=True)
    b = oracle.pipeline_fgf(l, r, 16, s=4, threads=8, want_volumes=True)
    assert np.array_equal(a["lvol"], b["lvol"]) and np.array_equal(a["rdisp"], b["rdisp"])
    with pytest.raises(ValueError):
        oracle.pipeline_fgf(l, r, 16, s=3)
