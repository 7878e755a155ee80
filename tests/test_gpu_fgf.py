"""-m gpu parity tests of the Fast Guided Filter variant (psm_cost_filter_fgf / DispEst.CostFilter_FGF_GPU,
reference DispEst::CostFilter_FGF src/DispEst.cpp:281-296 + src/fastguidedfilter.cpp) against the CPU oracle
(oracle psmo_fgf_*) and the committed golden vectors.  Bar: filtered volumes within 1e-4 (asserted), and -
because the kernels evaluate the oracle's canonical arithmetic op for op - bit-identical (asserted too)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope="module")
def psm():
    from primestereomatch_amd import capi
    capi.load()
    assert capi.device_count() >= 1, "no HIP device visible"
    import primestereomatch_amd as P
    return P


def _run(psm, l, r, D, s, **kw):
    with psm.DispEst(l, r, D, **kw) as de:
        de.setSubsampleRate(s)
        de.CostConst_GPU()
        de.CostFilter_FGF_GPU()
        de.DispSelect_GPU()
        return de.download_volume(0), de.download_volume(1), de.lDisMap.copy(), de.rDisMap.copy()


@pytest.mark.parametrize("s", [2, 4, 8])
@pytest.mark.parametrize("H,W,D", [(48, 64, 9), (45, 70, 6), (37, 61, 5), (120, 161, 20)])
def test_fgf_pipeline_bit_exact(psm, oracle, s, H, W, D):
    from primestereomatch_amd import synth
    l, r, _ = synth.make_pair(W, H, D, seed=H + s)
    l[H // 4:H // 2, W // 4:W // 2] = 77   # flat patch: ill-conditioned covariance
    r[H // 4:H // 2, W // 4:W // 2] = 77
    ref = oracle.pipeline_fgf(l, r, D, s=s, threads=4, want_volumes=True)
    lv, rv, ld, rd = _run(psm, l, r, D, s)
    dl = np.abs(lv.astype(np.float64) - ref["lvol"]).max()
    dr = np.abs(rv.astype(np.float64) - ref["rvol"]).max()
    print(f"[parity] fgf s={s} {W}x{H}x{D}: max|d| L={dl:.3e} R={dr:.3e}  bit-mismatch={np.mean(lv != ref['lvol']):.3e}")
    assert dl <= TOL and dr <= TOL
    assert np.array_equal(lv, ref["lvol"]) and np.array_equal(rv, ref["rvol"])
    assert np.array_equal(ld, ref["ldisp"]) and np.array_equal(rd, ref["rdisp"])


@pytest.mark.parametrize("name", ["cones", "teddy"])
def test_fgf_middlebury_golden(psm, golden, name):
    """Against the committed fixtures only (no oracle run): maps for s = 2, 4, 8, one filtered slice for s = 4."""
    import hashlib
    pair, gold = golden(f"{name}_pair.npz"), golden(f"{name}_oracle_fgf.npz")
    man = json.load(open(os.path.join(GOLDEN, "manifest.json")))[name.capitalize()]["fgf"]
    for s in (2, 4, 8):
        lv, rv, ld, rd = _run(psm, pair["l_bgr"], pair["r_bgr"], 64, s)
        assert np.array_equal(ld, gold[f"ldisp_s{s}"]) and np.array_equal(rd, gold[f"rdisp_s{s}"])
        assert hashlib.sha256(lv.tobytes()).hexdigest() == man[str(s)]["sha256"]["lvol"]
        assert hashlib.sha256(rv.tobytes()).hexdigest() == man[str(s)]["sha256"]["rvol"]
        if s == 4:
            assert np.array_equal(lv[17], gold["lvol_d17_s4"])


@pytest.mark.parametrize("W", [64, 70])
def test_fgf_virtual_volume_equals_materialised(psm, oracle, W):
    """After CostFilter_FGF_GPU the filtered volume stays virtual (smoothed low-resolution models) and the WTA
    consumes it directly; PSM_OPT_FLAGS=4096 writes it out instead.  Same maps, same volume, any call order
    (W = 70: no 16-byte rows, the volume is always written)."""
    from primestereomatch_amd import capi, synth
    H, D = 48, 11
    l, r, _ = synth.make_pair(W, H, D, seed=2)
    ref = oracle.pipeline_fgf(l, r, D, s=4, threads=4, want_volumes=True)
    for flags, order in ((0, "select-first"), (0, "download-first"), (4096, "select-first")):
        with psm.DispEst(l, r, D) as de:
            de.set_option(capi.PSM_OPT_FLAGS, flags)
            de.CostConst_GPU()
            de.CostFilter_FGF_GPU()
            if order == "download-first":
                assert np.array_equal(de.download_volume(1), ref["rvol"])      # materialises the right volume only
            de.DispSelect_GPU()
            assert np.array_equal(de.lDisMap, ref["ldisp"]) and np.array_equal(de.rDisMap, ref["rdisp"]), (flags, order)
            de.DispSelect_GPU()                                               # a second selection of a virtual volume
            assert np.array_equal(de.lDisMap, ref["ldisp"])
            assert np.array_equal(de.download_volume(0), ref["lvol"]) and np.array_equal(de.download_volume(1), ref["rvol"])
            de.CostFilter_GPU()                                               # the full filter on top of the FGF result
            de.DispSelect_GPU()
            assert de.lDisMap.shape == (H, W)


def test_fgf_applied_twice(psm, oracle):
    """A second psm_cost_filter_fgf filters the (virtual) result of the first one: it is materialised first."""
    from primestereomatch_amd import synth
    H, W, D, s = 40, 56, 3, 4
    l, r, _ = synth.make_pair(W, H, D, 8)
    lf = oracle.u8_to_f32(l)
    setup = oracle.fgf_setup(lf, s)
    once = oracle.pipeline_fgf(l, r, D, s=s, threads=2, want_volumes=True)["lvol"]
    twice = np.stack([oracle.fgf_filter(lf, setup, once[d], s) for d in range(D)])
    with psm.DispEst(l, r, D) as de:
        de.CostConst_GPU()
        de.CostFilter_FGF_GPU()
        de.CostFilter_FGF_GPU()
        assert np.array_equal(de.download_volume(0), twice)
        de.DispSelect_GPU()
        assert np.array_equal(de.lDisMap, oracle.wta(twice))


def test_fgf_harness_metric(psm, golden):
    """Headless StereoMatch::compute with the snapshot's live branch (CostFilter_FGF, subsample_rate 4)."""
    from primestereomatch_amd import harness
    pair = golden("cones_pair.npz")
    man = json.load(open(os.path.join(GOLDEN, "manifest.json")))["Cones"]["fgf"]["4"]
    out = harness.compute(pair["l_bgr"], pair["r_bgr"], 64, gt=pair["gt_l"], mask=pair["occl"], scale_factor=4,
                          subsample_rate=4)
    assert out["bad_pixels"] == man["bad_pixels_thr4_nonocc"]
    assert out["cvf_ms"] > 0


def test_fgf_sharded_equals_unsharded(psm, oracle):
    """The FGF filter is per slice too: D-shards filter their slices independently and merge exactly."""
    from primestereomatch_amd import synth
    H, W, D = 70, 120, 14
    l, r, _ = synth.make_pair(W, H, D, 6)
    ref = oracle.pipeline_fgf(l, r, D, s=4, threads=4)
    shards = [psm.DispEst(l, r, D, d_range=(0, 5)), psm.DispEst(l, r, D, d_range=(5, 14))]
    try:
        for sh in shards:
            sh.CostConst_GPU()
            sh.CostFilter_FGF_GPU()
            sh.DispSelect_partial()
        shards[0].DispSelect_merge_ctx(shards)
        assert np.array_equal(shards[0].lDisMap, ref["ldisp"]) and np.array_equal(shards[0].rDisMap, ref["rdisp"])
    finally:
        for sh in shards:
            sh.close()


def test_fgf_on_uploaded_volume_and_errors(psm, oracle):
    """psm_upload_volume -> psm_cost_filter_fgf filters arbitrary slices; argument checks fail loudly."""
    from primestereomatch_amd import synth
    H, W, D, s = 40, 56, 4, 4
    l, r, _ = synth.make_pair(W, H, D, 3)
    vol = synth.random_volume(D, H, W, seed=1)
    lf = oracle.u8_to_f32(l)
    setup = oracle.fgf_setup(lf, s)
    with psm.DispEst(l, r, D) as de:
        de.upload_volume(0, vol)
        de.upload_volume(1, vol)
        de.CostFilter_FGF_GPU()
        got = de.download_volume(0)
        for d in range(D):
            assert np.array_equal(got[d], oracle.fgf_filter(lf, setup, vol[d], s)), d
        de.setSubsampleRate(3)
        with pytest.raises(RuntimeError, match="subsample_rate"):
            de.CostFilter_FGF_GPU()
    with psm.DispEst(l, r, D, dtype="u8") as de8:
        de8.CostConst_GPU()
        with pytest.raises(RuntimeError, match="float contexts"):
            de8.CostFilter_FGF_GPU()
    small_l, small_r, _ = synth.make_pair(16, 8, 2, 0)
    with psm.DispEst(small_l, small_r, 2) as des:
        des.setSubsampleRate(2)          # 8/2 = 4 rows of small image <= radius 4
        des.CostConst_GPU()
        with pytest.raises(RuntimeError, match="too small"):
            des.CostFilter_FGF_GPU()


def test_fgf_cpp_demo(psm, golden, tmp_path):
    import subprocess
    from conftest import ROOT
    demo = os.path.join(ROOT, "primestereomatch_amd", "lib", "psm_demo")
    pair, gold = golden("teddy_pair.npz"), golden("teddy_oracle_fgf.npz")
    H, W, _ = pair["l_bgr"].shape
    pair["l_bgr"].tofile(tmp_path / "l.raw")
    pair["r_bgr"].tofile(tmp_path / "r.raw")
    env = dict(os.environ, PRIMESM_HIP_LIB=psm.capi.LIB_PATH)
    p = subprocess.run([demo, str(tmp_path / "l.raw"), str(tmp_path / "r.raw"), str(W), str(H), "64",
                        str(tmp_path / "o"), "2", "f32", "1", "4"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    ld = np.fromfile(tmp_path / "o_ldisp.raw", np.uint8).reshape(H, W)
    rd = np.fromfile(tmp_path / "o_rdisp.raw", np.uint8).reshape(H, W)
    assert np.array_equal(ld, gold["ldisp_s4"]) and np.array_equal(rd, gold["rdisp_s4"])


def test_fgf_full_size_properties(psm):
    """1920x1080x64 (BASELINE full-HD frame, a D the test finishes quickly at): a constant cost volume is a fixed
    point of the filter to fp32 rounding, and shard-wise filtering equals whole-volume filtering bit for bit."""
    from primestereomatch_amd import synth
    W, H, D = 1920, 1080, 32
    l, r, _ = synth.make_pair(W, H, D, 1)
    with psm.DispEst(l, r, D) as de:
        de.CostConst_GPU()
        de.CostFilter_FGF_GPU()
        whole = de.download_volume(0, 8, 12)
        de.DispSelect_GPU()
        lmap = de.lDisMap.copy()
        assert np.isfinite(whole).all()
    with psm.DispEst(l, r, D, d_range=(8, 12)) as sh:
        sh.CostConst_GPU()
        sh.CostFilter_FGF_GPU()
        assert np.array_equal(sh.download_volume(0), whole)
    assert lmap.min() >= 1 and lmap.max() <= D - 1
    const = np.full((2, H, W), 0.375, np.float32)
    with psm.DispEst(l, r, 2) as dc:
        dc.upload_volume(0, const)
        dc.upload_volume(1, const)
        dc.CostFilter_FGF_GPU()
        assert np.abs(dc.download_volume(0) - 0.375).max() < 5e-5
