"""-m gpu: PSM_FLAG_F32_TOL - the tolerance form of the fused select kernel (level 1 of the horizontal window sums in fp32).
BASELINE.json asks for 1e-4 in float mode; the default form is bit-exact and stays the reference for parity.  The tolerance form
is pinned three ways: (1) its maps AND the winning costs it leaves in the packed keys equal its own CPU model (oracle variant
PSMO_VAR_F32_L1) bit for bit - so the model's full-volume statistics are the kernel's; (2) the model stays within 1e-4 of the
canonical oracle over whole volumes; (3) on the Middlebury pairs no disparity changes at all.  (On the synthetic pairs, whose
flat textures produce many near-ties, a few pixels in 1e4 pick the neighbouring disparity - reported, bounded, and the reason
the form is not the default.)"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 33554432


@pytest.fixture(scope="module")
def psm():
    from primestereomatch_amd import capi
    capi.load()
    assert capi.device_count() >= 1, "no HIP device visible"
    import primestereomatch_amd as P
    return P


def device_keys(de):
    """[2][H][W] int64 packed minima of the context (device -> host through the HIP runtime the library is bound to)."""
    hip = C.CDLL("libamdhip64.so")
    ptr, nbytes = de.partial_keys()
    out = np.empty((2, de.hei, de.wid), np.int64)
    de.synchronize()
    assert hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), C.c_size_t(nbytes), 2) == 0
    return out


def key_costs(keys):
    """the float cost in the high half of pack_key_f32 (psm_dev.h)"""
    hi = (keys >> 32).astype(np.int64).astype(np.int32)
    bits = np.where(hi < 0, hi ^ np.int32(0x7fffffff), hi).astype(np.int32)
    return bits.view(np.float32)


def run_tol(psm, l, r, D, flags):
    from primestereomatch_amd import capi
    with psm.DispEst(l, r, D) as de:
        de.set_option(capi.PSM_OPT_FLAGS, flags)
        de.CostConst_GPU(); de.CostFilter_GPU(); de.DispSelect_GPU()
        return de.lDisMap.copy(), de.rDisMap.copy(), device_keys(de)


@pytest.mark.parametrize("name", ["cones", "teddy", "cones_crop"])
def test_tolerance_form_on_the_middlebury_pairs(psm, oracle, golden, name):
    g = golden("cones_pair.npz" if name.startswith("cones") else "teddy_pair.npz")
    l, r = g["l_bgr"], g["r_bgr"]
    if name == "cones_crop":
        l, r = np.ascontiguousarray(l[:288, :384]), np.ascontiguousarray(r[:288, :384])
    D = 64
    canon = oracle.pipeline_f32(l, r, D, threads=8, want_volumes=True)
    with oracle.variant(oracle.VAR_F32_L1):
        model = oracle.pipeline_f32(l, r, D, threads=8, want_volumes=True)
    for flags in (TOL, TOL | 1048576):                       # single-phase (planes) and forced two-phase (planes + keys)
        lm, rm, keys = run_tol(psm, l, r, D, flags)
        assert np.array_equal(lm, model["ldisp"]) and np.array_equal(rm, model["rdisp"])          # (1) the kernel IS its model
        for s, vol, mp in ((0, model["lvol"], lm), (1, model["rvol"], rm)):
            win = np.take_along_axis(vol, mp[None].astype(np.int64), axis=0)[0]
            assert np.array_equal(key_costs(keys[s])[mp > 0], win[mp > 0])
        assert np.array_equal(lm, canon["ldisp"]) and np.array_equal(rm, canon["rdisp"])          # (3) no disparity changed
    dq = max(float(np.abs(model[k].astype(np.float64) - canon[k]).max()) for k in ("lvol", "rvol"))
    print(f"[tol] {name}: max|dq| {dq:.2e}, voxels > 1e-4: 0, WTA pixels changed: 0")
    assert dq <= 1e-4                                                                             # (2)


@pytest.mark.parametrize("W,H,D,seed", [(200, 120, 40, 1), (450, 375, 64, 3), (131, 77, 120, 5), (640, 360, 128, 4)])
def test_tolerance_form_on_synthetic_pairs(psm, oracle, W, H, D, seed):
    from primestereomatch_amd import synth
    l, r, _ = synth.make_pair(W, H, D, seed=seed)
    canon = oracle.pipeline_f32(l, r, D, threads=8, want_volumes=True)
    with oracle.variant(oracle.VAR_F32_L1):
        model = oracle.pipeline_f32(l, r, D, threads=8, want_volumes=True)
    lm, rm, keys = run_tol(psm, l, r, D, TOL)
    assert np.array_equal(lm, model["ldisp"]) and np.array_equal(rm, model["rdisp"])
    dq = max(float(np.abs(model[k].astype(np.float64) - canon[k]).max()) for k in ("lvol", "rvol"))
    flips = int(np.count_nonzero(lm != canon["ldisp"]) + np.count_nonzero(rm != canon["rdisp"]))
    print(f"[tol] synthetic {W}x{H}x{D}: max|dq| {dq:.2e}, WTA pixels changed: {flips} of {2 * W * H}")
    assert dq <= 1e-4 and flips <= 2e-4 * 2 * W * H
    # where the disparity differs, the two candidates' canonical costs are within the tolerance of each other
    for mp, cm, vol in ((lm, canon["ldisp"], canon["lvol"]), (rm, canon["rdisp"], canon["rvol"])):
        a = np.take_along_axis(vol, mp[None].astype(np.int64), axis=0)[0]
        b = np.take_along_axis(vol, cm[None].astype(np.int64), axis=0)[0]
        assert float(np.abs(a - b)[mp != cm].max(initial=0.0)) <= 2e-4


def test_default_stays_bit_exact_and_flag_is_ignored_in_8bit_mode(psm, oracle):
    from primestereomatch_amd import capi, synth
    l, r, _ = synth.make_pair(160, 100, 32, seed=2)
    ref8 = oracle.pipeline_u8(l, r, 32, threads=8)
    with psm.DispEst(l, r, 32, dtype="u8") as de:
        de.set_option(capi.PSM_OPT_FLAGS, TOL)
        de.CostConst_GPU(); de.CostFilter_GPU(); de.DispSelect_GPU()
        assert np.array_equal(de.lDisMap, ref8["ldisp"]) and np.array_equal(de.rDisMap, ref8["rdisp"])
    ref = oracle.pipeline_f32(l, r, 32, threads=8, want_volumes=True)
    with psm.DispEst(l, r, 32) as de:
        de.set_option(capi.PSM_OPT_FLAGS, TOL)
        de.CostConst_GPU(); de.CostFilter_GPU(); de.DispSelect_GPU()
        assert np.array_equal(de.download_volume(0), ref["lvol"])        # readers of the volume get the exact storing form
