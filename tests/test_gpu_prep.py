"""-m gpu: image preparation folded into the guidance launch (round 6).

psm_cost_construct leaves CVC::preprocess (src/CVC.cpp:41-46; for 8-bit images also the convertTo of src/StereoMatch.cpp:195-196) to
psm_cost_filter when the cost volume is virtual: k_guide_march then reads the staged interleaved images itself, forms the image
planes, gray and the x-gradient, writes g1 and computes the guidance in ONE launch (k_prep + k_guide_march before).  Same bits:
all 14 planes (I0, I1, I2, GrdX, means, 1/DET, adjugate) equal those of the two-kernel path and the oracle's, at strip / border
widths, for 8-bit and float images, on row stripes, and for a second pair on the same context."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def psm():
    from primestereomatch_amd import capi
    capi.load()
    assert capi.device_count() >= 1, "no HIP device visible"
    import primestereomatch_amd as P
    return P


def planes_merged(psm, l, r, D):
    with psm.DispEst(l, r, D) as de:
        de.CostConst_GPU(); de.CostFilter_GPU()              # prep + guidance in one launch
        out = [de.download_guidance(0).copy(), de.download_guidance(1).copy()]
        de.DispSelect_GPU()
        return out, de.lDisMap.copy(), de.rDisMap.copy()


def planes_two_kernels(psm, l, r, D):
    with psm.DispEst(l, r, D) as de:
        de.CostConst_GPU()
        de.download_guidance(0)                              # someone asks for g1 before the filter: k_prep runs on its own ...
        de.CostFilter_GPU()                                  # ... and the guidance launch reads g1 as it always did
        out = [de.download_guidance(0).copy(), de.download_guidance(1).copy()]
        de.DispSelect_GPU()
        return out, de.lDisMap.copy(), de.rDisMap.copy()


@pytest.mark.parametrize("W,H,D,f32", [(57, 40, 8, False), (113, 41, 12, False), (450, 375, 16, False), (640, 97, 24, True), (56, 9, 4, False),
                                       (1280, 720, 8, False), (9, 9, 4, True)])
def test_merged_launch_equals_two_kernels_and_oracle(psm, oracle, W, H, D, f32):
    from primestereomatch_amd import synth
    l, r, _ = synth.make_pair(W, H, D, seed=W + H)
    if f32:
        l, r = oracle.u8_to_f32(l), oracle.u8_to_f32(r)
    a, alm, arm = planes_merged(psm, l, r, D)
    b, blm, brm = planes_two_kernels(psm, l, r, D)
    for s in range(2):
        assert np.array_equal(a[s].view(np.uint32), b[s].view(np.uint32)), ("side", s)
    assert np.array_equal(alm, blm) and np.array_equal(arm, brm)
    for s, img in enumerate((l, r)):
        f = img if f32 else oracle.u8_to_f32(img)
        assert np.array_equal(a[s][0], f[..., 0]) and np.array_equal(a[s][1], f[..., 1]) and np.array_equal(a[s][2], f[..., 2])
        assert np.array_equal(a[s][3], oracle.cvc_preprocess(f))          # gray + Sobel(1, 0, ksize 1), REFLECT_101
        rgb, mean, var = oracle.cvf_preprocess(f)
        assert np.array_equal(a[s][4:7], mean)


def test_merged_launch_on_row_stripes_and_a_second_pair(psm, oracle):
    from primestereomatch_amd import synth
    W, H, D = 321, 160, 24
    l, r, _ = synth.make_pair(W, H, D, seed=5)
    l2, r2, _ = synth.make_pair(W, H, D, seed=6)
    ref, ref2 = oracle.pipeline_f32(l, r, D, threads=8), oracle.pipeline_f32(l2, r2, D, threads=8)
    for y0, y1 in ((0, 50), (40, 100), (97, 160), (3, 5)):
        with psm.DispEst(l, r, D) as de:
            de.set_rows(y0, y1)
            de.CostConst_GPU(); de.CostFilter_GPU(); de.DispSelect_device()
            lm, rm = de.download_maps()
            assert np.array_equal(lm[y0:y1], ref["ldisp"][y0:y1]) and np.array_equal(rm[y0:y1], ref["rdisp"][y0:y1]), (y0, y1)
            de.setInputImages(l2, r2)                        # a new pair: the planes of the old one must not survive
            de.CostConst_GPU(); de.CostFilter_GPU(); de.DispSelect_device()
            lm, rm = de.download_maps()
            assert np.array_equal(lm[y0:y1], ref2["ldisp"][y0:y1]) and np.array_equal(rm[y0:y1], ref2["rdisp"][y0:y1]), (y0, y1)
            de.set_rows(0, 0)                                # whole image after a stripe: everything is prepared again
            de.CostConst_GPU(); de.CostFilter_GPU(); de.DispSelect_GPU()
            assert np.array_equal(de.lDisMap, ref2["ldisp"]) and np.array_equal(de.rDisMap, ref2["rdisp"])


def test_a_float_frame_launches_no_preparation_kernel(psm):
    from primestereomatch_amd import capi, synth
    l, r, _ = synth.make_pair(320, 200, 48, seed=1)
    for dtype, nprep in (("f32", 0), ("u8", 3)):            # 8-bit contexts keep k_prep (+ k_prep_u8 per image: their byte planes)
        with psm.DispEst(l, r, 48, dtype=dtype) as de:
            de.CostConst_GPU(); de.CostFilter_GPU(); de.DispSelect_GPU()
            de.set_option(capi.PSM_OPT_PROFILE, 1)
            de.reset_kernel_times()
            de.CostConst_GPU(); de.CostFilter_GPU(); de.DispSelect_GPU()
            de.synchronize()
            assert de.kernel_time_ms(capi.PSM_K_PREP)[1] == nprep and de.kernel_time_ms(capi.PSM_K_GUIDE)[1] == 1, dtype
