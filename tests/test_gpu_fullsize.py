"""-m gpu: the PRODUCT path at product geometry against the CPU oracle's complete maps.

At 1280x720x128, 1920x1080x256 and 3840x2160 the default path is the two-phase select form (minima planes for every S-th
slice, the key form at four workgroups per CU for the rest, 3-4 row segments per column group); tests that download volumes
only ever see its storing sibling.  Here the two u8 maps the select path leaves behind are compared with
`oracle.pipeline_f32` on the whole image and the whole disparity range (DispSel::CVSelect, src/DispSel.cpp:83-109, over
volumes filtered per src/CVF.cpp:72-165) - unsharded, as 8 row stripes gathered with psm_gather_rows_ctx, and as 8
disparity shards merged with psm_disp_merge_ctx (BASELINE.json configs[2], [3], [4]); plus the 8-bit mode on the 384x288
Cones crop BASELINE configs[0] quotes.  The oracle runs on the box's host cores (one pthread per disparity in blocks, like
the reference): ~1 s at 720p x 128, ~7 s at 1080p x 256 on 8 threads."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

THREADS = min(32, os.cpu_count() or 8)   # oracle threads (the reference's MAX_CPU_THREADS is 8; more only shortens the test)


@pytest.fixture(scope="module")
def psm():
    from primestereomatch_amd import capi
    capi.load()
    assert capi.device_count() >= 1, "no HIP device visible"
    import primestereomatch_amd as P
    return P


_REF = {}


def oracle_maps(oracle, W, H, D, seed=0):
    key = (W, H, D, seed)
    if key not in _REF:
        from primestereomatch_amd import synth
        l, r, _ = synth.make_pair(W, H, D, seed=seed)
        ref = oracle.pipeline_f32(l, r, D, threads=THREADS if W * H < (1 << 22) else min(16, THREADS))   # (workspace per thread: 9 planes)
        _REF[key] = (l, r, ref["ldisp"].copy(), ref["rdisp"].copy())
    return _REF[key]


def mism(a, b):
    return int(np.count_nonzero(a != b))


@pytest.mark.parametrize("W,H,D", [(1280, 720, 128), (1920, 1080, 256)])
def test_default_path_maps_equal_oracle(psm, oracle, W, H, D):
    l, r, el, er = oracle_maps(oracle, W, H, D)
    with psm.DispEst(l, r, D) as de:
        for _ in range(2):      # second frame: every scratch buffer is warm, nothing stale may survive
            de.CostConst_GPU(); de.CostFilter_GPU(); de.DispSelect_GPU()
            assert (mism(de.lDisMap, el), mism(de.rDisMap, er)) == (0, 0)


@pytest.mark.parametrize("W,H,D", [(1280, 720, 128), (1920, 1080, 256)])
def test_eight_row_stripes_gathered_equal_oracle(psm, oracle, W, H, D):
    from primestereomatch_amd import stripes
    l, r, el, er = oracle_maps(oracle, W, H, D)
    ctxs = [psm.DispEst(l, r, D) for _ in range(8)]
    try:
        for g, c in enumerate(ctxs):
            _, y0, y1 = stripes.stripe_bounds(H, 8, g)
            c.set_rows(y0, y1)
            c.CostConst_GPU(); c.CostFilter_GPU(); c.DispSelect_GPU()
        ctxs[0].gather_rows_ctx(ctxs)
        assert (mism(ctxs[0].lDisMap, el), mism(ctxs[0].rDisMap, er)) == (0, 0)
    finally:
        for c in ctxs:
            c.close()


@pytest.mark.parametrize("W,H,D", [(1280, 720, 128), (1920, 1080, 256)])
def test_eight_disparity_shards_merged_equal_oracle(psm, oracle, W, H, D):
    l, r, el, er = oracle_maps(oracle, W, H, D)
    shards = [psm.DispEst(l, r, D, d_range=(D * g // 8, D * (g + 1) // 8)) for g in range(8)]
    try:
        for s in shards:
            s.CostConst_GPU(); s.CostFilter_GPU(); s.DispSelect_partial()
        shards[0].DispSelect_merge_ctx(shards)
        assert (mism(shards[0].lDisMap, el), mism(shards[0].rDisMap, er)) == (0, 0)
    finally:
        for s in shards:
            s.close()


def test_4k_two_phase_maps_equal_oracle(psm, oracle):
    """3840x2160 (BASELINE configs[4]) at D = 112: the smallest range at which the two-phase selection is the default."""
    W, H, D = 3840, 2160, 112
    l, r, el, er = oracle_maps(oracle, W, H, D, seed=2)
    with psm.DispEst(l, r, D) as de:
        de.CostConst_GPU(); de.CostFilter_GPU(); de.DispSelect_GPU()
        assert (mism(de.lDisMap, el), mism(de.rDisMap, er)) == (0, 0)
        de.LRCheck_GPU()                       # configs[4]: "+ PP left-right check on-GPU"
        lv, rv = oracle.lr_check(el, er)
        assert np.array_equal(de.lValid, lv) and np.array_equal(de.rValid, rv)


_REF4K = {}


def oracle_maps_4k(oracle):
    """3840 x 2160 x 256 (BASELINE configs[4] as written): the streaming form of the oracle (psmo_pipeline_f32_maps - same
    jobs, same arithmetic, blocks of `threads` slices folded into the running WTA instead of two 8.5 GB volumes)."""
    if not _REF4K:
        from primestereomatch_amd import synth
        W, H, D = 3840, 2160, 256
        l, r, _ = synth.make_pair(W, H, D, seed=0)
        ref = oracle.pipeline_f32_maps(l, r, D, threads=min(32, THREADS))
        _REF4K["v"] = (l, r, ref["ldisp"], ref["rdisp"])
    return _REF4K["v"]


def test_4k_full_range_maps_equal_oracle(psm, oracle):
    """configs[4] end to end: the one geometry where the minima planes leave the L2s and the planner picks seed stride 4 -
    both maps over the whole image and all 256 disparities, + the L-R check the config names, two frames."""
    W, H, D = 3840, 2160, 256
    l, r, el, er = oracle_maps_4k(oracle)
    with psm.DispEst(l, r, D) as de:
        for _ in range(2):
            de.CostConst_GPU(); de.CostFilter_GPU(); de.DispSelect_GPU()
            assert (mism(de.lDisMap, el), mism(de.rDisMap, er)) == (0, 0)
        de.LRCheck_GPU()
        lv, rv = oracle.lr_check(el, er)
        assert np.array_equal(de.lValid, lv) and np.array_equal(de.rValid, rv)


def test_4k_full_range_eight_stripes_and_eight_shards_equal_oracle(psm, oracle):
    from primestereomatch_amd import stripes
    W, H, D = 3840, 2160, 256
    l, r, el, er = oracle_maps_4k(oracle)
    ctxs = [psm.DispEst(l, r, D) for _ in range(8)]
    try:
        for g, c in enumerate(ctxs):
            _, y0, y1 = stripes.stripe_bounds(H, 8, g)
            c.set_rows(y0, y1)
            c.CostConst_GPU(); c.CostFilter_GPU(); c.DispSelect_GPU()
        ctxs[0].gather_rows_ctx(ctxs)
        assert (mism(ctxs[0].lDisMap, el), mism(ctxs[0].rDisMap, er)) == (0, 0)
    finally:
        for c in ctxs:
            c.close()
    shards = [psm.DispEst(l, r, D, d_range=(D * g // 8, D * (g + 1) // 8)) for g in range(8)]
    try:
        for s in shards:
            s.CostConst_GPU(); s.CostFilter_GPU(); s.DispSelect_partial()
        shards[0].DispSelect_merge_ctx(shards)
        assert (mism(shards[0].lDisMap, el), mism(shards[0].rDisMap, er)) == (0, 0)
    finally:
        for s in shards:
            s.close()


def test_u8_mode_on_the_384x288_cones_crop(psm, oracle, golden):
    """BASELINE configs[0]: Middlebury Cones, 384x288, D=64, 8-bit char mode - bit-exact against oracle.pipeline_u8."""
    g = golden("cones_pair.npz")
    l = np.ascontiguousarray(g["l_bgr"][:288, :384])
    r = np.ascontiguousarray(g["r_bgr"][:288, :384])
    ref = oracle.pipeline_u8(l, r, 64, threads=8, want_volumes=True)
    for flags in (0, 1048576):                 # single-phase (the default at 64 slices) and forced two-phase
        from primestereomatch_amd import capi
        with psm.DispEst(l, r, 64, dtype="u8") as de:
            de.set_option(capi.PSM_OPT_FLAGS, flags)
            de.CostConst_GPU(); de.CostFilter_GPU(); de.DispSelect_GPU()
            assert (mism(de.lDisMap, ref["ldisp"]), mism(de.rDisMap, ref["rdisp"])) == (0, 0), flags
            assert np.array_equal(de.download_volume(0), ref["lvol"]) and np.array_equal(de.download_volume(1), ref["rvol"])
