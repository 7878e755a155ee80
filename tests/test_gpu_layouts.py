"""-m gpu: the fused filter's two workgroup layouts and the planner's hints (round 5).  Image widths for which the narrow layout
(one producer + one consumer wave per 50 columns) needs fewer waves per row than the wide one (2 + 2 waves per 107 columns) run
the narrow instantiations of k_cvf_pc; PSM_OPT_FRAMES_IN_FLIGHT and psm_compute_batch make the planner cut the launches as a
flow of many small items.  None of this may change a bit: the maps and the winning costs are the oracle's (src/CVF.cpp:66-147,
src/DispSel.cpp:33-77) at every width around the layout boundaries, in both select forms, in 8-bit mode, in batches, on row
stripes and disparity shards."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TWO_ON, STORE = 1048576, 8192
# widths at the edges of the ranges where 2 * ceil(W / 50) < 4 * ceil(W / 107) (narrow), and their wide neighbours
NARROW_W = [16, 49, 50, 108, 149, 150, 215, 250, 322, 350, 429, 450, 536, 550, 643, 650, 750]
WIDE_W = [51, 107, 151, 214, 251, 321, 351, 428, 451, 642, 749, 751]


@pytest.fixture(scope="module")
def psm():
    from primestereomatch_amd import capi
    capi.load()
    assert capi.device_count() >= 1, "no HIP device visible"
    import primestereomatch_amd as P
    return P


def narrow_rule(W):
    return 2 * -(-W // 50) < 4 * -(-W // 107)


def test_width_lists_match_the_rule():
    assert all(narrow_rule(w) for w in NARROW_W) and not any(narrow_rule(w) for w in WIDE_W)


@pytest.mark.parametrize("W", NARROW_W + WIDE_W)
def test_both_select_forms_at_layout_boundaries(psm, oracle, W):
    from primestereomatch_amd import capi, synth
    H, D = 37 + W % 23, min(24, max(4, W // 3))
    l, r, _ = synth.make_pair(W, H, D, seed=W)
    ref = oracle.pipeline_f32(l, r, D, threads=8, want_volumes=True)
    for flags in (0, TWO_ON):                      # chunk planes only; every 8th slice through planes, the rest against the keys
        with psm.DispEst(l, r, D) as de:
            de.set_option(capi.PSM_OPT_FLAGS, flags)
            de.CostConst_GPU(); de.CostFilter_GPU(); de.DispSelect_GPU()
            assert np.array_equal(de.lDisMap, ref["ldisp"]) and np.array_equal(de.rDisMap, ref["rdisp"]), (W, flags)
    ref8 = oracle.pipeline_u8(l, r, D, threads=8)
    with psm.DispEst(l, r, D, dtype="u8") as de:
        de.CostConst_GPU(); de.CostFilter_GPU(); de.DispSelect_GPU()
        assert np.array_equal(de.lDisMap, ref8["ldisp"]) and np.array_equal(de.rDisMap, ref8["rdisp"]), W


@pytest.mark.parametrize("W,H,D,seg", [(450, 375, 64, 0), (450, 375, 64, 41), (340, 256, 120, 0), (150, 120, 32, 0), (150, 97, 32, 9), (250, 131, 130, 33)])
def test_narrow_layout_segments_stripes_and_shards(psm, oracle, W, H, D, seg):
    from primestereomatch_amd import capi, synth
    assert narrow_rule(W)
    l, r, _ = synth.make_pair(W, H, D, seed=H)
    ref = oracle.pipeline_f32(l, r, D, threads=8)
    with psm.DispEst(l, r, D) as de:               # the planner's own cut (two phases from 112 slices) or forced short segments
        if seg:
            de.set_option(capi.PSM_OPT_SEG_ROWS, seg)
        de.CostConst_GPU(); de.CostFilter_GPU(); de.DispSelect_GPU()
        assert np.array_equal(de.lDisMap, ref["ldisp"]) and np.array_equal(de.rDisMap, ref["rdisp"])
    cut = H // 3 + 1
    parts = []
    for ya, yb in ((0, cut), (cut, H)):            # two row stripes
        de = psm.DispEst(l, r, D)
        if seg:
            de.set_option(capi.PSM_OPT_SEG_ROWS, seg)
        de.set_rows(ya, yb)
        de.CostConst_GPU(); de.CostFilter_GPU(); de.DispSelect_device()
        parts.append(de)
    parts[0].gather_rows_ctx(parts)
    assert np.array_equal(parts[0].lDisMap, ref["ldisp"]) and np.array_equal(parts[0].rDisMap, ref["rdisp"])
    for de in parts:
        de.close()
    shards = []
    for g in range(3):                             # three disparity shards
        de = psm.DispEst(l, r, D, d_range=(D * g // 3, D * (g + 1) // 3))
        de.CostConst_GPU(); de.CostFilter_GPU(); de.DispSelect_partial()
        shards.append(de)
    shards[0].DispSelect_merge_ctx(shards)
    assert np.array_equal(shards[0].lDisMap, ref["ldisp"]) and np.array_equal(shards[0].rDisMap, ref["rdisp"])
    for de in shards:
        de.close()


@pytest.mark.parametrize("W,H,D", [(450, 375, 64), (384, 288, 64), (150, 120, 32), (131, 77, 120), (640, 200, 128)])
@pytest.mark.parametrize("inflight", [2, 3, 8])
def test_frames_in_flight_hint_changes_no_result(psm, oracle, W, H, D, inflight):
    """PSM_OPT_FRAMES_IN_FLIGHT only moves the planner's cut (segments, slices per chunk): same maps, and - through the storing form,
    which the hint does not touch - the same volumes."""
    from primestereomatch_amd import capi, synth
    l, r, _ = synth.make_pair(W, H, D, seed=inflight)
    ref = oracle.pipeline_f32(l, r, D, threads=8)
    with psm.DispEst(l, r, D) as de:
        de.set_option(capi.PSM_OPT_FRAMES_IN_FLIGHT, inflight)
        for flags in (0, TWO_ON):
            de.set_option(capi.PSM_OPT_FLAGS, flags)
            de.CostConst_GPU(); de.CostFilter_GPU(); de.DispSelect_GPU()
            assert np.array_equal(de.lDisMap, ref["ldisp"]) and np.array_equal(de.rDisMap, ref["rdisp"])
        de.set_option(capi.PSM_OPT_FRAMES_IN_FLIGHT, 1)          # and back, same context (scratch re-sized for the other cut)
        de.set_option(capi.PSM_OPT_FLAGS, 0)
        de.CostConst_GPU(); de.CostFilter_GPU(); de.DispSelect_GPU()
        assert np.array_equal(de.lDisMap, ref["ldisp"]) and np.array_equal(de.rDisMap, ref["rdisp"])


def test_frames_in_flight_option_range(psm):
    from primestereomatch_amd import capi, synth
    l, r, _ = synth.make_pair(64, 48, 8, seed=0)
    with psm.DispEst(l, r, 8) as de:
        for bad in (0, -1, 65):
            with pytest.raises(capi.PsmError, match="frames in flight"):
                de.set_option(capi.PSM_OPT_FRAMES_IN_FLIGHT, bad)
        de.set_option(capi.PSM_OPT_FRAMES_IN_FLIGHT, 64)


def test_frame_ring_sets_the_hint_and_equals_single_contexts(psm, oracle):
    from primestereomatch_amd import synth
    W, H, D = 450, 120, 32
    frames = [synth.make_pair(W, H, D, seed=s)[:2] for s in range(5)]
    refs = [oracle.pipeline_f32(l, r, D, threads=8) for l, r in frames]
    ring = psm.FrameRing(frames[0][0], frames[0][1], D, frames=2)
    from primestereomatch_amd import capi
    assert all(c.options.get(capi.PSM_OPT_FRAMES_IN_FLIGHT) == 2 for c in ring.ctx)
    outs = [ring.push(l, r) for l, r in frames]
    outs = [o for o in outs if o is not None] + list(ring.flush())
    ring.close()
    assert len(outs) == 5
    for (lm, rm), ref in zip(outs, refs):
        assert np.array_equal(lm, ref["ldisp"]) and np.array_equal(rm, ref["rdisp"])


@pytest.mark.parametrize("dtype,W,H,D,B", [("f32", 150, 120, 32, 8), ("f32", 340, 100, 48, 3), ("u8", 450, 90, 32, 4), ("f32", 250, 64, 120, 2)])
def test_batches_in_the_narrow_layout_equal_the_oracle(psm, oracle, dtype, W, H, D, B):
    from primestereomatch_amd import synth
    from primestereomatch_amd.dispest import compute_batch
    pairs = [synth.make_pair(W, H, D, seed=10 + b)[:2] for b in range(B)]
    des = [psm.DispEst(l, r, D, dtype=dtype) for l, r in pairs]
    try:
        compute_batch(des)
        for de, (l, r) in zip(des, pairs):
            ref = oracle.pipeline_f32(l, r, D, threads=8) if dtype == "f32" else oracle.pipeline_u8(l, r, D, threads=8)
            lm, rm = de.download_maps()
            assert np.array_equal(lm, ref["ldisp"]) and np.array_equal(rm, ref["rdisp"])
    finally:
        for de in des:
            de.close()
