"""-m gpu: psm_compute_batch - several stereo pairs of one geometry through shared launches (the reference's loop over pairs,
src/main.cpp:64-73, src/StereoMatch.cpp:556-607).  Every pair's maps must be the bits of its own single-pair run, the oracle's
and - for the Middlebury pairs the reference ships - the committed goldens; every context must afterwards behave as after
CostConst_GPU + CostFilter_GPU + DispSelect_GPU (volumes, post-processing, shards)."""
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


def test_cones_and_teddy_in_one_batch_equal_goldens(psm, oracle, golden):
    """BASELINE configs[1] (and the float form of configs[0]): Cones and Teddy, 450 x 375, D = 64, as ONE batch."""
    from primestereomatch_amd.dispest import compute_batch
    pairs = [golden("cones_pair.npz"), golden("teddy_pair.npz")]
    gold = [golden("cones_oracle_d64.npz"), golden("teddy_oracle_d64.npz")]
    des = [psm.DispEst(p["l_bgr"], p["r_bgr"], 64) for p in pairs]
    try:
        for _ in range(2):                                   # second frame: warm scratch, cached table
            compute_batch(des)
            for de, g, p in zip(des, gold, pairs):
                lm, rm = de.download_maps()
                assert np.array_equal(lm, g["ldisp"]) and np.array_equal(rm, g["rdisp"])
        # a batched context is a complete context: the filtered volume (storing form re-run) and the L-R check
        ref = oracle.pipeline_f32(pairs[1]["l_bgr"], pairs[1]["r_bgr"], 64, threads=8, want_volumes=True)
        assert np.array_equal(des[1].download_volume(0, 30, 34), ref["lvol"][30:34])
        des[0].LRCheck_GPU()
        lv, rv = oracle.lr_check(gold[0]["ldisp"], gold[0]["rdisp"])
        assert np.array_equal(des[0].lValid, lv) and np.array_equal(des[0].rValid, rv)
    finally:
        for de in des:
            de.close()


@pytest.mark.parametrize("dtype,W,H,D,B,flags", [("f32", 450, 375, 64, 8, 0), ("u8", 384, 288, 64, 5, 0), ("f32", 200, 120, 40, 3, 1048576),
                                                 ("u8", 150, 90, 24, 4, 1048576), ("f32", 131, 77, 120, 2, 0), ("f32", 107, 64, 7, 9, 2097152)])
def test_batch_equals_single_pair_runs_and_oracle(psm, oracle, dtype, W, H, D, B, flags):
    """B different pairs per launch, both modes, single-phase and (forced / default at D >= 112) two-phase: maps == the
    single-pair API's == the oracle's."""
    from primestereomatch_amd import capi, synth
    from primestereomatch_amd.dispest import compute_batch
    pairs = [synth.make_pair(W, H, D, seed=10 + b)[:2] for b in range(B)]
    des = [psm.DispEst(l, r, D, dtype=dtype) for l, r in pairs]
    try:
        for de in des:
            de.set_option(capi.PSM_OPT_FLAGS, flags)
        compute_batch(des)
        got = [tuple(m.copy() for m in de.download_maps()) for de in des]
        for b, (l, r) in enumerate(pairs):
            ref = (oracle.pipeline_u8 if dtype == "u8" else oracle.pipeline_f32)(l, r, D, threads=8)
            assert np.array_equal(got[b][0], ref["ldisp"]) and np.array_equal(got[b][1], ref["rdisp"]), b
        with psm.DispEst(*pairs[B - 1], D, dtype=dtype) as one:
            one.set_option(capi.PSM_OPT_FLAGS, flags)
            one.CostConst_GPU(); one.CostFilter_GPU(); one.DispSelect_GPU()
            assert np.array_equal(one.lDisMap, got[B - 1][0]) and np.array_equal(one.rDisMap, got[B - 1][1])
        # a sub-batch and a new pair in one of the contexts: the pointer table follows
        l2, r2, _ = synth.make_pair(W, H, D, seed=99)
        des[0].setInputImages(l2, r2)
        compute_batch(des[:2])
        ref = (oracle.pipeline_u8 if dtype == "u8" else oracle.pipeline_f32)(l2, r2, D, threads=8)
        lm, rm = des[0].download_maps()
        assert np.array_equal(lm, ref["ldisp"]) and np.array_equal(rm, ref["rdisp"])
        lm, rm = des[1].download_maps()
        assert np.array_equal(lm, got[1][0]) and np.array_equal(rm, got[1][1])
    finally:
        for de in des:
            de.close()


def test_batch_of_disparity_shards_and_async_frames(psm, oracle):
    """Shards batch as well (their packed minima merge as usual), and the frame-loop entries work per context: the next
    pair staged asynchronously is adopted by the batch call, the maps return through the asynchronous download."""
    from primestereomatch_amd import synth
    from primestereomatch_amd.dispest import compute_batch
    W, H, D = 160, 100, 48
    l, r, _ = synth.make_pair(W, H, D, seed=5)
    ref = oracle.pipeline_f32(l, r, D, threads=8)
    shards = [psm.DispEst(l, r, D, d_range=(0, 24)), psm.DispEst(l, r, D, d_range=(24, 48))]
    try:
        compute_batch(shards[:1]); compute_batch(shards[1:])
        shards[0].DispSelect_merge_ctx(shards)
        assert np.array_equal(shards[0].lDisMap, ref["ldisp"]) and np.array_equal(shards[0].rDisMap, ref["rdisp"])
    finally:
        for s in shards:
            s.close()
    pairs = [synth.make_pair(W, H, D, seed=20 + b)[:2] for b in range(6)]
    des = [psm.DispEst(*pairs[b], D) for b in range(3)]
    try:
        from primestereomatch_amd.dispest import share_streams
        share_streams(des)                                      # one compute stream, one copy stream each way for the batch
        with pytest.raises(psm.capi.PsmError):
            share_streams(des[:2])                              # (once per context)
        compute_batch(des)
        for b in range(3):
            des[b].setInputImages_async(*pairs[3 + b])          # the next frame's pairs travel ...
            des[b].download_maps_async()                        # ... while this frame's maps return
        compute_batch(des)                                      # adopts the staged pairs
        for b in range(3):
            lm, rm = (m.copy() for m in des[b].download_maps_wait())
            e = oracle.pipeline_f32(*pairs[b], D, threads=8)
            assert np.array_equal(lm, e["ldisp"]) and np.array_equal(rm, e["rdisp"]), b
            lm, rm = des[b].download_maps()
            e = oracle.pipeline_f32(*pairs[3 + b], D, threads=8)
            assert np.array_equal(lm, e["ldisp"]) and np.array_equal(rm, e["rdisp"]), b
    finally:
        for de in des:
            de.close()


def test_batch_refuses_what_it_cannot_run(psm):
    from primestereomatch_amd import capi, synth
    from primestereomatch_amd.dispest import compute_batch
    l, r, _ = synth.make_pair(96, 64, 16, seed=1)
    a, b, c = psm.DispEst(l, r, 16), psm.DispEst(l[:, :88].copy(), r[:, :88].copy(), 16), psm.DispEst(l, r, 16)
    try:
        with pytest.raises(capi.PsmError):
            compute_batch([a, b])                               # another geometry
        c.set_option(capi.PSM_OPT_FLAGS, capi.PSM_FLAG_STORE_FILTERED)
        with pytest.raises(capi.PsmError):
            compute_batch([a, c])                               # storing form
        c.set_option(capi.PSM_OPT_FLAGS, 0)
        c.set_rows(0, 32)
        with pytest.raises(capi.PsmError):
            compute_batch([a, c])                               # row stripe
        with pytest.raises(capi.PsmError):
            compute_batch([a, a])
    finally:
        for d in (a, b, c):
            d.close()


def test_share_streams_rejects_duplicates_and_empty_lists_are_noops(psm):
    """(advisor, round 4) a context listed twice made psm_share_streams destroy the set's own upload stream on the second visit."""
    from primestereomatch_amd import synth
    from primestereomatch_amd.dispest import compute_batch, share_streams
    l, r, _ = synth.make_pair(128, 64, 16, seed=1)
    des = [psm.DispEst(l, r, 16) for _ in range(2)]
    try:
        share_streams([])                      # nothing to do, no IndexError
        compute_batch([])
        with pytest.raises(psm.capi.PsmError, match="appears twice"):
            share_streams([des[0], des[1], des[0]])
        share_streams(des)                     # the refused call left the contexts untouched
        for de in des:
            de.setInputImages_async(l, r)      # (the shared upload stream is alive)
        compute_batch(des)
        assert np.array_equal(des[0].download_maps()[0], des[1].download_maps()[0])
    finally:
        for de in des:
            de.close()


def test_batch_after_an_out_of_range_volume_upload(psm, oracle):
    """(advisor, round 4) an earlier out-of-range psm_upload_volume left vol_domain_ok false; the batch rebuilds the costs from the
    images (as psm_cost_construct does) and must neither be refused with a message about float images nor leave the flag stale."""
    from primestereomatch_amd import capi, synth
    from primestereomatch_amd.dispest import compute_batch
    W, H, D = 128, 64, 16
    l, r, _ = synth.make_pair(W, H, D, seed=3)
    ref = oracle.pipeline_f32(l, r, D, threads=8)
    with psm.DispEst(l, r, D) as de:
        de.CostConst_GPU()
        de.upload_volume(0, np.full((D, H, W), 2.0 ** 100, np.float32))      # outside 2^-60 .. 2^60: the storing form's domain
        compute_batch([de])
        lm, rm = de.download_maps()
        assert np.array_equal(lm, ref["ldisp"]) and np.array_equal(rm, ref["rdisp"])
        de.set_option(capi.PSM_OPT_PROFILE, 2)
        de.CostConst_GPU(); de.CostFilter_GPU(); de.DispSelect_GPU()          # the single-pair path is back on its select forms
        forms = [f for _, f in de.filter_launch_times()]
        assert forms and 0 not in forms, forms
        assert np.array_equal(de.lDisMap, ref["ldisp"])

