"""-m gpu randomized sweep: the whole path (CVC -> CVF -> WTA -> L-R check -> fill) and the Fast Guided Filter
variant against the CPU oracle on seeded random geometries - widths around the workgroup widths of the fused
kernels (96, 224 columns), heights around the batch / segment sizes, single slices, shards cut at random places.
Everything is expected bit-identical."""
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


def _geometries(n, seed):
    rng = np.random.default_rng(seed)
    edge_w = [8, 9, 12, 95, 96, 97, 100, 103, 104, 191, 192, 193, 200, 223, 224, 225, 228, 288, 300]
    edge_h = [8, 9, 10, 11, 12, 15, 16, 17, 19, 23, 31, 33, 64, 71]
    out = []
    for i in range(n):
        W = int(rng.choice(edge_w)) if rng.random() < 0.6 else int(rng.integers(8, 320))
        H = int(rng.choice(edge_h)) if rng.random() < 0.6 else int(rng.integers(8, 90))
        D = int(rng.integers(1, min(W, 48) + 1))
        out.append((W, H, D, int(rng.integers(0, 1 << 30))))
    return out


@pytest.mark.parametrize("W,H,D,seed", _geometries(36, 20260926))
def test_random_geometry_full_path(psm, oracle, W, H, D, seed):
    from primestereomatch_amd import capi, synth
    rng = np.random.default_rng(seed)
    l, r, _ = synth.make_pair(W, H, D, seed=seed & 0xffff)
    if rng.random() < 0.3:   # flat regions: ill-conditioned covariance, ties in the WTA
        l[: H // 2, : W // 2] = 90
        r[: H // 2, : W // 2] = 90
    ref = oracle.pipeline_f32(l, r, D, threads=4, want_volumes=True)
    flags = int(rng.choice([0, 0, 0, 8192, 128, 8192 + 128]))
    with psm.DispEst(l, r, D) as de:
        de.set_option(capi.PSM_OPT_FLAGS, flags)
        if rng.random() < 0.5:
            de.set_option(capi.PSM_OPT_SEG_ROWS, int(rng.integers(8, 40)))
        de.CostConst_GPU()
        de.CostFilter_GPU()
        de.DispSelect_GPU()
        assert np.array_equal(de.download_volume(0), ref["lvol"]), (W, H, D, flags)
        assert np.array_equal(de.download_volume(1), ref["rvol"]), (W, H, D, flags)
        assert np.array_equal(de.lDisMap, ref["ldisp"]) and np.array_equal(de.rDisMap, ref["rdisp"])
        de.LRCheck_GPU()
        lv, rv = oracle.lr_check(ref["ldisp"], ref["rdisp"])
        assert np.array_equal(de.lValid, lv) and np.array_equal(de.rValid, rv)
        de.FillInv_GPU()
        assert np.array_equal(de.lDisMap, oracle.fill_inv(ref["ldisp"], lv))
    if D >= 2:   # the same pair as two shards cut at a random slice
        cut = int(rng.integers(1, D))
        shards = [psm.DispEst(l, r, D, d_range=(0, cut)), psm.DispEst(l, r, D, d_range=(cut, D))]
        try:
            for sh in shards:
                sh.set_option(capi.PSM_OPT_FLAGS, flags)
                sh.CostConst_GPU(); sh.CostFilter_GPU(); sh.DispSelect_partial()
            shards[0].DispSelect_merge_ctx(shards)
            assert np.array_equal(shards[0].lDisMap, ref["ldisp"]) and np.array_equal(shards[0].rDisMap, ref["rdisp"])
        finally:
            for sh in shards:
                sh.close()


@pytest.mark.parametrize("W,H,D,seed", _geometries(18, 777))
def test_random_geometry_fgf(psm, oracle, W, H, D, seed):
    from primestereomatch_amd import capi, synth
    rng = np.random.default_rng(seed)
    W, H = max(W, 24), max(H, 24)   # the oracle wants a full blur window inside the subsampled image
    rates = [s for s in (2, 4, 8) if W // s >= 2 * (8 // s) + 1 and H // s >= 2 * (8 // s) + 1]
    s = int(rng.choice(rates))
    D = min(D, W)
    l, r, _ = synth.make_pair(W, H, D, seed=seed & 0xffff)
    ref = oracle.pipeline_fgf(l, r, D, s=s, threads=4, want_volumes=True)
    with psm.DispEst(l, r, D) as de:
        de.set_option(capi.PSM_OPT_FLAGS, int(rng.choice([0, 0, 4096, 128])))
        de.setSubsampleRate(s)
        de.CostConst_GPU()
        de.CostFilter_FGF_GPU()
        de.DispSelect_GPU()
        assert np.array_equal(de.lDisMap, ref["ldisp"]) and np.array_equal(de.rDisMap, ref["rdisp"]), (W, H, D, s)
        assert np.array_equal(de.download_volume(0), ref["lvol"]) and np.array_equal(de.download_volume(1), ref["rvol"])


@pytest.mark.parametrize("W,H,D,seed", _geometries(12, 4242))
def test_random_geometry_u8_mode(psm, oracle, W, H, D, seed):
    """8-bit char mode (build-defined contract, oracle/psm_oracle.h): integer work, bit-exact."""
    from primestereomatch_amd import synth
    l, r, _ = synth.make_pair(W, H, D, seed=seed & 0xffff)
    ref = oracle.pipeline_u8(l, r, D, threads=4, want_volumes=True, want_raw=True)
    with psm.DispEst(l, r, D, dtype="u8") as de:
        de.CostConst_GPU()
        assert np.array_equal(de.download_volume(0), ref["raw_l"]) and np.array_equal(de.download_volume(1), ref["raw_r"])
        de.CostFilter_GPU()
        de.DispSelect_GPU()
        assert np.array_equal(de.download_volume(0), ref["lvol"]) and np.array_equal(de.download_volume(1), ref["rvol"])
        assert np.array_equal(de.lDisMap, ref["ldisp"]) and np.array_equal(de.rDisMap, ref["rdisp"])


@pytest.mark.parametrize("W,H,D,seed", _geometries(24, 424242))
def test_random_geometry_two_phase_and_stripes(psm, oracle, W, H, D, seed):
    """Forced two-phase selection (planes for every 8th slice, keys for the rest) on random geometries, whole image and as
    row stripes cut at random rows (psm_set_rows / psm_gather_rows_ctx); float and 8-bit mode.  Maps bit-identical."""
    from primestereomatch_amd import capi, synth
    rng = np.random.default_rng(seed)
    D = max(D, 2)
    dtype = "u8" if rng.random() < 0.3 else "f32"
    l, r, _ = synth.make_pair(W, H, D, seed=seed & 0xffff)
    if rng.random() < 0.3:
        l[H // 3:, W // 3:] = 200
        r[H // 3:, W // 3:] = 200
    ref = (oracle.pipeline_u8 if dtype == "u8" else oracle.pipeline_f32)(l, r, D, threads=4)
    flags = 1048576 | int(rng.choice([0, 0, 128 if dtype == "f32" else 0]))
    with psm.DispEst(l, r, D, dtype=dtype) as de:
        de.set_option(capi.PSM_OPT_FLAGS, flags)
        if rng.random() < 0.5:
            de.set_option(capi.PSM_OPT_SEG_ROWS, int(rng.integers(8, 40)))
        de.CostConst_GPU(); de.CostFilter_GPU(); de.DispSelect_GPU()
        assert np.array_equal(de.lDisMap, ref["ldisp"]) and np.array_equal(de.rDisMap, ref["rdisp"]), (W, H, D, dtype, flags)
    cuts = sorted(set([0, H] + [int(c) for c in rng.integers(1, H, size=int(rng.integers(1, 4)))]))
    ctxs = [psm.DispEst(l, r, D, dtype=dtype) for _ in range(len(cuts) - 1)]
    try:
        for c, y0, y1 in zip(ctxs, cuts[:-1], cuts[1:]):
            c.set_option(capi.PSM_OPT_FLAGS, 1048576 if rng.random() < 0.5 else 0)
            c.set_rows(y0, y1)
            c.CostConst_GPU(); c.CostFilter_GPU(); c.DispSelect_GPU()
        ctxs[0].gather_rows_ctx(ctxs)
        assert np.array_equal(ctxs[0].lDisMap, ref["ldisp"]) and np.array_equal(ctxs[0].rDisMap, ref["rdisp"]), (W, H, D, dtype, cuts)
    finally:
        for c in ctxs:
            c.close()
