"""-m gpu: the second sharding axis of the hot path - row stripes (psm_set_rows / psm_gather_rows_ctx / psm_set_map_buffer).
A context restricted to output rows [y0, y1) must produce, for those rows, exactly the maps of the unrestricted run
(DispSel::CVSelect over the guided-filtered volumes, src/DispSel.cpp:96-104, src/CVF.cpp:72-165): the filter's vertical
support is bounded, borders reflect at the true image border, nothing is exchanged but the finished rows."""
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


def _bounds(H, G):
    return [(H * g // G, H * (g + 1) // G) for g in range(G)]


def _full(psm, oracle, l, r, D, dtype):
    ref = (oracle.pipeline_u8 if dtype == "u8" else oracle.pipeline_f32)(l, r, D, threads=8)
    return ref["ldisp"], ref["rdisp"]


@pytest.mark.parametrize("W,H,D,dtype,flags,G", [(260, 150, 12, "f32", 0, 3), (200, 97, 20, "f32", 1048576, 4), (107, 20, 9, "f32", 0, 7),
                                                 (230, 64, 33, "u8", 0, 2), (214, 40, 19, "u8", 1048576, 5), (260, 40, 200, "f32", 0, 2),
                                                 (330, 135, 16, "f32", 2097152, 3), (120, 33, 8, "f32", 1048576, 2),
                                                 (100, 12, 42, "u8", 0, 12), (100, 12, 42, "f32", 1048576, 12)])   # one row per stripe
def test_row_stripes_equal_the_whole_image(psm, oracle, W, H, D, dtype, flags, G):
    from primestereomatch_amd import capi, synth
    l, r, _ = synth.make_pair(W, H, D, seed=W + H + D)
    el, er = _full(psm, oracle, l, r, D, dtype)
    ctxs = [psm.DispEst(l, r, D, dtype=dtype) for _ in range(G)]
    try:
        for c, (y0, y1) in zip(ctxs, _bounds(H, G)):
            c.set_option(capi.PSM_OPT_FLAGS, flags)
            c.set_rows(y0, y1)
            c.CostConst_GPU(); c.CostFilter_GPU(); c.DispSelect_GPU()
            assert np.array_equal(c.lDisMap[y0:y1], el[y0:y1]) and np.array_equal(c.rDisMap[y0:y1], er[y0:y1]), (y0, y1)
        ctxs[0].gather_rows_ctx(ctxs)
        assert np.array_equal(ctxs[0].lDisMap, el) and np.array_equal(ctxs[0].rDisMap, er)
        # the gathered maps are whole: post-processing runs on them
        ctxs[0].LRCheck_GPU()
        assert np.array_equal(ctxs[0].lValid, oracle.lr_check(el, er)[0])
        # a second frame - another pair - through the same stripes (prepared image rows / guidance rows of the first frame
        # must not survive), and back to the whole image on one of the contexts
        l2, r2, _ = synth.make_pair(W, H, D, seed=W + H + D + 1)
        el2, er2 = _full(psm, oracle, l2, r2, D, dtype)
        for c in ctxs:
            c.setInputImages(l2, r2)
            c.CostConst_GPU(); c.CostFilter_GPU(); c.DispSelect_GPU()
        ctxs[-1].gather_rows_ctx(ctxs)
        assert np.array_equal(ctxs[-1].lDisMap, el2) and np.array_equal(ctxs[-1].rDisMap, er2)
        ctxs[1].set_rows(0, 0)
        ctxs[1].CostConst_GPU(); ctxs[1].CostFilter_GPU(); ctxs[1].DispSelect_GPU()
        assert np.array_equal(ctxs[1].lDisMap, el2) and np.array_equal(ctxs[1].rDisMap, er2)
        # a stripe context asked for something that reads whole planes (the filtered volume; the weighted median's colours)
        if dtype == "f32" and G > 1:
            q = ctxs[0].download_volume(0, 1, 2)
            full = oracle.pipeline_f32(l2, r2, D, threads=8, want_volumes=True)
            assert np.array_equal(q[0], full["lvol"][1])
    finally:
        for c in ctxs:
            c.close()


def test_stripes_of_disparity_shards(psm, oracle):
    """Both axes at once: a context holds slices [d0, d1) and rows [y0, y1); its minima merge over d, its rows gather over y."""
    from primestereomatch_amd import synth
    W, H, D = 180, 60, 21
    l, r, _ = synth.make_pair(W, H, D, seed=4)
    el, er = _full(psm, oracle, l, r, D, "f32")
    out_l, out_r = np.zeros_like(el), np.zeros_like(er)
    for (y0, y1) in _bounds(H, 2):
        shards = [psm.DispEst(l, r, D, d_range=rg) for rg in ((0, 8), (8, 21))]
        try:
            for s in shards:
                s.set_rows(y0, y1)
                s.CostConst_GPU(); s.CostFilter_GPU(); s.DispSelect_partial()
            shards[0].DispSelect_merge_ctx(shards)
            out_l[y0:y1], out_r[y0:y1] = shards[0].lDisMap[y0:y1], shards[0].rDisMap[y0:y1]
        finally:
            for s in shards:
                s.close()
    assert np.array_equal(out_l, el) and np.array_equal(out_r, er)


def test_stripe_errors(psm):
    from primestereomatch_amd import capi, synth
    W, H, D = 120, 48, 8
    l, r, _ = synth.make_pair(W, H, D, seed=1)
    a, b = psm.DispEst(l, r, D), psm.DispEst(l, r, D)
    try:
        with pytest.raises(capi.PsmError):
            a.set_rows(10, 10)
        with pytest.raises(capi.PsmError):
            a.set_rows(-1, 5)
        with pytest.raises(capi.PsmError):
            a.set_rows(0, H + 1)
        a.set_rows(0, 30); b.set_rows(24, H)
        for c in (a, b):
            c.CostConst_GPU(); c.CostFilter_GPU()
        with pytest.raises(capi.PsmError):
            a.gather_rows_ctx([a, b])              # no maps yet
        for c in (a, b):
            c.DispSelect_GPU()
        with pytest.raises(capi.PsmError):
            a.LRCheck_GPU()                        # stripe-only maps
        with pytest.raises(capi.PsmError):
            a.gather_rows_ctx([a, b])              # rows 24..29 held twice
        with pytest.raises(capi.PsmError):
            a.gather_rows_ctx([a])                 # rows 30.. missing
        b.set_rows(30, H)
        b.CostConst_GPU(); b.CostFilter_GPU(); b.DispSelect_GPU()
        a.gather_rows_ctx([a, b])
        # the storing form of the filter has no stripe support: refused, not silently whole
        a.set_option(capi.PSM_OPT_FLAGS, 8192)
        a.CostConst_GPU()
        with pytest.raises(capi.PsmError):
            a.CostFilter_GPU()
    finally:
        a.close(); b.close()


_MAP_BUFFER_SCRIPT = r"""
import sys, numpy as np, torch
torch.cuda.init()                              # torch's HIP runtime first: the library then binds to the same one (as in bench.py)
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/oracle")
import primestereomatch_amd as P
import psm_oracle_py as oracle
from primestereomatch_amd import synth
W, H, D = 160, 50, 10
l, r, _ = synth.make_pair(W, H, D, seed=2)
ref = oracle.pipeline_f32(l, r, D, threads=8)
el, er = ref["ldisp"], ref["rdisp"]
buf = torch.zeros(2 * H * W + 4, dtype=torch.uint8, device="cuda:0")
with P.DispEst(l, r, D) as de:
    de.set_map_buffer(buf.data_ptr())
    de.CostConst_GPU(); de.CostFilter_GPU(); de.DispSelect_device()
    torch.cuda.synchronize()
    got = buf[:2 * H * W].cpu().numpy().reshape(2, H, W)
    assert np.array_equal(got[0], el) and np.array_equal(got[1], er)
    de.LRCheck_GPU(); de.FillInv_GPU()         # post-processing works in the caller's buffer too
    torch.cuda.synchronize()
    lv = oracle.lr_check(el, er)[0]
    assert np.array_equal(buf[:H * W].cpu().numpy().reshape(H, W), oracle.fill_inv(el, lv))
    # stripes gathered by the caller: mark the buffer whole
    de.set_rows(0, 20)
    de.CostConst_GPU(); de.CostFilter_GPU(); de.DispSelect_device()
    try:
        de.LRCheck_GPU(); raise SystemExit("stripe-only maps were accepted")
    except P.capi.PsmError:
        pass
    buf[:2 * H * W] = torch.from_numpy(np.stack([el, er]).reshape(-1)).cuda()
    de.set_map_buffer(buf.data_ptr(), whole=True)
    de.LRCheck_GPU()
    assert np.array_equal(de.lValid, lv)
    de.set_map_buffer(None)
    try:
        de.download_maps(); raise SystemExit("maps reported in a buffer that holds none")
    except P.capi.PsmError:
        pass
print("map-buffer-ok")
"""


def test_map_buffer_in_caller_memory():
    """psm_set_map_buffer: the maps are written straight into a torch tensor (what bench.py all-gathers between ranks).
    Own process: torch's HIP runtime has to be the first one loaded."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    p = subprocess.run([sys.executable, "-c", _MAP_BUFFER_SCRIPT, ROOT], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0 and "map-buffer-ok" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])


@pytest.mark.parametrize("ndev,mode", [(3, "f32"), (2, "u8")])
def test_cpp_host_mirror_row_stripes(psm, oracle, golden, tmp_path, ndev, mode):
    """The C++ DispEst mirror with several contexts: one row stripe each (logical stripes on this GPU via
    PSM_HOST_LOGICAL_STRIPES; on a multi-GPU host one per device), psm_gather_rows_ctx, then the post-processing on the
    gathered maps - same files as the one-context run."""
    import os
    import subprocess
    from conftest import ROOT
    demo = os.path.join(ROOT, "primestereomatch_amd", "lib", "psm_demo")
    subprocess.run(["make", "-C", os.path.join(ROOT, "primestereomatch_amd", "host")], check=True, capture_output=True)
    pair, gold = golden("teddy_pair.npz"), golden("teddy_oracle_d64.npz")
    H, W, _ = pair["l_bgr"].shape
    pair["l_bgr"].tofile(tmp_path / "l.raw")
    pair["r_bgr"].tofile(tmp_path / "r.raw")
    env = dict(os.environ, PRIMESM_HIP_LIB=psm.capi.LIB_PATH, PSM_HOST_LOGICAL_STRIPES="1")
    p = subprocess.run([demo, str(tmp_path / "l.raw"), str(tmp_path / "r.raw"), str(W), str(H), "64",
                        str(tmp_path / "o"), str(ndev), mode, "0", "0", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    ld = np.fromfile(tmp_path / "o_ldisp.raw", np.uint8).reshape(H, W)
    rd = np.fromfile(tmp_path / "o_rdisp.raw", np.uint8).reshape(H, W)
    key = ("ldisp", "rdisp") if mode == "f32" else ("ldisp_u8mode", "rdisp_u8mode")
    assert np.array_equal(ld, gold[key[0]]) and np.array_equal(rd, gold[key[1]])
    lv = np.fromfile(tmp_path / "o_lvalid.raw", np.uint8).reshape(H, W)
    assert np.array_equal(lv, oracle.lr_check(ld, rd)[0])
    lpp = np.fromfile(tmp_path / "o_ldisp_pp.raw", np.uint8).reshape(H, W)
    assert np.array_equal(lpp, oracle.wgt_median(oracle.u8_to_f32(pair["l_bgr"]), oracle.fill_inv(ld, lv), lv, 64, right=False))


def test_cpp_host_mirror_fgf_on_a_striped_host(psm, oracle, golden, tmp_path):
    """The C++ DispEst mirror with several (logical) devices and the Fast Guided Filter: the FGF path has no stripes - the
    first device filters the whole image (round-2 advisor finding: every device filtered the full image and the gather then
    saw stripe bookkeeping that did not match) - and the maps equal the one-context FGF run."""
    import os
    import subprocess
    from conftest import ROOT
    demo = os.path.join(ROOT, "primestereomatch_amd", "lib", "psm_demo")
    subprocess.run(["make", "-C", os.path.join(ROOT, "primestereomatch_amd", "host")], check=True, capture_output=True)
    pair = golden("teddy_pair.npz")
    H, W, _ = pair["l_bgr"].shape
    pair["l_bgr"].tofile(tmp_path / "l.raw")
    pair["r_bgr"].tofile(tmp_path / "r.raw")
    env = dict(os.environ, PRIMESM_HIP_LIB=psm.capi.LIB_PATH, PSM_HOST_LOGICAL_STRIPES="1")
    p = subprocess.run([demo, str(tmp_path / "l.raw"), str(tmp_path / "r.raw"), str(W), str(H), "64",
                        str(tmp_path / "o"), "3", "f32", "0", "4", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    ref = oracle.pipeline_fgf(pair["l_bgr"], pair["r_bgr"], 64, s=4)
    ld = np.fromfile(tmp_path / "o_ldisp.raw", np.uint8).reshape(H, W)
    rd = np.fromfile(tmp_path / "o_rdisp.raw", np.uint8).reshape(H, W)
    assert np.array_equal(ld, ref["ldisp"]) and np.array_equal(rd, ref["rdisp"])
    lv = np.fromfile(tmp_path / "o_lvalid.raw", np.uint8).reshape(H, W)
    assert np.array_equal(lv, oracle.lr_check(ld, rd)[0])


def test_single_process_exchange_through_host_memory(psm, oracle):
    """psm_gather_rows_ctx / psm_disp_merge_ctx between devices WITHOUT peer access (hipDeviceCanAccessPeer says no) move every
    stripe / shard through a page-locked bounce buffer of the root.  One-GPU boxes force that path with PSM_OPT_GATHER_STAGED: same
    maps as the device-copy path and the oracle, and the legs are counted (a C++ host driving several GPUs from one process -
    host/DispEst.cpp - relies on exactly these two calls)."""
    from primestereomatch_amd import capi, synth
    W, H, D = 333, 150, 40
    l, r, _ = synth.make_pair(W, H, D, seed=9)
    ref = oracle.pipeline_f32(l, r, D, threads=8)
    lib = capi.load()
    for staged in (0, 1):
        parts = []
        for ya, yb in ((0, 50), (50, 51), (51, H)):
            de = psm.DispEst(l, r, D)
            de.set_rows(ya, yb)
            de.CostConst_GPU(); de.CostFilter_GPU(); de.DispSelect_device()
            parts.append(de)
        root = parts[1]                                      # (a root that is not the first stripe)
        root.set_option(capi.PSM_OPT_GATHER_STAGED, staged)
        root.gather_rows_ctx(parts)
        assert np.array_equal(root.lDisMap, ref["ldisp"]) and np.array_equal(root.rDisMap, ref["rdisp"])
        assert lib.psm_gather_staged_legs(root._h) == (4 if staged else 0)      # two other stripes x two maps
        for de in parts:
            de.close()
        shards = []
        for g in range(4):
            de = psm.DispEst(l, r, D, d_range=(D * g // 4, D * (g + 1) // 4))
            de.CostConst_GPU(); de.CostFilter_GPU(); de.DispSelect_partial()
            shards.append(de)
        root = shards[2]
        root.set_option(capi.PSM_OPT_GATHER_STAGED, staged)
        root.DispSelect_merge_ctx(shards)
        assert np.array_equal(root.lDisMap, ref["ldisp"]) and np.array_equal(root.rDisMap, ref["rdisp"])
        assert lib.psm_gather_staged_legs(root._h) == (4 if staged else 0)      # every shard's key planes (the root's own included)
        root.DispSelect_merge_ctx(shards)                    # a second frame reuses the bounce buffer
        assert np.array_equal(root.lDisMap, ref["ldisp"])
        for de in shards:
            de.close()

