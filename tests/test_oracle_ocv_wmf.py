"""CPU-only tests of two oracle additions of round 2:

* PSMO_BOX_OCV - the summation order OpenCV's own engines execute for cv::boxFilter / cv::blur on CV_32F
  (RowSum<float,double> running sum + ColumnSum<double,float> running column accumulator).  Pinned against an
  independent pure-Python statement, shown to be a DIFFERENT order from the canonical tree (adversarial plane), and
  shown to give bit-identical filtered volumes and maps on the reference's Middlebury pairs - i.e. on that data the
  canonical order the HIP kernels evaluate IS the order the reference binary executes.
* psmo_wgt_median - src/PP.cpp:145-247, pinned against an independent pure-Python statement and properties.
"""
import math

import numpy as np
import pytest


def r101(k, n):
    k = -k if k < 0 else k
    return 2 * (n - 1) - k if k >= n else k


def py_box_ocv(p, k):
    """RowSum<float,double> + ColumnSum<double,float>, written from the algorithm description, scalar Python."""
    H, W = p.shape
    an = k // 2
    hs = np.zeros((H, W), np.float64)
    for y in range(H):
        ext = [float(p[y, r101(i - an, W)]) for i in range(W + k - 1)]
        if k in (3, 5):
            for x in range(W):
                a = ext[x]
                for i in range(1, k):
                    a = a + ext[x + i]
                hs[y, x] = a
        else:
            s = 0.0
            for i in range(k):
                s += ext[i]
            hs[y, 0] = s
            for x in range(W - 1):
                s += ext[x + k] - ext[x]
                hs[y, x + 1] = s
    out = np.zeros((H, W), np.float32)
    SUM = [0.0] * W
    for j in range(k - 1):
        row = hs[r101(j - an, H)]
        for x in range(W):
            SUM[x] += row[x]
    scale = 1.0 / (k * k)
    for y in range(H):
        sp = hs[r101(y + k - 1 - an, H)]
        sm = hs[r101(y - an, H)]
        for x in range(W):
            s0 = SUM[x] + sp[x]
            out[y, x] = np.float32(s0 * scale)
            SUM[x] = s0 - sm[x]
    return out


def test_box_ocv_matches_independent_statement(oracle):
    rng = np.random.default_rng(5)
    for shape in ((9, 13), (16, 24), (23, 17)):
        p = (rng.standard_normal(shape) * 10.0 ** rng.integers(-6, 6, shape)).astype(np.float32)
        with oracle.box_order(oracle.BOX_OCV):
            got = oracle.box8(p)
        assert np.array_equal(got, py_box_ocv(p, 8))


def test_box_ocv_is_a_different_order_than_the_tree(oracle):
    # a huge value passing through the window: the running sum keeps the rounding residue of 1e12 after the value has
    # left the window, the per-output tree does not
    p = np.zeros((16, 32), np.float32)
    p[:, 10] = 1e12
    p[:, 11:] = 1e-3
    tree = oracle.box8(p)
    with oracle.box_order(oracle.BOX_OCV):
        ocv = oracle.box8(p)
    assert np.array_equal(ocv, py_box_ocv(p, 8))
    assert tree[8, 20] == np.float32(1e-3)
    assert ocv[8, 20] != tree[8, 20]           # 9.765625e-4: drifted
    # and the switch is restored on exit
    assert np.array_equal(oracle.box8(p), tree)


def test_box_orders_agree_on_cost_like_planes(oracle):
    # values in the observed range of costs / products / model planes: every double addition is exact, so any order
    # gives the same bits
    rng = np.random.default_rng(6)
    for scale in (2.7, 1.0, 300.0):
        p = (rng.random((40, 56), dtype=np.float32) * np.float32(scale)).astype(np.float32)
        tree = oracle.box8(p)
        with oracle.box_order(oracle.BOX_OCV):
            assert np.array_equal(oracle.box8(p), tree)


@pytest.mark.parametrize("name", ["cones", "teddy"])
def test_ocv_order_pipeline_equals_canonical_on_middlebury(oracle, golden, name):
    """The number DESIGN.md 2 quotes: on the reference's own data the OpenCV-order evaluation and the canonical tree give
    the same filtered volumes bit for bit (max|dq| = 0, 0 voxels > 1e-4, 0 WTA pixels changed)."""
    g = golden(f"{name}_pair.npz")
    l, r = g["l_bgr"], g["r_bgr"]
    D = 64
    a = oracle.pipeline_f32(l, r, D, threads=8, want_volumes=True)
    with oracle.box_order(oracle.BOX_OCV):
        b = oracle.pipeline_f32(l, r, D, threads=8, want_volumes=True)
    for k in ("lvol", "rvol"):
        d = np.abs(a[k].astype(np.float64) - b[k])
        assert d.max() <= 1e-4                     # the stated float-mode tolerance
        assert np.array_equal(a[k], b[k])          # and in fact bit-identical
    assert np.array_equal(a["ldisp"], b["ldisp"]) and np.array_equal(a["rdisp"], b["rdisp"])


def test_ocv_order_fgf_equals_canonical_on_cones(oracle, golden):
    g = golden("cones_pair.npz")
    l, r = g["l_bgr"], g["r_bgr"]
    for s in (2, 4, 8):      # blur kernels 9, 5, 3: running row sums for 9, per-output row sums for 5 and 3
        a = oracle.pipeline_fgf(l, r, 32, s=s, want_volumes=True)
        with oracle.box_order(oracle.BOX_OCV):
            b = oracle.pipeline_fgf(l, r, 32, s=s, want_volumes=True)
        assert np.max(np.abs(a["lvol"].astype(np.float64) - b["lvol"])) <= 1e-4
        assert np.array_equal(a["ldisp"], b["ldisp"]) and np.array_equal(a["rdisp"], b["rdisp"])


# ---------------------------------------------------------------------------------------------
# weighted median (src/PP.cpp:145-247)
# ---------------------------------------------------------------------------------------------

def py_wgt_median(img, dis, valid, maxDis, right):
    """Independent scalar statement.  float32 arithmetic is spelled out with numpy scalars."""
    f32 = np.float32
    H, W = dis.shape
    dis = dis.copy()
    for y in range(H):
        for x in range(W):
            if valid[y, x]:
                continue
            hist = [f32(0)] * maxDis
            tot = f32(0)
            for wy in range(-9, 10):
                qy = (y + wy + H) % H
                for wx in range(-9, 10):
                    qx = (x + wx + W) % W
                    qd = int(dis[qy, qx])
                    if qd == 0:
                        continue
                    dw = f32(wx * wx + wy * wy)
                    e = [f32(img[y, x, c] - img[qy, qx, c]) for c in range(3)]
                    cw = f32(f32(f32(e[0] * e[0]) + f32(e[1] * e[1])) + f32(e[2] * e[2]))
                    if right:
                        dw = f32(np.sqrt(dw))
                        cw = f32(np.sqrt(cw))
                    w = f32(math.exp(float(f32(-dw / f32(81))) - float(cw) / (0.1 * 0.1)))
                    hist[qd] = f32(hist[qd] + w)
                    tot = f32(tot + w)
            half = f32(tot / f32(2))
            acc = f32(0)
            out = 0
            for d in range(maxDis):
                acc = f32(acc + hist[d])
                if acc >= half:
                    out = d
                    break
            dis[y, x] = out
    return dis


def _wm_case(seed, H=21, W=26, D=16, frac_invalid=0.3):
    rng = np.random.default_rng(seed)
    img = (rng.integers(0, 256, (H, W, 3)).astype(np.float32) * np.float32(1 / 255.0)).astype(np.float32)
    # piecewise-smooth image so that colour weights are not all ~0
    img[:, : W // 2] = img[0, 0] + (img[:, : W // 2] - img[0, 0]) * np.float32(0.05)
    dis = rng.integers(0, D, (H, W)).astype(np.uint8)
    valid = (rng.random((H, W)) > frac_invalid).astype(np.uint8)
    return img, dis, valid, D


@pytest.mark.parametrize("right", [False, True])
def test_wgt_median_matches_independent_statement(oracle, right):
    img, dis, valid, D = _wm_case(11)
    got = oracle.wgt_median(img, dis, valid, D, right=right)
    assert np.array_equal(got, py_wgt_median(img, dis, valid, D, right))


def test_wgt_median_properties(oracle):
    img, dis, valid, D = _wm_case(12, H=30, W=40, D=32)
    out = oracle.wgt_median(img, dis, valid, D)
    assert np.array_equal(out[valid != 0], dis[valid != 0])        # valid pixels are never touched
    assert out.max() < D
    # a constant map stays constant (the only voting bin); an all-zero map stays zero (nobody votes -> bin 0)
    const = np.full_like(dis, 7)
    assert np.array_equal(oracle.wgt_median(img, const, valid, D), const)
    zero = np.zeros_like(dis)
    assert np.array_equal(oracle.wgt_median(img, zero, valid, D), zero)
    # everything valid -> identity
    assert np.array_equal(oracle.wgt_median(img, dis, np.ones_like(valid), D), dis)
    # the left and the right formula are different filters
    assert not np.array_equal(oracle.wgt_median(img, dis, valid, D, right=True), out)


def test_wgt_median_is_sequential_in_place(oracle):
    """The reference updates the map in place in raster order: a filtered pixel sees the filtered pixels before it.
    A Jacobi-style (all from the input map) evaluation is a different filter - the device form must not be that."""
    img, dis, valid, D = _wm_case(13, H=24, W=24, D=16, frac_invalid=0.6)
    img[:] = np.float32(0.5)          # constant colour: weights depend on distance only, every neighbour votes
    dis = np.where(np.random.default_rng(3).random(dis.shape) < 0.5, 2, 12).astype(np.uint8)   # two camps, ~50/50
    seq = oracle.wgt_median(img, dis, valid, D)
    jac = dis.copy()
    ys, xs = np.nonzero(valid == 0)
    for y, x in zip(ys, xs):
        one = np.ones_like(valid)
        one[y, x] = 0
        jac[y, x] = oracle.wgt_median(img, dis, one, D)[y, x]
    assert not np.array_equal(seq, jac)


@pytest.mark.parametrize("seed,frac,two_camps", [(21, 0.5, False), (22, 0.7, True), (23, 1.0, True)])
def test_wgt_median_is_the_fixed_point_of_parallel_sweeps(oracle, seed, frac, two_camps):
    """What the device's product form relies on (psm_wgt_median): the in-place raster-order map s is the unique
    solution of s[p] = f(s[q] for invalid q earlier than p, input[q] otherwise).  Sweeping new[p] = f(cur[earlier], input[later])
    over all invalid pixels at once (Jacobi, from cur = input) reaches a map that no sweep changes, and that map is s - also
    when every pixel is invalid and on a two-camp map built to make changes propagate.  f is evaluated with the oracle itself
    (one invalid pixel per call)."""
    H, W = 14, 16
    img, dis, valid, D = _wm_case(seed, H=H, W=W, D=12, frac_invalid=frac)
    if two_camps:
        img[:] = np.float32(0.5)
        dis = np.where(np.random.default_rng(seed).random(dis.shape) < 0.5, 2, 9).astype(np.uint8)
    seq = oracle.wgt_median(img, dis, valid, D)
    idx = np.arange(H * W).reshape(H, W)
    cur = dis.copy()
    inv = list(zip(*np.nonzero(valid == 0)))
    for sweep in range(1, 200):
        new = cur.copy()
        for (y, x) in inv:
            seen = np.where(idx < idx[y, x], cur, dis)        # earlier pixels: current iterate, later ones (and p): the input
            one = np.ones_like(valid)
            one[y, x] = 0
            new[y, x] = oracle.wgt_median(img, seen, one, D)[y, x]
        if np.array_equal(new, cur):
            break
        cur = new
    else:
        pytest.fail("no fixed point within 200 sweeps")
    assert np.array_equal(cur, seq)
    assert sweep <= len(inv) + 1                               # (worst case: one pixel of the dependency chain per sweep)
    print(f"[wmf-fixed-point] {len(inv)} invalid pixels, {sweep} sweeps")


@pytest.mark.parametrize("seed,frac,two_camps,geo", [(31, 0.5, False, (14, 16, 12)), (32, 0.7, True, (14, 16, 12)), (33, 1.0, True, (14, 16, 12)),
                                                     (34, 0.9, False, (30, 46, 24)), (35, 1.0, True, (26, 40, 16))])
def test_wgt_median_sweeps_with_changes_visible_at_once(oracle, seed, frac, two_camps, geo):
    """Round 6's form of the sweeps: a change is written the moment it is found, so an evaluation of the same sweep
    sees the old value or the new one depending on timing, and only the pixels a change can reach - the later invalid pixels with
    the changed one in their window - are evaluated in the next sweep.  Modelled here with a random order of the evaluations
    inside a sweep and a coin per evaluation for "reads the values of this sweep's earlier writers or the sweep's starting map":
    whatever the timing, the sweeps end (a sweep without a change) in the sequential map."""
    H, W, D0 = geo
    R = 9
    img, dis, valid, D = _wm_case(seed, H=H, W=W, D=D0, frac_invalid=frac)
    if two_camps:
        img[:] = np.float32(0.5)
        dis = np.where(np.random.default_rng(seed).random(dis.shape) < 0.5, 2, 9).astype(np.uint8)
    seq = oracle.wgt_median(img, dis, valid, D)
    idx = np.arange(H * W).reshape(H, W)
    rng = np.random.default_rng(seed + 100)
    cur = dis.copy()
    active = [tuple(p) for p in zip(*np.nonzero(valid == 0))]
    sweeps = evals = 0
    while active:
        sweeps += 1
        assert sweeps < 400
        start = cur.copy()
        nxt = set()
        for k in rng.permutation(len(active)):
            y, x = active[k]
            view = cur if rng.random() < 0.5 else start            # (timing: this sweep's writes seen, or not)
            seen = np.where(idx < idx[y, x], view, dis)
            one = np.ones_like(valid)
            one[y, x] = 0
            v = oracle.wgt_median(img, seen, one, D)[y, x]
            evals += 1
            if v != cur[y, x]:
                cur[y, x] = v                                      # written at once ...
                for wy in range(-R, R + 1):                        # ... and its later invalid window neighbours queued
                    for wx in range(-R, R + 1):
                        py, px = (y - wy) % H, (x - wx) % W
                        if idx[py, px] > idx[y, x] and valid[py, px] == 0:
                            nxt.add((py, px))
        active = sorted(nxt)
    assert np.array_equal(cur, seq)
    print(f"[wmf-in-sweep] {int((valid == 0).sum())} invalid pixels, {sweeps} sweeps, {evals} evaluations")
