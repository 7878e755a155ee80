"""CPU model of the marching-kernel schedule (primestereomatch_amd/csrc/psm_kernels.hip:
march_pos / hsum8 / vstep / k_box8), lane for lane in numpy.  It proves on the CPU that the
block->(strip,segment,slice) decode covers every voxel exactly once and that the sliding-tree
evaluation order equals the oracle's canonical box filter bit for bit.  (The compiled kernels
themselves are checked against the oracle in the -m gpu tests.)"""
import numpy as np
import pytest

OUT_PER_WAVE = 56


def r101c(k, n):
    k = abs(k)
    if k >= n:
        k = 2 * (n - 1) - k
    return min(max(k, 0), n - 1)


def hsum8(v):
    """64 lanes; lane l gets the balanced-tree sum of lanes l..l+7 (indices wrap like & 63)."""
    v = v.astype(np.float32)
    s2 = v.astype(np.float64) + np.roll(v, -1).astype(np.float64)
    s4 = s2 + np.roll(s2, -2)
    return s4 + np.roll(s4, -4)


class VTree:
    def __init__(self):
        self.hp = np.zeros(64)
        self.s2 = [np.zeros(64), np.zeros(64)]
        self.s4 = [np.zeros(64) for _ in range(4)]

    def step(self, K, hs):
        n2 = self.hp + hs
        n4 = self.s2[K & 1] + n2
        n8 = self.s4[K & 3] + n4
        self.s2[K & 1] = n2
        self.s4[K & 3] = n4
        self.hp = hs
        return n8


def march_box8(vol, seg_rows, waves, order=0):
    D, H, W = vol.shape
    out = np.full(vol.shape, np.nan, np.float32)
    written = np.zeros(vol.shape, np.int32)
    nstrips = (W + OUT_PER_WAVE - 1) // OUT_PER_WAVE
    seg_rows = min(seg_rows if seg_rows > 0 else H, H)
    nsegs = (H + seg_rows - 1) // seg_rows
    nzg = (D + waves - 1) // waves
    npairs = nstrips * nsegs
    p8 = (npairs + 7) >> 3
    s8 = (nstrips + 7) >> 3
    nblocks = 8 * p8 * nzg if order == 0 else 8 * s8 * nsegs * nzg
    lanes = np.arange(64)
    for bid in range(nblocks):
        xcd, j = bid & 7, bid >> 3
        if order == 0:
            zg, pl = j % nzg, j // nzg
            pair = xcd * p8 + pl
            ok = pl < p8 and pair < npairs
            strip, seg = pair % nstrips, pair // nstrips
        else:
            sl, rest = j % s8, j // s8
            strip = xcd * s8 + sl
            if order == 1:
                zg, seg = rest % nzg, rest // nzg
            else:
                seg, zg = rest % nsegs, rest // nsegs
            ok = strip < nstrips and seg < nsegs and zg < nzg
        for wave in range(waves):
            d = zg * waves + wave
            if not (ok and d < D):
                continue
            x0 = strip * OUT_PER_WAVE
            cs = np.array([r101c(x0 - 4 + l, W) for l in lanes])
            xo = x0 + lanes
            ovalid = (lanes < OUT_PER_WAVE) & (xo < W)
            y0 = seg * seg_rows
            y1 = min(H, y0 + seg_rows)
            n = (y1 - y0) + 7
            ybase = y0 - 4
            t = VTree()
            i = 0
            while i < n:
                for K in range(8):
                    step = i + K
                    row = vol[d, r101c(ybase + step, H), cs]
                    n8 = t.step(K, hsum8(row))
                    if step >= 7 and step < n:
                        yo = ybase + step - 3
                        val = (n8 * 0.015625).astype(np.float32)
                        out[d, yo, xo[ovalid]] = val[ovalid]
                        written[d, yo, xo[ovalid]] += 1
                i += 8
    return out, written


@pytest.mark.parametrize("order", [0, 1, 2])
@pytest.mark.parametrize("shape,seg_rows,waves", [((3, 19, 70), 0, 4), ((5, 33, 130), 8, 2),
                                                  ((2, 8, 8), 3, 1), ((9, 40, 57), 16, 8), ((3, 20, 600), 7, 2)])
def test_march_model_matches_oracle(oracle, shape, seg_rows, waves, order):
    rng = np.random.default_rng(11)
    vol = (rng.random(shape, dtype=np.float32) * 2.7).astype(np.float32)
    out, written = march_box8(vol, seg_rows, waves, order)
    assert np.all(written == 1)                     # every voxel produced exactly once
    for d in range(shape[0]):
        assert np.array_equal(out[d], oracle.box8(vol[d]))   # bit-exact: same tree order


# ------------------------------------------------------------------------------------------
# Row bookkeeping of the fused producer/consumer kernel (psm_pc.hip: y0/y1, mstart/mend, nbA, consumer feeds):
# which model rows a segment [y0, y1) of an H-row image produces, against which ones its consumer taps read.
# (A one-row segment at the top of the image needed model row 4 - REFLECT_101 of row -4 - which the first version
# of the formula left out; the randomised stripe test on the GPU found it, this model pins it on the CPU.)
# ------------------------------------------------------------------------------------------
def r101(k, n):
    k = -k if k < 0 else k
    return 2 * (n - 1) - k if k >= n else k


def pc_segment_rows(H, y0, y1):
    mstart = max(0, y0 - 4)
    mend = min(H - 1, max(y1 + 2, 4 - y0))
    nbA = (mend - mstart + 1 + 3) >> 2                 # producer batches of four rows
    produced = set(range(mstart, min(H, mstart + 4 * nbA)))   # rows the producer waves write into the ring (clamped to the image)
    nf = (y1 - y0) + 7                                 # consumer feeds j = 0 .. nf-1: model row r101(y0 - 4 + j)
    needed = {r101(y0 - 4 + j, H) for j in range(nf)}
    return mstart, mend, produced, needed


@pytest.mark.parametrize("H", [8, 9, 12, 13, 33, 64, 135, 375])
def test_fused_kernel_segments_produce_every_model_row_they_consume(H):
    rng = np.random.default_rng(H)
    cases = [(0, 1), (0, 2), (1, 2), (H - 1, H), (H - 2, H), (0, H), (3, 4), (4, 5)]
    cases += [tuple(sorted(rng.choice(H + 1, size=2, replace=False))) for _ in range(200)]
    for y0, y1 in cases:
        y0, y1 = int(y0), int(y1)
        if not (0 <= y0 < y1 <= H):
            continue
        mstart, mend, produced, needed = pc_segment_rows(H, y0, y1)
        assert needed <= produced, (H, y0, y1, sorted(needed - produced))
        assert min(needed) >= mstart and max(needed) <= mend, (H, y0, y1)
        # the old formula (mend = min(H-1, y1+2)) fails exactly for the one-row segment at the top
        old_mend = min(H - 1, y1 + 2)
        if max(needed) > old_mend:
            assert (y0, y1) == (0, 1), (H, y0, y1)
