"""CPU model of the weighted median's "who is evaluated next" step (psm_pp.hip: k_wm_apply scatter form against the
mark + gather form): both must queue exactly the invalid pixels that have a pixel changed by the sweep among the EARLIER taps
of their 19 x 19 modulo-wrapped window (src/PP.cpp:164-166 reads the map in place, so only earlier pixels are dependencies) -
on images small enough that the window wraps onto itself, too."""
import numpy as np
import pytest

R = 9


def _scatter(changed, invalid, H, W):
    """k_wm_apply: every changed pixel stamps the later invalid pixels whose window holds it."""
    out = set()
    for pix in changed:
        y, x = divmod(pix, W)
        for wy in range(-R, R + 1):
            for wx in range(-R, R + 1):
                pp = ((y - wy) % H) * W + (x - wx) % W
                if pp > pix and pp in invalid:
                    out.add(pp)
    return out


def _gather(changed, invalid, H, W):
    """k_wm_apply (marking) + k_wm_gather: rowany[y][x] = row y changed within x +- 9 (wrapped); an invalid pixel looks at
    rowany of the window rows above it (in raster order: also through the wrap) at its own column, and at the earlier taps
    of its own row."""
    chgb = np.zeros((H, W), bool)
    rowany = np.zeros((H, W), bool)
    for pix in changed:
        y, x = divmod(pix, W)
        chgb[y, x] = True
        for wx in range(-R, R + 1):
            rowany[y, (x + wx) % W] = True
    out = set()
    for pix in invalid:
        y, x = divmod(pix, W)
        want = False
        for w in range(-R, R + 1):
            if w == 0:
                continue
            qy, qx = (y + w) % H, (x + w) % W
            if qy < y:
                want |= bool(rowany[qy, x])
            if qx < x:
                want |= bool(chgb[y, qx])
        if want:
            out.add(pix)
    return out


def _definition(changed, invalid, H, W):
    """The definition: p is evaluated again iff some tap (wy, wx) of its window is a changed pixel that precedes it."""
    out = set()
    for pix in invalid:
        y, x = divmod(pix, W)
        if any((((y + wy) % H) * W + (x + wx) % W) in changed and (((y + wy) % H) * W + (x + wx) % W) < pix
               for wy in range(-R, R + 1) for wx in range(-R, R + 1)):
            out.add(pix)
    return out


@pytest.mark.parametrize("H,W", [(9, 9), (9, 30), (12, 10), (19, 19), (25, 40), (10, 23)])
@pytest.mark.parametrize("frac", [0.05, 0.5])
def test_gather_form_queues_what_the_scatter_form_queues(H, W, frac):
    rng = np.random.default_rng(H * 100 + W)
    invalid = set(int(i) for i in np.flatnonzero(rng.random(H * W) < 0.5))
    changed = set(p for p in invalid if rng.random() < frac)
    d = _definition(changed, invalid, H, W)
    assert _scatter(changed, invalid, H, W) == d
    assert _gather(changed, invalid, H, W) == d
