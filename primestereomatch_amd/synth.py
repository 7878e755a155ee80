"""Synthetic stereo pairs for the configurations BASELINE.json names.

Left image: uint8 B,G,R; per channel a sum of 6 random-phase 2-D sinusoids (periods 8..128 px)
plus i.i.d. uniform noise +-12, clipped to [0,255]; ~10 % of the area is overwritten with
constant-colour rectangles (the ill-conditioned det ~ eps^3 regime of the guided filter).
Ground-truth disparity: a background ramp plus 5..9 constant rectangles with values in [2, D-2].
Right image: the left image warped by -disparity (nearest pixel, nearer surface wins), holes
filled from the left neighbour.  Everything is a pure function of (W, H, D, seed).
"""
from __future__ import annotations

import numpy as np


def make_pair(W: int, H: int, D: int, seed: int = 0):
    """Returns (left_bgr_u8, right_bgr_u8, gt_disparity_int32)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    left = np.empty((H, W, 3), np.float32)
    for c in range(3):
        acc = np.full((H, W), 127.5, np.float32)
        for _ in range(6):
            period = rng.uniform(8.0, 128.0)
            theta = rng.uniform(0.0, np.pi)
            phase = rng.uniform(0.0, 2 * np.pi)
            amp = rng.uniform(10.0, 28.0)
            k = 2 * np.pi / period
            acc += amp * np.sin(k * (np.cos(theta) * xx + np.sin(theta) * yy) + phase).astype(np.float32)
        acc += rng.uniform(-12.0, 12.0, size=(H, W)).astype(np.float32)
        left[:, :, c] = acc
    # constant-colour rectangles, ~10 % of the area
    area, target = 0, 0.10 * W * H
    while area < target:
        rw, rh = int(rng.integers(W // 16, W // 5 + 2)), int(rng.integers(H // 16, H // 5 + 2))
        x0, y0 = int(rng.integers(0, max(1, W - rw))), int(rng.integers(0, max(1, H - rh)))
        left[y0:y0 + rh, x0:x0 + rw, :] = rng.uniform(20.0, 235.0, size=3).astype(np.float32)
        area += rw * rh
    left_u8 = np.clip(np.rint(left), 0, 255).astype(np.uint8)

    # ground-truth disparity: ramp + rectangles
    hi = max(3, D - 2)
    ramp = 2 + (hi - 2) * 0.25 * (xx / max(1, W - 1)) + (hi - 2) * 0.15 * (yy / max(1, H - 1))
    disp = np.clip(np.rint(ramp), 2, hi).astype(np.int32)
    for _ in range(int(rng.integers(5, 10))):
        rw, rh = int(rng.integers(W // 10, W // 3 + 2)), int(rng.integers(H // 10, H // 3 + 2))
        x0, y0 = int(rng.integers(0, max(1, W - rw))), int(rng.integers(0, max(1, H - rh)))
        disp[y0:y0 + rh, x0:x0 + rw] = int(rng.integers(2, hi + 1))

    # forward warp: right[y, x - d] = left[y, x]; larger disparity (nearer) wins collisions
    xi = np.broadcast_to(np.arange(W, dtype=np.int64), (H, W))
    yi = np.broadcast_to(np.arange(H, dtype=np.int64)[:, None], (H, W))
    xt = xi - disp
    ok = xt >= 0
    tdisp = np.full((H, W), -1, np.int64)
    np.maximum.at(tdisp, (yi[ok], xt[ok]), disp[ok].astype(np.int64))
    have = tdisp >= 0
    src_x = np.where(have, np.minimum(xi + np.maximum(tdisp, 0), W - 1), 0)
    right_u8 = left_u8[yi, src_x]
    # holes: copy from the nearest filled pixel on the left (or the first filled one in the row)
    idx = np.where(have, xi, -1)
    idx = np.maximum.accumulate(idx, axis=1)
    first = np.argmax(have, axis=1)[:, None]
    idx = np.where(idx < 0, first, idx)
    right_u8 = right_u8[yi, idx]
    return np.ascontiguousarray(left_u8), np.ascontiguousarray(right_u8), disp


def random_volume(D: int, H: int, W: int, seed: int = 0) -> np.ndarray:
    """Cost-like volume for the box-filter roofline test: uniform in [0, 2.7)."""
    rng = np.random.default_rng(seed)
    return (rng.random((D, H, W), dtype=np.float32) * np.float32(2.7)).astype(np.float32)
