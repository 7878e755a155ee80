"""Host-side description of the disparity sharding used for N > 1 GPUs.

Rank g of G owns the contiguous slices d in [D*g//G, D*(g+1)//G) of both volumes.  After the
local WTA each rank holds one packed 64-bit key per pixel and side; one all-gather (RCCL) of
these keys followed by a signed minimum reproduces DispSel::CVSelect exactly (strict '<',
lowest d wins ties, d = 0 never a candidate, NaN never wins).

pack_keys/unpack_disp restate in numpy what k_wta / k_merge do on the device
(primestereomatch_amd/csrc/psm_kernels.hip: pack_key_f32); they exist so the protocol can be
tested on the CPU with gloo.  The product path never calls them.
"""
from __future__ import annotations

import numpy as np


def shard_bounds(D: int, G: int):
    return [(D * g // G, D * (g + 1) // G) for g in range(G)]


def pack_keys(min_cost: np.ndarray, min_disp: np.ndarray) -> np.ndarray:
    """(float32 cost, int32 d) -> int64 key; signed order == (cost, d) lexicographic order."""
    c = np.ascontiguousarray(min_cost, np.float32) + np.float32(0.0)   # -0 -> +0
    u = c.view(np.uint32).astype(np.uint64)
    neg = (u & np.uint64(0x80000000)) != 0
    u = np.where(neg, (~u) & np.uint64(0xFFFFFFFF), u | np.uint64(0x80000000))
    k = (u << np.uint64(32)) | np.ascontiguousarray(min_disp).astype(np.uint64)
    return (k ^ np.uint64(0x8000000000000000)).view(np.int64)


def unpack_disp(keys: np.ndarray) -> np.ndarray:
    return (keys.view(np.uint64) & np.uint64(0xFFFFFFFF)).astype(np.uint8)
