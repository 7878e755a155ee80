"""primestereomatch_amd - MI355X-native DispEst hot path (CVC -> CVF guided filter -> DispSel WTA).

The product is libprimesm_hip.so (csrc/, C ABI in include/primesm_hip.h) plus the host-side
mirrors of the reference's DispEst interface (host/ in C++, dispest.py in Python).
"""
from . import capi  # noqa: F401
from .dispest import DispEst, FrameRing  # noqa: F401

__all__ = ["capi", "DispEst", "FrameRing"]
