"""ctypes binding of oracle/libpsm_oracle.so (the CPU restatement of the reference hot path).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package (primestereomatch_amd) never imports this.
"parity unpinned": see oracle/psm_oracle.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpsm_oracle.so")
_lib = None


class Times(C.Structure):
    _fields_ = [("cvc_ms", C.c_double), ("cvf_ms", C.c_double), ("dispsel_ms", C.c_double)]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "psm_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B", "libpsm_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.psmo_pipeline_f32.restype = C.c_int
        _lib.psmo_pipeline_u8.restype = C.c_int
        _lib.psmo_eval_bad_pixels.restype = C.c_uint
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def u8_to_f32(img_u8):
    src = _u8(img_u8)
    dst = np.empty(src.shape, np.float32)
    lib().psmo_u8_to_f32(_p(src), C.c_size_t(src.size), _p(dst))
    return dst


def cvc_preprocess(img_f32):
    img = _f32(img_f32)
    H, W, _ = img.shape
    g = np.empty((H, W), np.float32)
    lib().psmo_cvc_preprocess(_p(img), H, W, _p(g))
    return g


def cvc_build(lImg, rImg, lG, rG, d, right=False):
    """Argument order follows the reference (right: caller swaps the images)."""
    lImg, rImg, lG, rG = _f32(lImg), _f32(rImg), _f32(lG), _f32(rG)
    H, W, _ = lImg.shape
    out = np.empty((H, W), np.float32)
    fn = lib().psmo_cvc_build_right if right else lib().psmo_cvc_build_left
    fn(_p(lImg), _p(rImg), _p(lG), _p(rG), H, W, int(d), _p(out))
    return out


def box8(plane):
    src = _f32(plane)
    H, W = src.shape
    dst = np.empty((H, W), np.float32)
    lib().psmo_box8(_p(src), H, W, _p(dst))
    return dst


def cvf_preprocess(img_f32):
    img = _f32(img_f32)
    H, W, _ = img.shape
    rgb = np.empty((3, H, W), np.float32)
    mean = np.empty((3, H, W), np.float32)
    var = np.empty((6, H, W), np.float32)
    lib().psmo_cvf_preprocess(_p(img), H, W, _p(rgb), _p(mean), _p(var))
    return rgb, mean, var


def guided_filter(rgb, mean, var, p, want_ab=False):
    rgb, mean, var = _f32(rgb), _f32(mean), _f32(var)
    q = _f32(p).copy()
    H, W = q.shape
    ab = np.empty((4, H, W), np.float32) if want_ab else None
    lib().psmo_guided_filter(_p(rgb), _p(mean), _p(var), H, W, _p(q), _p(ab))
    return (q, ab) if want_ab else q


def solve_models(var6, cov3):
    """src/CVF.cpp:102-149 alone on planar inputs: var6 [6][n], cov3 [3][n] -> a [3][n] (honours `variant`)."""
    v, c = _f32(var6), _f32(cov3)
    n = v.shape[1]
    assert v.shape == (6, n) and c.shape == (3, n)
    a = np.empty((3, n), np.float32)
    lib().psmo_solve_models(_p(v), _p(c), C.c_size_t(n), _p(a))
    return a


FMA_PROBE_PATH = os.path.join(_HERE, "libpsm_fma_probe.so")


def fma_probe_solve(var6, cov3):
    """The same solve loop as plain C compiled by THIS host's gcc with -O2 -mfma -ffp-contract=fast (oracle/fma_probe.c);
    None when the probe could not be built (no x86-64 FMA host)."""
    if not os.path.exists(FMA_PROBE_PATH):
        subprocess.run(["make", "-C", _HERE, "fma_probe"], check=False, stdout=subprocess.DEVNULL)
    if not os.path.exists(FMA_PROBE_PATH):
        return None
    pl = C.CDLL(FMA_PROBE_PATH)
    v, c = _f32(var6), _f32(cov3)
    n = v.shape[1]
    a = np.empty((3, n), np.float32)
    pl.psmo_probe_solve(_p(v), _p(c), C.c_size_t(n), _p(a))
    return a


def wta(vol):
    vol = _f32(vol)
    D, H, W = vol.shape
    out = np.empty((H, W), np.uint8)
    lib().psmo_wta(_p(vol), D, H, W, _p(out))
    return out


def wta_partial(vol, d_begin, d_end):
    vol = _f32(vol)
    _, H, W = vol.shape
    mc = np.empty((H, W), np.float32)
    md = np.empty((H, W), np.int32)
    lib().psmo_wta_partial(_p(vol), int(d_begin), int(d_end), H, W, _p(mc), _p(md))
    return mc, md


def pipeline_f32(l_bgr, r_bgr, D, threads=8, want_volumes=False, want_raw=False):
    l, r = _u8(l_bgr), _u8(r_bgr)
    H, W, _ = l.shape
    ld = np.empty((H, W), np.uint8)
    rd = np.empty((H, W), np.uint8)
    lv = np.empty((D, H, W), np.float32) if want_volumes else None
    rv = np.empty((D, H, W), np.float32) if want_volumes else None
    rl = np.empty((D, H, W), np.float32) if want_raw else None
    rr = np.empty((D, H, W), np.float32) if want_raw else None
    t = Times()
    rc = lib().psmo_pipeline_f32(_p(l), _p(r), H, W, int(D), int(threads), _p(ld), _p(rd), _p(lv),
                                 _p(rv), _p(rl), _p(rr), C.byref(t))
    if rc != 0:
        raise ValueError("psmo_pipeline_f32 rejected the arguments (rc=%d)" % rc)
    out = {"ldisp": ld, "rdisp": rd, "cvc_ms": t.cvc_ms, "cvf_ms": t.cvf_ms,
           "dispsel_ms": t.dispsel_ms}
    if want_volumes:
        out["lvol"], out["rvol"] = lv, rv
    if want_raw:
        out["raw_l"], out["raw_r"] = rl, rr
    return out


def pipeline_f32_maps(l_bgr, r_bgr, D, threads=8):
    """The two maps of pipeline_f32 without holding the volumes (blocks of `threads` slices folded into the running WTA)."""
    l, r = _u8(l_bgr), _u8(r_bgr)
    H, W, _ = l.shape
    ld = np.empty((H, W), np.uint8)
    rd = np.empty((H, W), np.uint8)
    fn = lib().psmo_pipeline_f32_maps
    fn.restype = C.c_int
    rc = fn(_p(l), _p(r), H, W, int(D), int(threads), _p(ld), _p(rd))
    if rc != 0:
        raise ValueError("psmo_pipeline_f32_maps rejected the arguments (rc=%d)" % rc)
    return {"ldisp": ld, "rdisp": rd}


def wm_weights(p3, q3, wx, wy, right):
    """wgtMedian's bilateral weights for n operand tuples (p3, q3: n x 3 float32 colours; wx, wy: n ints), as the host forms them."""
    p3, q3 = _f32(p3), _f32(q3)
    fn = lib().psmo_wm_weight
    fn.restype = C.c_float
    out = np.empty(len(p3), np.float32)
    for i in range(len(p3)):
        out[i] = fn(_p(p3[i]), _p(q3[i]), int(wx[i]), int(wy[i]), int(bool(right)))
    return out


def gray_grad_u8(img_u8):
    img = _u8(img_u8)
    H, W, _ = img.shape
    gray = np.empty((H, W), np.uint8)
    grd = np.empty((H, W), np.uint8)
    lib().psmo_gray_grad_u8(_p(img), H, W, _p(gray), _p(grd))
    return gray, grd


def pipeline_u8(l_bgr, r_bgr, D, threads=8, want_volumes=False, want_raw=False):
    l, r = _u8(l_bgr), _u8(r_bgr)
    H, W, _ = l.shape
    ld = np.empty((H, W), np.uint8)
    rd = np.empty((H, W), np.uint8)
    lv = np.empty((D, H, W), np.uint8) if want_volumes else None
    rv = np.empty((D, H, W), np.uint8) if want_volumes else None
    rl = np.empty((D, H, W), np.uint8) if want_raw else None
    rr = np.empty((D, H, W), np.uint8) if want_raw else None
    t = Times()
    rc = lib().psmo_pipeline_u8(_p(l), _p(r), H, W, int(D), int(threads), _p(ld), _p(rd), _p(lv),
                                _p(rv), _p(rl), _p(rr), C.byref(t))
    if rc != 0:
        raise ValueError("psmo_pipeline_u8 rejected the arguments (rc=%d)" % rc)
    out = {"ldisp": ld, "rdisp": rd, "cvc_ms": t.cvc_ms, "cvf_ms": t.cvf_ms,
           "dispsel_ms": t.dispsel_ms}
    if want_volumes:
        out["lvol"], out["rvol"] = lv, rv
    if want_raw:
        out["raw_l"], out["raw_r"] = rl, rr
    return out


def lr_check(ldis, rdis):
    l, r = _u8(ldis), _u8(rdis)
    H, W = l.shape
    lv = np.empty((H, W), np.uint8)
    rv = np.empty((H, W), np.uint8)
    lib().psmo_lr_check(_p(l), _p(r), H, W, _p(lv), _p(rv))
    return lv, rv


def fill_inv(dis, valid):
    d = _u8(dis).copy()
    v = _u8(valid)
    H, W = d.shape
    lib().psmo_fill_inv(_p(d), _p(v), H, W)
    return d


def eval_bad_pixels(disp, gt, mask, maxDis, scale_factor, error_threshold=4):
    d, g = _u8(disp), _u8(gt)
    m = _u8(mask) if mask is not None else None
    H, W = d.shape
    avg = C.c_float(0)
    bad = lib().psmo_eval_bad_pixels(_p(d), _p(g), _p(m), H, W, int(maxDis), int(scale_factor),
                                     int(error_threshold), C.byref(avg))
    return int(bad), float(avg.value)


def fgf_setup(img_f32, s):
    img = _f32(img_f32)
    H, W, _ = img.shape
    out = np.empty((12, H // s, W // s), np.float32)
    lib().psmo_fgf_setup(_p(img), H, W, int(s), _p(out))
    return out


def fgf_filter(img_f32, setup, p, s):
    img, setup = _f32(img_f32), _f32(setup)
    q = _f32(p).copy()
    H, W = q.shape
    lib().psmo_fgf_filter(_p(img), _p(setup), H, W, int(s), _p(q))
    return q


def pipeline_fgf(l_bgr, r_bgr, D, s=4, threads=8, want_volumes=False):
    l, r = _u8(l_bgr), _u8(r_bgr)
    H, W, _ = l.shape
    ld = np.empty((H, W), np.uint8)
    rd = np.empty((H, W), np.uint8)
    lv = np.empty((D, H, W), np.float32) if want_volumes else None
    rv = np.empty((D, H, W), np.float32) if want_volumes else None
    t = Times()
    lib().psmo_pipeline_fgf.restype = C.c_int
    rc = lib().psmo_pipeline_fgf(_p(l), _p(r), H, W, int(D), int(threads), int(s), _p(ld), _p(rd), _p(lv), _p(rv),
                                 C.byref(t))
    if rc != 0:
        raise ValueError("psmo_pipeline_fgf rejected the arguments (rc=%d)" % rc)
    out = {"ldisp": ld, "rdisp": rd, "cvc_ms": t.cvc_ms, "cvf_ms": t.cvf_ms, "dispsel_ms": t.dispsel_ms}
    if want_volumes:
        out["lvol"], out["rvol"] = lv, rv
    return out


BOX_TREE, BOX_OCV = 0, 1
VAR_FABS_DOUBLE, VAR_FMA_SOLVE = 1, 2
VAR_F32_L1, VAR_F32_L2, VAR_RUNCOL = 4, 8, 16     # tolerance-form models of the per-slice box filters (psm_oracle.c: box8_slice)


class variant:
    """Context manager: evaluate the oracle under an alternative, toolchain-dependent READING of the reference
    (VAR_FABS_DOUBLE: colour sum of myCostGrd in double; VAR_FMA_SOLVE: FMA-contracted 3x3 solve).  Never the canon."""

    def __init__(self, bits):
        self.bits = bits

    def __enter__(self):
        lib().psmo_get_variant.restype = C.c_int
        self.prev = lib().psmo_get_variant()
        lib().psmo_set_variant(self.bits)
        return self

    def __exit__(self, *a):
        lib().psmo_set_variant(self.prev)
        return False




class box_order:
    """Context manager: run the enclosed oracle calls with the given box-filter summation order
    (BOX_TREE = canonical balanced tree, BOX_OCV = OpenCV's running RowSum/ColumnSum order)."""

    def __init__(self, order):
        self.order = int(order)

    def __enter__(self):
        lib().psmo_get_box_order.restype = C.c_int
        self.prev = lib().psmo_get_box_order()
        lib().psmo_set_box_order(self.order)
        return self

    def __exit__(self, *exc):
        lib().psmo_set_box_order(self.prev)
        return False


def wgt_median(img_f32, dis, valid, maxDis, right=False):
    """src/PP.cpp:145-247 wgtMedian for one map (right: the right-map formula with the two sqrt)."""
    img = _f32(img_f32)
    d = _u8(dis).copy()
    v = _u8(valid)
    H, W = d.shape
    lib().psmo_wgt_median(_p(img), _p(d), _p(v), H, W, int(maxDis), int(bool(right)))
    return d
