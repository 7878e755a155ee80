"""Python mirror of the reference's `DispEst` accelerator interface (include/DispEst.h:21-51,
src/DispEst.cpp:272-328) on top of the C ABI.  Method names, argument meaning and return
conventions follow the reference so that tests read like calls into the reference:

    de = DispEst(l, r, maxDis, threads, useHIP=True)
    de.setInputImages(l, r); de.CostConst_GPU(); de.CostFilter_GPU(); de.DispSelect_GPU()
    de.lDisMap, de.rDisMap            # H x W uint8

The C++ twin (primestereomatch_amd/host/DispEst.h) is what a C++ host links; this class exists
for the Python tests and bench.  No CPU path lives here: without the HIP library and a GPU the
constructor raises.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi

MAX_CPU_THREADS = 8  # include/ComFunc.h:52


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class DispEst:
    def __init__(self, l, r, d: int, t: int = 8, ocl: bool = True, *, dtype: str = "f32",
                 device: int = 0, d_range=None, d_stride=None):
        """l, r: H x W x 3 images (uint8 as loaded by imread, or float32 scaled by 1/255 as
        StereoMatch::compute hands them over, src/StereoMatch.cpp:193-198); d: maxDis;
        t: host threads (kept for interface parity; unused by the GPU path); ocl: must be true
        (the accelerator path is the only one this package implements).
        dtype: "f32" | "u8" (8-bit char mode).  d_range=(d_begin,d_end): disparity shard."""
        if not ocl:
            raise capi.PsmError("DispEst: only the accelerator ('m' / OCL_DE) path exists in this package")
        l = np.asarray(l)
        r = np.asarray(r)
        if l.shape != r.shape or l.dtype != r.dtype:
            # src/DispEst.cpp:21-29: exits on mismatching types
            raise ValueError("DE: Error - Left & Right images are of different types.")
        if l.ndim != 3 or l.shape[2] != 3:
            raise ValueError("DispEst: images must be H x W x 3")
        self.hei, self.wid = int(l.shape[0]), int(l.shape[1])
        self.maxDis = int(d)
        self.threads = int(t)
        self.useOCL = True
        self.subsample_rate = 4
        self._lib = capi.load()
        self._dtype = capi.PSM_U8 if dtype == "u8" else capi.PSM_F32
        self._h = C.c_void_p()
        self.options = {}
        d0, d1 = (0, self.maxDis) if d_range is None else (int(d_range[0]), int(d_range[1]))
        self.d_begin, self.d_end = d0, d1
        if d_stride is not None:
            # strided ownership (psm_create_shard_strided): this object holds the slices d_stride[0], + d_stride[1], ... < maxDis
            self.d_begin, self.d_end = int(d_stride[0]), self.maxDis
            rc = self._lib.psm_create_shard_strided(C.byref(self._h), self.wid, self.hei, self.maxDis, int(d_stride[0]), int(d_stride[1]),
                                                    self._dtype, int(device))
        else:
            rc = self._lib.psm_create_shard(C.byref(self._h), self.wid, self.hei, self.maxDis, d0, d1,
                                            self._dtype, int(device))
        if rc != 0:
            self._h = C.c_void_p()
            raise capi.PsmError("DispEst: " + capi.last_error(None))
        self.lDisMap = np.zeros((self.hei, self.wid), np.uint8)
        self.rDisMap = np.zeros((self.hei, self.wid), np.uint8)
        self.lValid = np.zeros((self.hei, self.wid), np.uint8)
        self.rValid = np.zeros((self.hei, self.wid), np.uint8)
        self.setInputImages(l, r)

    # ---- lifetime -------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.psm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _ck(self, rc, what):
        capi.check(rc, self._h, what)

    # ---- reference interface ---------------------------------------------------------------
    def setInputImages(self, l, r) -> int:
        l = np.ascontiguousarray(l)
        r = np.ascontiguousarray(r)
        assert l.dtype == r.dtype  # src/DispEst.cpp:166
        if l.shape != (self.hei, self.wid, 3) or r.shape != l.shape:
            raise ValueError("setInputImages: image size differs from the one DispEst was built for")
        if l.dtype == np.uint8:
            depth = capi.PSM_IMG_U8
        elif l.dtype == np.float32:
            depth = capi.PSM_IMG_F32
        else:
            raise ValueError("setInputImages: images must be uint8 or float32")
        self._ck(self._lib.psm_upload_pair(self._h, _ptr(l), _ptr(r), 3, l.strides[0], depth),
                 "setInputImages")
        return 0

    def setThreads(self, newThreads: int) -> int:
        if newThreads > MAX_CPU_THREADS:  # src/DispEst.cpp:172-179
            return -1
        self.threads = int(newThreads)
        return 0

    def setSubsampleRate(self, newRate: int) -> None:
        self.subsample_rate = int(newRate)

    def CostConst_GPU(self) -> int:
        self._ck(self._lib.psm_cost_construct(self._h), "CostConst_GPU")
        return 0

    def CostFilter_GPU(self) -> int:
        self._ck(self._lib.psm_cost_filter(self._h), "CostFilter_GPU")
        return 0

    def CostFilter_FGF_GPU(self) -> int:
        """DispEst::CostFilter_FGF (src/DispEst.cpp:281-296) on the device: FastGuidedFilterColor with
        r = GIF_R_WIN, eps = GIF_EPS and s = subsample_rate (setSubsampleRate; default 4)."""
        self._ck(self._lib.psm_cost_filter_fgf(self._h, int(self.subsample_rate)), "CostFilter_FGF_GPU")
        return 0

    def DispSelect_GPU(self) -> int:
        self._ck(self._lib.psm_disp_select(self._h, _ptr(self.lDisMap), _ptr(self.rDisMap), self.wid),
                 "DispSelect_GPU")
        return 0

    # ---- extensions beyond the reference surface --------------------------------------------
    def LRCheck_GPU(self) -> int:
        """PP lrCheck (src/PP.cpp:17-50) on the device -> lValid / rValid."""
        self._ck(self._lib.psm_lr_check(self._h, _ptr(self.lValid), _ptr(self.rValid), self.wid),
                 "LRCheck_GPU")
        return 0

    def LRCheck_device(self):
        """PP lrCheck with the validity masks left on the device (bench: D2H excluded from the timed region)."""
        self._ck(self._lib.psm_lr_check(self._h, None, None, 0), "LRCheck_device")

    def download_valid(self):
        """The validity masks of the last LRCheck (psm_lr_check is idempotent on unchanged maps: run again, with download)."""
        self.LRCheck_GPU()
        return self.lValid, self.rValid

    def FillInv_GPU(self) -> int:
        """PP fillInv (src/PP.cpp:52-143) on the device; updates lDisMap / rDisMap."""
        self._ck(self._lib.psm_fill_invalid(self._h, _ptr(self.lDisMap), _ptr(self.rDisMap), self.wid),
                 "FillInv_GPU")
        return 0

    def WgtMedian_GPU(self) -> int:
        """PP wgtMedian (src/PP.cpp:145-247) on the device, for the pixels LRCheck_GPU marked invalid; updates
        lDisMap / rDisMap.  Same result as the reference's sequential in-place form."""
        self._ck(self._lib.psm_wgt_median(self._h, _ptr(self.lDisMap), _ptr(self.rDisMap), self.wid), "WgtMedian_GPU")
        return 0

    def wgt_median_stats(self):
        """(sweeps, evaluations) of the last WgtMedian_GPU per map [left, right]; sweeps = -1: the dataflow form ran."""
        import ctypes as C
        sw, ev = (C.c_int * 2)(), (C.c_longlong * 2)()
        self._ck(self._lib.psm_wgt_median_stats(self._h, sw, ev), "wgt_median_stats")
        return list(sw), list(ev)

    def upload_maps(self, lmap=None, rmap=None, lvalid=None, rvalid=None):
        arrs = [None if a is None else np.ascontiguousarray(a, dtype=np.uint8) for a in (lmap, rmap, lvalid, rvalid)]
        for a in arrs:
            assert a is None or a.shape == (self.hei, self.wid)
        self._ck(self._lib.psm_upload_maps(self._h, *[_ptr(a) for a in arrs], self.wid), "upload_maps")

    def set_option(self, option: int, value: int):
        self._ck(self._lib.psm_set_option(self._h, int(option), int(value)), "set_option")
        self.options[int(option)] = int(value)           # (what was accepted, for callers that inspect a context)

    def set_stream(self, stream_ptr: int | None):
        self._ck(self._lib.psm_set_stream(self._h, C.c_void_p(stream_ptr or 0)), "set_stream")

    def release_scratch(self):
        """Give back the on-first-use scratch (weighted-median cache, minima planes, exchange buffers); see psm_release_scratch."""
        self._ck(self._lib.psm_release_scratch(self._h), "release_scratch")

    def synchronize(self):
        self._ck(self._lib.psm_synchronize(self._h), "synchronize")

    def DispSelect_partial(self, dev_keys_ptr: int | None = None):
        self._ck(self._lib.psm_disp_select_partial(self._h, C.c_void_p(dev_keys_ptr or 0)),
                 "DispSelect_partial")

    def CostFilter_side(self, side: int):
        self._ck(self._lib.psm_cost_filter_side(self._h, int(side)), "CostFilter_side")

    def DispSelect_partial_side(self, side: int, dev_keys_ptr: int | None = None):
        self._ck(self._lib.psm_disp_select_partial_side(self._h, int(side), C.c_void_p(dev_keys_ptr or 0)),
                 "DispSelect_partial_side")

    def set_key_buffer(self, dev_keys_ptr: int | None):
        """The packed minima of the following frames go straight into this device buffer (2*H*W int64); None: own buffer."""
        self._ck(self._lib.psm_set_key_buffer(self._h, C.c_void_p(dev_keys_ptr or 0)), "set_key_buffer")

    def partial_keys(self):
        p, n = C.c_void_p(), C.c_size_t()
        self._ck(self._lib.psm_partial_keys(self._h, C.byref(p), C.byref(n)), "partial_keys")
        return p.value, n.value

    def DispSelect_merge(self, dev_keys_all_ptr: int, nranks: int, download: bool = True):
        self._ck(self._lib.psm_disp_merge(self._h, C.c_void_p(dev_keys_all_ptr), int(nranks),
                                          _ptr(self.lDisMap) if download else None,
                                          _ptr(self.rDisMap) if download else None, self.wid),
                 "DispSelect_merge")

    def DispSelect_merge_ctx(self, shards, download: bool = True):
        """Single-process exchange: merge the partial keys of `shards` (DispEst objects)."""
        arr = (C.c_void_p * len(shards))(*[s._h for s in shards])
        self._ck(self._lib.psm_disp_merge_ctx(self._h, arr, len(shards),
                                              _ptr(self.lDisMap) if download else None,
                                              _ptr(self.rDisMap) if download else None, self.wid),
                 "DispSelect_merge_ctx")

    def set_rows(self, y_begin: int = 0, y_end: int = 0):
        """Row stripe: CostFilter_GPU / DispSelect* compute output rows [y_begin, y_end) of the whole image only (all
        slices, both volumes, identical values); (0, 0): whole image.  Call before CostFilter_GPU."""
        self._ck(self._lib.psm_set_rows(self._h, int(y_begin), int(y_end)), "set_rows")

    def set_map_buffer(self, dev_maps_ptr: int | None, whole: bool = False):
        """The device maps [2][H][W] uint8 live in this device buffer (>= 2*H*W + 4 bytes) from now on; None: own buffer.
        whole: the buffer already holds both complete maps of the current frame."""
        self._ck(self._lib.psm_set_map_buffer(self._h, C.c_void_p(dev_maps_ptr or 0), int(whole)), "set_map_buffer")

    def gather_rows_ctx(self, stripes, download: bool = True):
        """Single-process exchange of the row-stripe sharding: the stripe rows of every context's maps -> this context."""
        arr = (C.c_void_p * len(stripes))(*[s._h for s in stripes])
        self._ck(self._lib.psm_gather_rows_ctx(self._h, arr, len(stripes),
                                               _ptr(self.lDisMap) if download else None,
                                               _ptr(self.rDisMap) if download else None, self.wid),
                 "gather_rows_ctx")

    def DispSelect_device(self):
        """WTA with the maps left on the device (bench: D2H excluded from the timed region)."""
        self._ck(self._lib.psm_disp_select(self._h, None, None, 0), "DispSelect_device")

    def download_maps(self):
        self._ck(self._lib.psm_download_maps(self._h, _ptr(self.lDisMap), _ptr(self.rDisMap), self.wid),
                 "download_maps")
        return self.lDisMap, self.rDisMap

    # ---- frame loop: the PCIe legs next to the kernels (src/main.cpp:64-73) ----
    def setInputImages_async(self, l, r) -> int:
        """The NEXT frame's pair: staged and copied on the copy stream while the current frame computes; the next
        CostConst_GPU adopts it."""
        l = np.ascontiguousarray(l)
        r = np.ascontiguousarray(r)
        if l.shape != (self.hei, self.wid, 3) or r.shape != l.shape or l.dtype != r.dtype:
            raise ValueError("setInputImages_async: image size / type differs from the one DispEst was built for")
        depth = capi.PSM_IMG_U8 if l.dtype == np.uint8 else capi.PSM_IMG_F32
        self._ck(self._lib.psm_upload_pair_async(self._h, _ptr(l), _ptr(r), 3, l.strides[0], depth), "setInputImages_async")
        return 0

    def download_maps_async(self):
        self._ck(self._lib.psm_download_maps_async(self._h), "download_maps_async")

    def download_maps_wait(self):
        self._ck(self._lib.psm_download_maps_wait(self._h, _ptr(self.lDisMap), _ptr(self.rDisMap), self.wid), "download_maps_wait")
        return self.lDisMap, self.rDisMap

    def filter_launch_times(self, max_launches: int = 4096):
        """PSM_OPT_PROFILE 2: [(ms, form)] of every launch of the fused filter kernel since the last call (form 1 = minima
        planes, 2 = key plane, 0 = storing), from time stamps the kernel takes itself; resets the record."""
        ms = (C.c_double * max_launches)()
        form = (C.c_int * max_launches)()
        n = C.c_int()
        self._ck(self._lib.psm_filter_launch_times(self._h, ms, form, max_launches, C.byref(n)), "filter_launch_times")
        return [(ms[i], form[i]) for i in range(n.value)]

    def _vdtype(self):
        return np.uint8 if self._dtype == capi.PSM_U8 else np.float32

    def download_volume(self, side: int, d0: int | None = None, d1: int | None = None):
        d0 = self.d_begin if d0 is None else d0
        d1 = self.d_end if d1 is None else d1
        out = np.empty((d1 - d0, self.hei, self.wid), self._vdtype())
        self._ck(self._lib.psm_download_volume(self._h, side, d0, d1, _ptr(out)), "download_volume")
        return out

    def upload_volume(self, side: int, vol, d0: int | None = None):
        vol = np.ascontiguousarray(vol, dtype=self._vdtype())
        d0 = self.d_begin if d0 is None else d0
        assert vol.shape[1:] == (self.hei, self.wid)
        self._ck(self._lib.psm_upload_volume(self._h, side, d0, d0 + vol.shape[0], _ptr(vol)),
                 "upload_volume")

    def filter_stage_a(self, side: int):
        self._ck(self._lib.psm_filter_stage_a(self._h, side), "filter_stage_a")

    def download_ab(self, d0: int | None = None, d1: int | None = None):
        d0 = self.d_begin if d0 is None else d0
        d1 = self.d_end if d1 is None else d1
        out = np.empty((d1 - d0, self.hei, self.wid, 4), np.float32)
        self._ck(self._lib.psm_download_ab(self._h, d0, d1, _ptr(out)), "download_ab")
        return out

    def download_guidance(self, side: int):
        out = np.empty((14, self.hei, self.wid), np.float32)
        self._ck(self._lib.psm_download_guidance(self._h, side, _ptr(out)), "download_guidance")
        return out

    def box8_volume(self, side: int, download: bool = True):
        out = np.empty((self.d_end - self.d_begin, self.hei, self.wid), np.float32) if download else None
        self._ck(self._lib.psm_box8_volume(self._h, side, _ptr(out)), "box8_volume")
        return out

    def stage_time_us(self, stage: int) -> float:
        v = C.c_double()
        self._ck(self._lib.psm_stage_time_us(self._h, stage, C.byref(v)), "stage_time_us")
        return v.value

    def kernel_time_ms(self, kernel: int):
        v, n = C.c_double(), C.c_int()
        self._ck(self._lib.psm_kernel_time_ms(self._h, kernel, C.byref(v), C.byref(n)), "kernel_time_ms")
        return v.value, n.value

    def reset_kernel_times(self):
        self._ck(self._lib.psm_reset_kernel_times(self._h), "reset_kernel_times")


def compute_batch(des):
    """CostConst_GPU + CostFilter_GPU + DispSelect (maps left on the device) of several DispEst objects of one geometry in
    shared launches (psm_compute_batch): the reference's loop over pairs / datasets (src/main.cpp:64-73,
    src/StereoMatch.cpp:556-607) as one grid.  Every object afterwards behaves as after the three single-pair calls
    (download_maps(), LRCheck_GPU(), download_volume(), ...)."""
    des = list(des)
    if not des:
        return
    arr = (C.c_void_p * len(des))(*[d._h for d in des])
    capi.check(des[0]._lib.psm_compute_batch(arr, len(des)), des[0]._h, "compute_batch")


def share_streams(des):
    """The DispEst objects of a batch run on one compute stream and one copy stream each way (psm_share_streams) - call once
    before a frame loop over batches."""
    des = list(des)
    if not des:
        return
    arr = (C.c_void_p * len(des))(*[d._h for d in des])
    capi.check(des[0]._lib.psm_share_streams(arr, len(des)), des[0]._h, "share_streams")


class FrameRing:
    """The reference's frame loop (src/main.cpp:64-73: one compute() per frame) with `frames` frames in the device's queues: F
    contexts of one geometry, each on its own stream, take the frames of a stream in turn.  While frame i runs, frame i + 1 is
    already queued behind it on another stream, so a frame's short kernels (image preparation, guidance, reduction, merge) and the
    half-empty last round of its fused launch run beside the next frame's fused kernel instead of alone.  Results are those of
    the single-context calls, bit for bit (every context is an ordinary DispEst).  Measured on MI355X (profiles/r05): -13 % per
    frame at 1280 x 720 x 128, -24 % at 450 x 375 x 64, nothing at 1920 x 1080 x 256 (there the fused launches fill the chip).

        ring = FrameRing(l0, r0, maxDis, frames=2)
        for l, r in stream:
            done = ring.push(l, r)          # maps of the frame pushed `frames` calls earlier (None while the ring fills)
        for lm, rm in ring.flush(): ...     # the frames still in flight, oldest first
    """

    def __init__(self, l, r, d: int, frames: int = 2, *, dtype: str = "f32", device: int = 0, lr_check: bool = False,
                 seg_rows: int = 0):
        """Every context is told PSM_OPT_FRAMES_IN_FLIGHT = frames: the planner of the fused launches then cuts them for a shared
        device (450 x 375 x 64: 0.207 ms per frame against 0.22-0.23 without the hint; no result changes).  seg_rows > 0:
        PSM_OPT_SEG_ROWS of every context on top of that (round 5's first finding - one segment per launch, seg_rows = image height
        - is what the hint replaced: profiles/r05/exp_plan_model.txt)."""
        if frames < 1:
            raise ValueError("FrameRing: frames must be >= 1")
        self.ctx = [DispEst(l, r, d, dtype=dtype, device=device) for _ in range(frames)]
        for c in self.ctx:
            c.set_option(capi.PSM_OPT_ASYNC, 1)
            c.set_option(capi.PSM_OPT_FRAMES_IN_FLIGHT, frames)      # the planner cuts the launches for `frames` pairs at a time
            if seg_rows > 0:
                c.set_option(capi.PSM_OPT_SEG_ROWS, int(seg_rows))
        self._n = 0
        self._busy = [False] * frames
        self._lrc = lr_check

    def push(self, l, r):
        i = self._n % len(self.ctx)
        self._n += 1
        c = self.ctx[i]
        out = None
        if self._busy[i]:
            out = tuple(m.copy() for m in c.download_maps_wait())
        c.setInputImages(l, r)
        c.CostConst_GPU()
        c.CostFilter_GPU()
        c.DispSelect_device()
        if self._lrc:
            c.LRCheck_device()
        c.download_maps_async()
        self._busy[i] = True
        return out

    def flush(self):
        out = []
        F = len(self.ctx)
        for k in range(F):
            i = (self._n + k) % F
            if self._busy[i]:
                out.append(tuple(m.copy() for m in self.ctx[i].download_maps_wait()))
                self._busy[i] = False
        return out

    def close(self):
        for c in self.ctx:
            c.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

