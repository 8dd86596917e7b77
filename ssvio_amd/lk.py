"""Pyramidal Lucas-Kanade tracking: the host-side mirror of the two
cv::calcOpticalFlowPyrLK calls of the reference front-end (frontend.cpp:156-166 and :374-384)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import Context


class LkParams(C.Structure):
    _fields_ = [("win", C.c_int32), ("max_level", C.c_int32), ("max_iters", C.c_int32), ("eps", C.c_double),
                ("min_eig_threshold", C.c_float), ("use_initial_flow", C.c_int32)]


u8_p = C.POINTER(C.c_ubyte)
f32_p = C.POINTER(C.c_float)


def _img(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    assert a.ndim == 2
    return a


def calcOpticalFlowPyrLK(ctx: Context, prev, nxt, prev_pts, next_pts=None, winSize=11, maxLevel=3, maxCount=30,
                         epsilon=0.01, minEigThreshold=1e-4):
    """cv::calcOpticalFlowPyrLK(prev, next, prevPts, nextPts, status, err, Size(win, win), maxLevel,
    TermCriteria(COUNT+EPS, maxCount, epsilon), next_pts is None ? 0 : OPTFLOW_USE_INITIAL_FLOW, minEigThreshold).
    Returns (next_pts [n,2] f32, status [n] u8, err [n] f32, top_level).
    prev=None chains frames (ssx_lk_track_next): the previous image is the `nxt` image of the last call on ctx,
    whose pyramid is still on the device."""
    nxt = _img(nxt)
    chain = prev is None
    if not chain:
        prev = _img(prev)
        if prev.shape != nxt.shape:
            raise ValueError("calcOpticalFlowPyrLK: the two images must have the same size")
    pp = np.ascontiguousarray(prev_pts, dtype=np.float32).reshape(-1, 2)
    use_init = next_pts is not None
    npts = np.ascontiguousarray(next_pts, dtype=np.float32).reshape(-1, 2).copy() if use_init else pp.copy()
    if len(npts) != len(pp):
        raise ValueError("calcOpticalFlowPyrLK: prev_pts and next_pts differ in length")
    n = len(pp)
    status = np.zeros(n, np.uint8); err = np.zeros(n, np.float32)
    prm = LkParams(int(winSize), int(maxLevel), int(maxCount), float(epsilon), float(minEigThreshold), int(use_init))
    top = C.c_int32(0)
    lib = ctx.lib
    lib.ssx_lk_track.restype = C.c_int
    lib.ssx_lk_track_next.restype = C.c_int
    if chain:
        ctx.check(lib.ssx_lk_track_next(ctx.handle, nxt.ctypes.data_as(u8_p), nxt.strides[0], nxt.shape[0], nxt.shape[1], n,
                                        pp.ctypes.data_as(f32_p), npts.ctypes.data_as(f32_p), status.ctypes.data_as(u8_p),
                                        err.ctypes.data_as(f32_p), C.byref(prm), C.byref(top)))
        return npts, status, err, top.value
    ctx.check(lib.ssx_lk_track(ctx.handle, prev.ctypes.data_as(u8_p), prev.strides[0], nxt.ctypes.data_as(u8_p), nxt.strides[0],
                               prev.shape[0], prev.shape[1], n, pp.ctypes.data_as(f32_p), npts.ctypes.data_as(f32_p),
                               status.ctypes.data_as(u8_p), err.ctypes.data_as(f32_p), C.byref(prm), C.byref(top)))
    return npts, status, err, top.value


class LkJob(C.Structure):
    _fields_ = [("slot", C.c_int32), ("prev", u8_p), ("prev_stride", C.c_int32), ("next", u8_p), ("next_stride", C.c_int32), ("n", C.c_int32),
                ("prev_pts", f32_p), ("next_pts", f32_p), ("status", u8_p), ("err", f32_p)]


class PreparedTrackBatch:
    """ssx_lk_track_batch with everything a C caller holds between frames prepared once: the job structs, the points, and the images in
    PINNED memory (ssx_host_alloc) handed over with images_on_device = 1 -- what ssvio_amd/host/stream_batcher.cpp does per frame.
    run() is the library call alone; outputs in .outs [(next_pts, status, err)] (the guesses are restored before every run)."""

    def __init__(self, ctx: Context, jobs, winSize=11, maxLevel=3, maxCount=30, epsilon=0.01, minEigThreshold=1e-4):
        self.ctx = ctx
        lib = ctx.lib
        lib.ssx_host_alloc.restype = C.c_void_p; lib.ssx_host_alloc.argtypes = [C.c_size_t]
        lib.ssx_host_free.restype = None; lib.ssx_host_free.argtypes = [C.c_void_p]
        n = len(jobs)
        self.arr = (LkJob * n)()
        self.keep, self.outs, self.guess, self.pins = [], [], [], []
        use_init = any(j.get("next_pts") is not None for j in jobs)
        for i, j in enumerate(jobs):
            nxt = _img(j["next"])
            self.rows, self.cols = nxt.shape
            pin = lib.ssx_host_alloc(nxt.size)
            if not pin:
                raise MemoryError("ssx_host_alloc")
            C.memmove(pin, nxt.ctypes.data, nxt.size)
            self.pins.append(pin)
            pp = np.ascontiguousarray(j["prev_pts"], dtype=np.float32).reshape(-1, 2)
            g = np.ascontiguousarray(j["next_pts"], dtype=np.float32).reshape(-1, 2).copy() if j.get("next_pts") is not None else pp.copy()
            npts = g.copy(); st = np.zeros(len(pp), np.uint8); er = np.zeros(len(pp), np.float32)
            a = self.arr[i]
            a.slot = int(j["slot"]); a.prev = None; a.prev_stride = 0
            a.next = C.cast(pin, u8_p); a.next_stride = self.cols
            a.n = len(pp); a.prev_pts = pp.ctypes.data_as(f32_p); a.next_pts = npts.ctypes.data_as(f32_p)
            a.status = st.ctypes.data_as(u8_p); a.err = er.ctypes.data_as(f32_p)
            self.keep.append(pp); self.outs.append((npts, st, er)); self.guess.append(g)
        self.n = n
        self.prm = LkParams(int(winSize), int(maxLevel), int(maxCount), float(epsilon), float(minEigThreshold), int(use_init))
        lib.ssx_lk_track_batch.restype = C.c_int
        lib.ssx_lk_track_batch.argtypes = [C.c_void_p, C.c_int32, C.POINTER(LkJob), C.c_int32, C.c_int32, C.POINTER(LkParams), C.c_int32]

    def run(self):
        for (npts, _, _), g in zip(self.outs, self.guess):
            npts[:] = g
        self.ctx.check(self.ctx.lib.ssx_lk_track_batch(self.ctx.handle, self.n, self.arr, self.rows, self.cols, C.byref(self.prm), 1))
        return self.outs

    def close(self):
        for p_ in self.pins:
            self.ctx.lib.ssx_host_free(p_)
        self.pins = []


def track_batch(ctx: Context, jobs, winSize=11, maxLevel=3, maxCount=30, epsilon=0.01, minEigThreshold=1e-4, images_on_device=False):
    """ssx_lk_track_batch: jobs = [dict(slot, prev (image or None = chained to the slot's last job), next, prev_pts, next_pts or None)];
    every job with the same image size.  -> [(next_pts, status, err)] per job.  images_on_device: the arrays' memory is readable by
    the GPU (pinned)."""
    n = len(jobs)
    arr = (LkJob * n)()
    keep, outs = [], []
    use_init = any(j.get("next_pts") is not None for j in jobs)
    rows = cols = None
    for i, j in enumerate(jobs):
        nxt = _img(j["next"]); prev = None if j.get("prev") is None else _img(j["prev"])
        rows, cols = nxt.shape
        pp = np.ascontiguousarray(j["prev_pts"], dtype=np.float32).reshape(-1, 2)
        npts = np.ascontiguousarray(j["next_pts"], dtype=np.float32).reshape(-1, 2).copy() if j.get("next_pts") is not None else pp.copy()
        st = np.zeros(len(pp), np.uint8); er = np.zeros(len(pp), np.float32)
        a = arr[i]
        a.slot = int(j["slot"])
        a.prev = None if prev is None else prev.ctypes.data_as(u8_p); a.prev_stride = 0 if prev is None else prev.strides[0]
        a.next = nxt.ctypes.data_as(u8_p); a.next_stride = nxt.strides[0]
        a.n = len(pp); a.prev_pts = pp.ctypes.data_as(f32_p); a.next_pts = npts.ctypes.data_as(f32_p)
        a.status = st.ctypes.data_as(u8_p); a.err = er.ctypes.data_as(f32_p)
        keep.append((nxt, prev, pp)); outs.append((npts, st, er))
    prm = LkParams(int(winSize), int(maxLevel), int(maxCount), float(epsilon), float(minEigThreshold), int(use_init))
    ctx.lib.ssx_lk_track_batch.restype = C.c_int
    ctx.lib.ssx_lk_track_batch.argtypes = [C.c_void_p, C.c_int32, C.POINTER(LkJob), C.c_int32, C.c_int32, C.POINTER(LkParams), C.c_int32]
    ctx.check(ctx.lib.ssx_lk_track_batch(ctx.handle, n, arr, rows, cols, C.byref(prm), 1 if images_on_device else 0))
    return outs


def stage_level(ctx: Context, which, level):
    r = C.c_int32(0); c = C.c_int32(0)
    ctx.check(ctx.lib.ssx_lk_stage_level(ctx.handle, which, level, None, 0, C.byref(r), C.byref(c)))
    out = np.zeros((r.value, c.value), np.uint8)
    ctx.check(ctx.lib.ssx_lk_stage_level(ctx.handle, which, level, out.ctypes.data_as(u8_p), out.size, C.byref(r), C.byref(c)))
    return out


def stage_deriv(ctx: Context, level):
    r = C.c_int32(0); c = C.c_int32(0)
    ctx.check(ctx.lib.ssx_lk_stage_deriv(ctx.handle, level, None, 0, C.byref(r), C.byref(c)))
    out = np.zeros((r.value, c.value, 2), np.int16)
    ctx.check(ctx.lib.ssx_lk_stage_deriv(ctx.handle, level, out.ctypes.data_as(C.c_void_p), out.size, C.byref(r), C.byref(c)))
    return out
