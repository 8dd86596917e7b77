"""ORB extraction / stereo matching / triangulation through libssx.so.

`ORBextractor` mirrors the reference class surface (include/ssvio/orbextractor.hpp:44-59): same constructor
arguments, `Detect(image, mask)` and `DetectAndCompute(image, mask)` with numpy arrays standing in for
cv::Mat / std::vector<cv::KeyPoint> (KP_DTYPE has the binary layout of cv::KeyPoint).  Every method calls the
HIP library; nothing is computed in Python.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import KP_DTYPE, Context, MatchParams, OrbParams, StereoFrameOut, StereoRig, dbl_p, i32_p, ptr, u8_p


def _img(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if a.ndim != 2:
        raise ValueError("CV_8UC1 image expected")   # the reference asserts image.type() == CV_8UC1
    return a


class ORBextractor:
    """ssvio::ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)."""

    def __init__(self, ctx: Context, nfeatures=2000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7):
        self.ctx = ctx
        self.prm = OrbParams(int(nfeatures), float(scaleFactor), int(nlevels), int(iniThFAST), int(minThFAST))

    def _cap(self):
        # a level can return up to max(budget + 3, 4 * nIni) keypoints (first quadtree subdivision), see plan() in orb.hip
        return self.prm.nfeatures + 260 * self.prm.nlevels + 64

    def Detect(self, image, mask=None):
        """ORBextractor::Detect (orbextractor.cpp:755-842): returns keypoints; empty image -> empty result."""
        image = np.asarray(image)
        if image.size == 0:
            return np.zeros(0, dtype=KP_DTYPE)
        image = _img(image)
        mask = None if mask is None else _img(mask)
        cap = self._cap()
        kps = np.zeros(cap, dtype=KP_DTYPE)
        n = C.c_int32(0)
        self.ctx.check(self.ctx.lib.ssx_orb_detect(
            self.ctx.handle, ptr(image, u8_p), image.strides[0], image.shape[0], image.shape[1], ptr(mask, u8_p),
            0 if mask is None else mask.strides[0], C.byref(self.prm), cap, kps.ctypes.data_as(C.c_void_p), C.byref(n)))
        return kps[:n.value].copy()

    def DetectBoxes(self, image, boxes):
        """ssx_orb_detect_boxes: ORBextractor::Detect under the mask of FrontEnd::DetectFeatures (frontend.cpp:302-312) given as its
        rectangles -- boxes [n, 4] int32 (x0, y0, x1, y1), corners inclusive; the mask is rasterised on the device."""
        image = np.asarray(image)
        if image.size == 0:
            return np.zeros(0, dtype=KP_DTYPE)
        image = _img(image)
        boxes = np.ascontiguousarray(boxes, dtype=np.int32).reshape(-1, 4)
        cap = self._cap()
        kps = np.zeros(cap, dtype=KP_DTYPE)
        n = C.c_int32(0)
        self.ctx.check(self.ctx.lib.ssx_orb_detect_boxes(
            self.ctx.handle, ptr(image, u8_p), image.strides[0], image.shape[0], image.shape[1], boxes.ctypes.data_as(C.POINTER(C.c_int32)),
            len(boxes), C.byref(self.prm), cap, kps.ctypes.data_as(C.c_void_p), C.byref(n)))
        return kps[:n.value].copy()

    def DetectBoxesBatch(self, images, boxes_list):
        """ssx_orb_detect_boxes_batch: n images of one size (host arrays: staged by the library), each with its own rectangles, in
        one call -> list of keypoint arrays (per image the result of DetectBoxes)."""
        class Job(C.Structure):
            _fields_ = [("img", u8_p), ("stride", C.c_int32), ("boxes", C.POINTER(C.c_int32)), ("n_boxes", C.c_int32), ("cap", C.c_int32),
                        ("kps_out", C.c_void_p), ("n_out", C.POINTER(C.c_int32))]
        n = len(images)
        arr = (Job * n)()
        keep, outs = [], []
        cap = self._cap()
        for i, (im, bx) in enumerate(zip(images, boxes_list)):
            im = _img(im); bx = np.ascontiguousarray(bx, dtype=np.int32).reshape(-1, 4)
            kps = np.zeros(cap, dtype=KP_DTYPE); cnt = (C.c_int32 * 1)(0)
            a = arr[i]
            a.img = ptr(im, u8_p); a.stride = im.strides[0]; a.boxes = bx.ctypes.data_as(C.POINTER(C.c_int32)); a.n_boxes = len(bx)
            a.cap = cap; a.kps_out = kps.ctypes.data_as(C.c_void_p); a.n_out = C.cast(cnt, C.POINTER(C.c_int32))
            keep.append((im, bx)); outs.append((kps, cnt))
        rows, cols = keep[0][0].shape
        self.ctx.lib.ssx_orb_detect_boxes_batch.restype = C.c_int32
        self.ctx.lib.ssx_orb_detect_boxes_batch.argtypes = [C.c_void_p, C.c_int32, C.POINTER(Job), C.c_int32, C.c_int32, C.POINTER(OrbParams), C.c_int32]
        self.ctx.check(self.ctx.lib.ssx_orb_detect_boxes_batch(self.ctx.handle, n, arr, rows, cols, C.byref(self.prm), 0))
        return [k[:c[0]].copy() for k, c in outs]

    def DetectAndCompute(self, image, mask=None):
        """ORBextractor::DetectAndCompute (orbextractor.cpp:687-753): (keypoints, N x 32 uint8 descriptors)."""
        image = np.asarray(image)
        if image.size == 0:
            return np.zeros(0, dtype=KP_DTYPE), np.zeros((0, 32), np.uint8)
        image = _img(image)
        mask = None if mask is None else _img(mask)
        cap = self._cap()
        kps = np.zeros(cap, dtype=KP_DTYPE); desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int32(0)
        self.ctx.check(self.ctx.lib.ssx_orb_extract(
            self.ctx.handle, ptr(image, u8_p), image.strides[0], image.shape[0], image.shape[1], ptr(mask, u8_p),
            0 if mask is None else mask.strides[0], C.byref(self.prm), cap, kps.ctypes.data_as(C.c_void_p),
            ptr(desc, u8_p), C.byref(n)))
        return kps[:n.value].copy(), desc[:n.value].copy()

    def ScreenAndComputeKPsParams_CalcDescriptors(self, image, keypoints):
        """ORBextractor::ScreenAndComputeKPsParams followed by CalcDescriptors (orbextractor.cpp:844-991), the
        loop-closing use (loopclosing.cpp:622-629): -> (kept keypoints with angle/size, N x 32 descriptors)."""
        image = np.asarray(image)
        keypoints = np.ascontiguousarray(keypoints, dtype=KP_DTYPE)
        if image.size == 0 or len(keypoints) == 0:
            return np.zeros(0, dtype=KP_DTYPE), np.zeros((0, 32), np.uint8)
        image = _img(image)
        n_in = len(keypoints)
        kps = np.zeros(n_in, dtype=KP_DTYPE); desc = np.zeros((n_in, 32), np.uint8)
        n = C.c_int32(0)
        self.ctx.check(self.ctx.lib.ssx_orb_describe_at(
            self.ctx.handle, ptr(image, u8_p), image.strides[0], image.shape[0], image.shape[1], C.byref(self.prm),
            keypoints.ctypes.data_as(C.c_void_p), n_in, kps.ctypes.data_as(C.c_void_p), ptr(desc, u8_p), C.byref(n)))
        return kps[:n.value].copy(), desc[:n.value].copy()

    # parity hooks ------------------------------------------------------------------------------
    def stage_level(self, level, blurred=False, image=0):
        r = C.c_int32(0); c = C.c_int32(0)
        self.ctx.check(self.ctx.lib.ssx_orb_stage_level(self.ctx.handle, image, level, int(blurred), None, 0, C.byref(r), C.byref(c)))
        out = np.zeros((r.value, c.value), np.uint8)
        self.ctx.check(self.ctx.lib.ssx_orb_stage_level(self.ctx.handle, image, level, int(blurred), ptr(out, u8_p), out.size,
                                                        C.byref(r), C.byref(c)))
        return out

    def stage_candidates(self, level, image=0, cap=1 << 17):
        out = np.zeros(cap, dtype=KP_DTYPE); n = C.c_int32(0)
        self.ctx.check(self.ctx.lib.ssx_orb_stage_candidates(self.ctx.handle, image, level, cap,
                                                             out.ctypes.data_as(C.c_void_p), C.byref(n)))
        return out[:n.value].copy()


def match_params(band_px=2.0, min_disp=0.0, max_disp=120.0, max_dist=80, max_octave_diff=1, scale_factor=1.2):
    return MatchParams(band_px, min_disp, max_disp, max_dist, max_octave_diff, scale_factor)


KITTI_K = (718.856, 718.856, 607.1928, 185.2157)   # fx fy cx cy  (/root/reference/config/kitti_00.yaml:3-6)
KITTI_BASELINE = 386.1448 / KITTI_K[0]               # bf / fx (kitti_00.yaml:26, system.cpp:69-70)


def stereo_rig(K=None, baseline=None):
    K = KITTI_K if K is None else K
    return StereoRig(float(K[0]), float(K[1]), float(K[2]), float(K[3]), float(KITTI_BASELINE if baseline is None else baseline))


def stereo_match(ctx: Context, kL, dL, kR, dR, prm=None):
    prm = prm or match_params()
    kL = np.ascontiguousarray(kL, dtype=KP_DTYPE); kR = np.ascontiguousarray(kR, dtype=KP_DTYPE)
    dL = np.ascontiguousarray(dL, dtype=np.uint8).reshape(-1, 32); dR = np.ascontiguousarray(dR, dtype=np.uint8).reshape(-1, 32)
    idx = np.zeros(len(kL), np.int32); dist = np.zeros(len(kL), np.int32)
    ctx.check(ctx.lib.ssx_stereo_match(ctx.handle, kL.ctypes.data_as(C.c_void_p), ptr(dL, u8_p), len(kL),
                                       kR.ctypes.data_as(C.c_void_p), ptr(dR, u8_p), len(kR), C.byref(prm),
                                       ptr(idx, i32_p), ptr(dist, i32_p)))
    return idx, dist


def bf_match(ctx: Context, dq, dt):
    dq = np.ascontiguousarray(dq, dtype=np.uint8).reshape(-1, 32); dt = np.ascontiguousarray(dt, dtype=np.uint8).reshape(-1, 32)
    idx = np.zeros(len(dq), np.int32); dist = np.zeros(len(dq), np.int32)
    ctx.check(ctx.lib.ssx_bf_match(ctx.handle, ptr(dq, u8_p), len(dq), ptr(dt, u8_p), len(dt), ptr(idx, i32_p), ptr(dist, i32_p)))
    return idx, dist


def triangulate(ctx: Context, uvL, uvR, rig=None, T_wc=None):
    rig = rig or stereo_rig()
    uvL = np.ascontiguousarray(uvL, dtype=np.float64).reshape(-1, 2); uvR = np.ascontiguousarray(uvR, dtype=np.float64).reshape(-1, 2)
    n = len(uvL)
    xyz = np.zeros((n, 3)); ok = np.zeros(n, np.uint8)
    T = None if T_wc is None else np.ascontiguousarray(T_wc, dtype=np.float64)
    ctx.check(ctx.lib.ssx_triangulate(ctx.handle, n, ptr(uvL, dbl_p), ptr(uvR, dbl_p), C.byref(rig), ptr(T, dbl_p),
                                      ptr(xyz, dbl_p), ptr(ok, u8_p)))
    return xyz, ok


def triangulate_batch(ctx: Context, jobs):
    """ssx_triangulate_batch: jobs = [dict(uvL, uvR, rig=None, T_wc=None)] -> [(xyz, ok)], one launch for all."""
    class Job(C.Structure):
        _fields_ = [("n", C.c_int32), ("uvL", dbl_p), ("uvR", dbl_p), ("rig", C.POINTER(StereoRig)), ("T_wc", dbl_p), ("xyz_out", dbl_p), ("ok_out", u8_p)]
    n = len(jobs)
    arr = (Job * n)()
    keep, outs = [], []
    for i, j in enumerate(jobs):
        rig = j.get("rig") or stereo_rig()
        uvL = np.ascontiguousarray(j["uvL"], dtype=np.float64).reshape(-1, 2); uvR = np.ascontiguousarray(j["uvR"], dtype=np.float64).reshape(-1, 2)
        T = None if j.get("T_wc") is None else np.ascontiguousarray(j["T_wc"], dtype=np.float64)
        xyz = np.zeros((len(uvL), 3)); ok = np.zeros(len(uvL), np.uint8)
        a = arr[i]
        a.n = len(uvL); a.uvL = ptr(uvL, dbl_p); a.uvR = ptr(uvR, dbl_p); a.rig = C.pointer(rig); a.T_wc = ptr(T, dbl_p); a.xyz_out = ptr(xyz, dbl_p); a.ok_out = ptr(ok, u8_p)
        keep.append((rig, uvL, uvR, T)); outs.append((xyz, ok))
    ctx.lib.ssx_triangulate_batch.restype = C.c_int32
    ctx.lib.ssx_triangulate_batch.argtypes = [C.c_void_p, C.c_int32, C.POINTER(Job)]
    ctx.check(ctx.lib.ssx_triangulate_batch(ctx.handle, n, arr))
    return outs


class _FrameBuffers:
    def __init__(self, cap):
        # (np.empty: the library writes the first nL / nR entries and result() returns exactly those)
        self.kL = np.empty(cap, dtype=KP_DTYPE); self.kR = np.empty(cap, dtype=KP_DTYPE)
        self.dL = np.empty((cap, 32), np.uint8); self.dR = np.empty((cap, 32), np.uint8)
        self.idx = np.empty(cap, np.int32); self.dist = np.empty(cap, np.int32)
        self.xyz = np.empty((cap, 3)); self.ok = np.empty(cap, np.uint8)
        o = StereoFrameOut()
        o.cap = cap
        o.kpsL = self.kL.ctypes.data_as(C.c_void_p); o.kpsR = self.kR.ctypes.data_as(C.c_void_p)
        o.descL = ptr(self.dL, u8_p); o.descR = ptr(self.dR, u8_p)
        o.match_idx = ptr(self.idx, i32_p); o.match_dist = ptr(self.dist, i32_p)
        o.xyz = ptr(self.xyz, dbl_p); o.ok = ptr(self.ok, u8_p)
        self.out = o

    def result(self):
        o = self.out
        nL, nR = o.nL, o.nR
        return dict(kL=self.kL[:nL].copy(), dL=self.dL[:nL].copy(), kR=self.kR[:nR].copy(), dR=self.dR[:nR].copy(),
                    match_idx=self.idx[:nL].copy(), match_dist=self.dist[:nL].copy(), xyz=self.xyz[:nL].copy(),
                    ok=self.ok[:nL].copy(), n_matched=o.n_matched, n_triangulated=o.n_triangulated)


def stereo_frame(ctx: Context, imgL, imgR, orb: OrbParams | None = None, mp=None, rig=None, T_wc=None):
    """extract L+R, row-band match, triangulate -- one call (ssx_stereo_frame)."""
    imgL = _img(imgL); imgR = _img(imgR)
    assert imgL.shape == imgR.shape and imgL.strides == imgR.strides
    orb = orb or OrbParams(2000, 1.2, 8, 20, 7)
    mp = mp or match_params(scale_factor=orb.scale_factor)
    rig = rig or stereo_rig()
    fb = _FrameBuffers(orb.nfeatures + 260 * orb.nlevels + 64)
    T = None if T_wc is None else np.ascontiguousarray(T_wc, dtype=np.float64)
    ctx.check(ctx.lib.ssx_stereo_frame(ctx.handle, ptr(imgL, u8_p), ptr(imgR, u8_p), imgL.strides[0], imgL.shape[0],
                                       imgL.shape[1], C.byref(orb), C.byref(mp), C.byref(rig), ptr(T, dbl_p), C.byref(fb.out)))
    return fb.result()


def stereo_batch_dev(ctx: Context, imgs_dev_ptr: int, pairs: int, stride: int, rows: int, cols: int, orb=None, mp=None, rig=None):
    """`imgs_dev_ptr`: DEVICE pointer to [pairs][2][rows][stride] u8 (e.g. a torch.uint8 CUDA tensor's data_ptr()).
    Returns counts[pairs, 4] = nL, nR, n_matched, n_triangulated.  Results stay on the device."""
    orb = orb or OrbParams(2000, 1.2, 8, 20, 7)
    mp = mp or match_params(scale_factor=orb.scale_factor)
    rig = rig or stereo_rig()
    counts = np.zeros((pairs, 4), np.int32)
    ctx.check(ctx.lib.ssx_stereo_batch_dev(ctx.handle, pairs, C.c_void_p(imgs_dev_ptr), stride, rows, cols, C.byref(orb),
                                           C.byref(mp), C.byref(rig), ptr(counts, i32_p)))
    return counts


class StereoStream:
    """ssx_stereo_batch_host / ssx_stereo_batch_counts: a new batch of host images per call (pinned memory -> asynchronous upload on
    the ctx's copy stream, double-buffered on the device), counts of the last batch on request."""

    def __init__(self, ctx: Context, pairs: int, rows: int, cols: int, stride: int | None = None, orb=None, mp=None, rig=None):
        self.ctx, self.pairs, self.rows, self.cols, self.stride = ctx, pairs, rows, cols, stride or cols
        self.orb = orb or OrbParams(2000, 1.2, 8, 20, 7)
        self.mp = mp or match_params(scale_factor=self.orb.scale_factor)
        self.rig = rig or stereo_rig()
        self.counts = np.zeros((pairs, 4), np.int32)
        ctx.lib.ssx_stereo_batch_host.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(OrbParams),
                                                  C.POINTER(MatchParams), C.POINTER(StereoRig)]
        ctx.lib.ssx_stereo_batch_counts.argtypes = [C.c_void_p, i32_p]
        ctx.lib.ssx_stereo_batch_upload.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
        ctx.lib.ssx_stereo_batch_run.argtypes = [C.c_void_p, C.POINTER(OrbParams), C.POINTER(MatchParams), C.POINTER(StereoRig)]

    def enqueue(self, host_ptr: int):
        """host_ptr: address of [pairs][2][rows][stride] u8 in (pinned) host memory, e.g. torch.Tensor.pin_memory().data_ptr()"""
        self.ctx.check(self.ctx.lib.ssx_stereo_batch_host(self.ctx.handle, self.pairs, C.c_void_p(host_ptr), self.stride, self.rows, self.cols,
                                                          C.byref(self.orb), C.byref(self.mp), C.byref(self.rig)))

    def upload(self, host_ptr: int):
        """start the upload of a batch (one may be uploaded ahead of the one being run)"""
        self.ctx.check(self.ctx.lib.ssx_stereo_batch_upload(self.ctx.handle, self.pairs, C.c_void_p(host_ptr), self.stride, self.rows, self.cols))

    def run(self):
        """enqueue the front-end on the oldest uploaded batch"""
        self.ctx.check(self.ctx.lib.ssx_stereo_batch_run(self.ctx.handle, C.byref(self.orb), C.byref(self.mp), C.byref(self.rig)))

    def wait_counts(self):
        self.ctx.check(self.ctx.lib.ssx_stereo_batch_counts(self.ctx.handle, ptr(self.counts, i32_p)))
        return self.counts


def stereo_batch_enqueue(ctx: Context):
    ctx.check(ctx.lib.ssx_stereo_batch_enqueue(ctx.handle))


def stereo_batch_fetch(ctx: Context, pair: int, cap: int):
    fb = _FrameBuffers(cap)
    ctx.check(ctx.lib.ssx_stereo_batch_fetch(ctx.handle, pair, C.byref(fb.out)))
    return fb.result()
