"""ctypes binding of libssx.so (include/ssx.h).  The ONLY compute path of this package.

There is deliberately no Python/NumPy fallback: if libssx.so is missing or no gfx950 device is
visible, `load()` / `Context()` raise.  (Building: `python -m ssvio_amd.build`.)
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (SSX_LIB points the loader at another build of the same library: same-box A/B runs of tools/, never used by tests or bench)
LIB_PATH = os.environ.get("SSX_LIB") or os.path.join(_HERE, "libssx.so")

SSX_OK = 0
SSX_ERR_INVALID_ARG = -1
SSX_ERR_NO_DEVICE = -2
SSX_ERR_HIP = -3
SSX_ERR_CAPACITY = -4
SSX_ERR_UNSUPPORTED = -5
SSX_ERR_COMM = -6
SSX_BA_MAX_STATS = 128
SSX_VERSION = 120        # include/ssx.h

dbl_p = C.POINTER(C.c_double)
u8_p = C.POINTER(C.c_uint8)
i32_p = C.POINTER(C.c_int32)
f32_p = C.POINTER(C.c_float)


class SsxError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"ssx status {status}: {msg}")
        self.status = status


class Config(C.Structure):
    _fields_ = [("device", C.c_int), ("stream", C.c_void_p), ("max_width", C.c_int), ("max_height", C.c_int),
                ("cu_first", C.c_int), ("cu_count", C.c_int)]


class BaProblem(C.Structure):
    _fields_ = [("P", C.c_int32), ("poses", dbl_p), ("pose_fixed", u8_p),
                ("L", C.c_int32), ("points", dbl_p), ("point_fixed", u8_p),
                ("E", C.c_int32), ("edge_pose", i32_p), ("edge_point", i32_p), ("edge_uv", dbl_p),
                ("edge_cam", u8_p), ("K", C.c_double * 4), ("cam_ext", C.c_double * 14)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


class BaOptions(C.Structure):
    _fields_ = [("outer_rounds", C.c_int32), ("iters", C.c_int32), ("chi2_th", C.c_double),
                ("huber_delta", C.c_double), ("inlier_ratio", C.c_double), ("jac_mode", C.c_int32),
                ("allreduce", ALLREDUCE_FN), ("allreduce_user", C.c_void_p),
                ("rank", C.c_int32), ("world_size", C.c_int32), ("comm", C.c_void_p), ("collect_stats", C.c_int32), ("large_solver", C.c_int32)]


class BaResult(C.Structure):
    _fields_ = [("poses_out", dbl_p), ("points_out", dbl_p), ("edge_chi2", dbl_p), ("edge_outlier", u8_p),
                ("rounds", C.c_int32), ("n_iters", C.c_int32),
                ("iter_chi2", C.c_double * SSX_BA_MAX_STATS), ("iter_lambda", C.c_double * SSX_BA_MAX_STATS),
                ("iter_trials", C.c_int32 * SSX_BA_MAX_STATS),
                ("n_inliers", C.c_int32), ("n_outliers", C.c_int32),
                ("ms_total", C.c_float), ("ms_setup", C.c_float), ("ms_linearize", C.c_float), ("ms_schur", C.c_float),
                ("ms_linear_solution", C.c_float), ("ms_update", C.c_float), ("ms_reduce", C.c_float), ("ms_comm", C.c_float)]


class BaWindowUpdate(C.Structure):
    """ssx_ba_window_update (include/ssx.h): one keyframe replaced in one window of an ssx_ba_window_update_batch call"""
    _fields_ = [("pop", C.c_int32), ("push", C.c_int32), ("pop_kf_id", C.c_int64), ("kf_id", C.c_int64), ("pose7", dbl_p),
                ("pose_fixed", C.c_int32), ("n_new", C.c_int32), ("new_ids", C.POINTER(C.c_int64)), ("new_xyz", dbl_p), ("new_fixed", u8_p),
                ("new_slots_out", C.POINTER(C.c_int32)), ("n_obs", C.c_int32), ("reserved", C.c_int32), ("obs_lm", C.POINTER(C.c_int64)),
                ("obs_slot", C.POINTER(C.c_int32)), ("obs_uv", dbl_p), ("obs_cam", u8_p),
                ("n_remove_flags", C.c_int32), ("n_remove_lm", C.c_int32), ("remove_flags", u8_p), ("remove_lm_ids", C.POINTER(C.c_int64))]


class KeyPoint(C.Structure):
    """cv::KeyPoint layout (28 bytes): pt.x pt.y size angle response octave class_id."""
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("size", C.c_float), ("angle", C.c_float),
                ("response", C.c_float), ("octave", C.c_int32), ("class_id", C.c_int32)]


KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])

_lib = None


def load() -> C.CDLL:
    """Load libssx.so (no GPU needed just to load and inspect symbols)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SsxError(SSX_ERR_NO_DEVICE, f"{LIB_PATH} not built; run `python -m ssvio_amd.build` "
                                              "(the HIP library is the only compute path)")
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 / libhsa-runtime64 and
        # refuses to see the GPU when a different copy (/opt/rocm) was loaded first.  Import torch BEFORE
        # dlopen-ing libssx.so so that its DT_NEEDED libamdhip64.so.7 resolves to the copy torch loaded
        # (bench.py and the RCCL hook need torch.cuda / torch.distributed in the same process).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(LIB_PATH)
        lib.ssx_last_error.restype = C.c_char_p
        lib.ssx_last_error.argtypes = [C.c_void_p]
        lib.ssx_ctx_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
        lib.ssx_ctx_destroy.argtypes = [C.c_void_p]
        lib.ssx_ctx_destroy.restype = None
        lib.ssx_ctx_stream.restype = C.c_void_p
        lib.ssx_ctx_stream.argtypes = [C.c_void_p]
        lib.ssx_ctx_synchronize.argtypes = [C.c_void_p]
        # the ctypes mirrors of the header's structs must be the library's (ssx_abi_check, include/ssx.h)
        lib.ssx_abi_check.argtypes = [C.c_int] + [C.c_size_t] * 5
        if lib.ssx_abi_check(SSX_VERSION, C.sizeof(Config), C.sizeof(BaProblem), C.sizeof(BaOptions), C.sizeof(BaResult),
                             C.sizeof(BaWindowUpdate)) != SSX_OK:
            raise SsxError(SSX_ERR_UNSUPPORTED, f"{LIB_PATH} does not match the struct layouts of ssvio_amd/_lib.py (version "
                                                f"{lib.ssx_version()} vs {SSX_VERSION}): rebuild with `python -m ssvio_amd.build`")
        _lib = lib
    return _lib


def ptr(a, t):
    if a is None:
        return None
    return a.ctypes.data_as(t)


class Context:
    """One GPU + one HIP stream (ssx_ctx).  stream=None creates a private stream; pass
    `torch.cuda.current_stream().cuda_stream` to run on torch's stream (needed for the RCCL hook)."""

    def __init__(self, device: int = 0, stream: int | None = None, cu_first: int = 0, cu_count: int = 0):
        """cu_count > 0: the streams the ctx creates are restricted to CUs [cu_first, cu_first + cu_count) (ssx_config)"""
        self.lib = load()
        cfg = Config(device, C.c_void_p(stream) if stream else None, 0, 0, cu_first, cu_count)
        h = C.c_void_p()
        st = self.lib.ssx_ctx_create(C.byref(cfg), C.byref(h))
        if st != SSX_OK:
            raise SsxError(st, "ssx_ctx_create failed (no gfx950 device visible? there is no CPU fallback)")
        self.handle = h
        self.device = device

    def check(self, st):
        if st != SSX_OK:
            raise SsxError(st, self.lib.ssx_last_error(self.handle).decode())

    def synchronize(self):
        self.check(self.lib.ssx_ctx_synchronize(self.handle))

    @property
    def stream(self) -> int:
        return self.lib.ssx_ctx_stream(self.handle)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.ssx_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class OrbParams(C.Structure):
    """ORBextractor ctor arguments (orbextractor.hpp:44-45)."""
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32)]


class MatchParams(C.Structure):
    _fields_ = [("band_px", C.c_float), ("min_disp", C.c_float), ("max_disp", C.c_float),
                ("max_dist", C.c_int32), ("max_octave_diff", C.c_int32), ("scale_factor", C.c_float)]


class StereoRig(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("baseline", C.c_double)]


class StereoFrameOut(C.Structure):
    _fields_ = [("cap", C.c_int32), ("kpsL", C.c_void_p), ("kpsR", C.c_void_p), ("descL", u8_p), ("descR", u8_p),
                ("nL", C.c_int32), ("nR", C.c_int32), ("match_idx", i32_p), ("match_dist", i32_p),
                ("xyz", dbl_p), ("ok", u8_p), ("n_matched", C.c_int32), ("n_triangulated", C.c_int32)]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("calls", C.c_int32), ("total_ms", C.c_double)]


def profile_begin(ctx: "Context"):
    ctx.check(ctx.lib.ssx_profile_begin(ctx.handle))


def profile_end(ctx: "Context"):
    """-> {kernel name: (calls, total_ms)} measured with HIP events on the ctx stream."""
    arr = (KernelTime * 64)()
    n = C.c_int32(0)
    ctx.check(ctx.lib.ssx_profile_end(ctx.handle, arr, 64, C.byref(n)))
    return {arr[i].name.decode(): (arr[i].calls, arr[i].total_ms) for i in range(n.value)}
