"""oracle/pyoracle.py -- TEST INFRASTRUCTURE ONLY.

ctypes bindings of
  * oracle/liboracle.so            our CPU restatement (oracle/src/*.cpp), and
  * oracle/_ref/libssvio_ref.so    the real reference arithmetic (oracle/ref_driver.cpp + /root/reference
                                   sources, built by oracle/Makefile; prebuilt file travels to the GPU box).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_SO = os.path.join(_HERE, "liboracle.so")
_REF_SO = os.path.join(_HERE, "_ref", "libssvio_ref.so")

dbl_p = C.POINTER(C.c_double)
u8_p = C.POINTER(C.c_ubyte)
i32_p = C.POINTER(C.c_int32)
f32_p = C.POINTER(C.c_float)


def _p(a, t):
    if a is None:
        return None
    return a.ctypes.data_as(t)


def build(force=False):
    """Compile liboracle.so (always possible) and, when /root/reference exists, the reference .so."""
    srcs = [os.path.join(_HERE, "src", f) for f in os.listdir(os.path.join(_HERE, "src"))]
    newest = max(os.path.getmtime(s) for s in srcs)
    if force or not os.path.exists(_ORACLE_SO) or os.path.getmtime(_ORACLE_SO) < newest:
        subprocess.check_call(["make", "-C", _HERE, "oracle"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/include/ssvio") and (
            force or not os.path.exists(_REF_SO)
            or os.path.getmtime(_REF_SO) < os.path.getmtime(os.path.join(_HERE, "ref_driver.cpp"))):
        subprocess.check_call(["make", "-C", _HERE, "-j8", "ref"], stdout=subprocess.DEVNULL)


class KeyPoint(C.Structure):
    """cv::KeyPoint layout (28 bytes)."""
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("size", C.c_float), ("angle", C.c_float),
                ("response", C.c_float), ("octave", C.c_int32), ("class_id", C.c_int32)]


KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28


class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int), ("scale_factor", C.c_float), ("nlevels", C.c_int),
                ("ini_th_fast", C.c_int), ("min_th_fast", C.c_int)]


class MatchParams(C.Structure):
    _fields_ = [("band_px", C.c_float), ("min_disp", C.c_float), ("max_disp", C.c_float),
                ("max_dist", C.c_int), ("max_octave_diff", C.c_int), ("scale_factor", C.c_float)]


class BaOptions(C.Structure):
    _fields_ = [("outer_rounds", C.c_int), ("iters", C.c_int), ("chi2_th", C.c_double),
                ("huber_delta", C.c_double), ("inlier_ratio", C.c_double), ("jac_mode", C.c_int)]


_oracle = None
_ref = None


def oracle_lib():
    global _oracle
    if _oracle is None:
        build()
        _oracle = C.CDLL(_ORACLE_SO)
        for name, rt in (("orc_ic_angle", C.c_float), ("orc_fast_atan2", C.c_float),
                         ("orc_brief_pattern", C.POINTER(C.c_int8))):
            if hasattr(_oracle, name):
                getattr(_oracle, name).restype = rt
    return _oracle


def have_ref():
    return os.path.exists(_REF_SO) or os.path.isdir("/root/reference/include/ssvio")


def ref_lib():
    global _ref
    if _ref is None:
        build()
        _ref = C.CDLL(_REF_SO)
    return _ref


# ------------------------------------------------------------------ BA
def _ba_args(pr, poses, points):
    return (pr["P"], _p(poses, dbl_p), _p(pr["pose_fixed"], u8_p), pr["L"], _p(points, dbl_p),
            _p(pr["point_fixed"], u8_p), pr["E"], _p(pr["edge_pose"], i32_p), _p(pr["edge_point"], i32_p),
            _p(pr["edge_uv"], dbl_p), _p(pr["edge_cam"], u8_p), _p(pr["K"], dbl_p), _p(pr["cam_ext"], dbl_p))


def ba_solve(pr, which="oracle", outer_rounds=5, iters=10, chi2_th=5.891, huber_delta=5.891,
             inlier_ratio=0.7, jac_mode=1):
    """Run the BA on a problem dict (tools.synth.make_ba_problem layout). which = oracle | ref."""
    poses = np.ascontiguousarray(pr["poses"], dtype=np.float64).copy()
    points = np.ascontiguousarray(pr["points"], dtype=np.float64).copy()
    E = pr["E"]
    chi2 = np.zeros(E)
    cap = outer_rounds * iters + 8
    s_chi = np.zeros(cap); s_lam = np.zeros(cap); s_tr = np.zeros(cap, dtype=np.int32)
    n = C.c_int(0)
    if which == "ref":
        rounds = ref_lib().ref_ba_solve(*_ba_args(pr, poses, points), outer_rounds, iters,
                                        C.c_double(chi2_th), C.c_double(huber_delta), C.c_double(inlier_ratio),
                                        _p(chi2, dbl_p), cap, C.byref(n), _p(s_chi, dbl_p), _p(s_lam, dbl_p),
                                        _p(s_tr, i32_p))
        outl = (chi2 > chi2_th).astype(np.uint8)
    else:
        opt = BaOptions(outer_rounds, iters, chi2_th, huber_delta, inlier_ratio, jac_mode)
        outl = np.zeros(E, dtype=np.uint8)
        rounds = oracle_lib().orc_ba_solve(*_ba_args(pr, poses, points), C.byref(opt), _p(chi2, dbl_p),
                                           _p(outl, u8_p), cap, C.byref(n), _p(s_chi, dbl_p),
                                           _p(s_lam, dbl_p), _p(s_tr, i32_p))
    k = n.value
    return dict(rounds=rounds, poses=poses, points=points, edge_chi2=chi2, edge_outlier=outl,
                chi2=s_chi[:k].copy(), lam=s_lam[:k].copy(), trials=s_tr[:k].copy())


def ba_linearize(pr, huber_delta=5.891, jac_mode=0):
    P, L, E = pr["P"], pr["L"], pr["E"]
    Hpp = np.zeros((P, 6, 6)); bp = np.zeros((P, 6)); Hll = np.zeros((L, 3, 3)); bl = np.zeros((L, 3))
    Hpl = np.zeros((E, 6, 3)); err = np.zeros((E, 2)); chi = C.c_double(0)
    poses = np.ascontiguousarray(pr["poses"]); points = np.ascontiguousarray(pr["points"])
    oracle_lib().orc_ba_linearize(*_ba_args(pr, poses, points), C.c_double(huber_delta), jac_mode,
                                  _p(Hpp, dbl_p), _p(bp, dbl_p), _p(Hll, dbl_p), _p(bl, dbl_p),
                                  _p(Hpl, dbl_p), _p(err, dbl_p), C.byref(chi))
    return dict(Hpp=Hpp, bp=bp, Hll=Hll, bl=bl, Hpl=Hpl, err=err, chi2=chi.value)


def edge_eval(pose, p, uv, K, ext, huber_delta=5.891, which="oracle", jac_mode=1):
    pose = np.ascontiguousarray(pose, dtype=np.float64); p = np.ascontiguousarray(p, dtype=np.float64)
    uv = np.ascontiguousarray(uv, dtype=np.float64); K = np.ascontiguousarray(K, dtype=np.float64)
    ext = np.ascontiguousarray(ext, dtype=np.float64)
    e = np.zeros(2); Ji = np.zeros((2, 6)); Jj = np.zeros((2, 3)); chi = C.c_double(0); rho = np.zeros(3)
    if which == "ref":
        ref_lib().ref_edge_eval(_p(pose, dbl_p), _p(p, dbl_p), _p(uv, dbl_p), _p(K, dbl_p), _p(ext, dbl_p),
                                C.c_double(huber_delta), _p(e, dbl_p), _p(Ji, dbl_p), _p(Jj, dbl_p),
                                C.byref(chi), _p(rho, dbl_p))
    else:
        oracle_lib().orc_edge_eval(_p(pose, dbl_p), _p(p, dbl_p), _p(uv, dbl_p), _p(K, dbl_p), _p(ext, dbl_p),
                                   C.c_double(huber_delta), jac_mode, _p(e, dbl_p), _p(Ji, dbl_p),
                                   _p(Jj, dbl_p), C.byref(chi), _p(rho, dbl_p))
    return dict(e=e, Ji=Ji, Jj=Jj, chi2=chi.value, rho=rho)


def se3_exp(a, which="oracle"):
    a = np.ascontiguousarray(a, dtype=np.float64); out = np.zeros(7)
    (ref_lib().ref_se3_exp if which == "ref" else oracle_lib().orc_se3_exp)(_p(a, dbl_p), _p(out, dbl_p))
    return out


def pose_oplus(pose, d, which="oracle"):
    pose = np.ascontiguousarray(pose, dtype=np.float64); d = np.ascontiguousarray(d, dtype=np.float64)
    out = np.zeros(7)
    (ref_lib().ref_pose_oplus if which == "ref" else oracle_lib().orc_pose_oplus)(
        _p(pose, dbl_p), _p(d, dbl_p), _p(out, dbl_p))
    return out


def se3_act(pose, p, which="oracle"):
    pose = np.ascontiguousarray(pose, dtype=np.float64); p = np.ascontiguousarray(p, dtype=np.float64)
    out = np.zeros(3)
    (ref_lib().ref_se3_act if which == "ref" else oracle_lib().orc_se3_act)(
        _p(pose, dbl_p), _p(p, dbl_p), _p(out, dbl_p))
    return out


def pose_only(pr, which="oracle", rounds=4, iters=10, chi2_th=5.991, huber_delta=1.0):
    pose = np.ascontiguousarray(pr["pose"], dtype=np.float64).copy()
    M = pr["M"]
    inl = np.zeros(M, dtype=np.uint8)
    xyz = np.ascontiguousarray(pr["xyz"]); uv = np.ascontiguousarray(pr["uv"]); K = np.ascontiguousarray(pr["K"])
    if which == "ref":
        n = ref_lib().ref_pose_only(_p(pose, dbl_p), _p(K, dbl_p), M, _p(xyz, dbl_p), _p(uv, dbl_p), rounds,
                                    iters, C.c_double(chi2_th), _p(inl, u8_p))
    else:
        n = oracle_lib().orc_pose_only(_p(pose, dbl_p), _p(K, dbl_p), M, _p(xyz, dbl_p), _p(uv, dbl_p), rounds,
                                       iters, C.c_double(chi2_th), C.c_double(huber_delta), _p(inl, u8_p))
    return dict(pose=pose, inliers=inl, n_inliers=n)


def triangulate(uvL, uvR, K, baseline, T_wc=None, which="oracle"):
    uvL = np.ascontiguousarray(uvL, dtype=np.float64); uvR = np.ascontiguousarray(uvR, dtype=np.float64)
    n = uvL.shape[0]
    xyz = np.zeros((n, 3)); ok = np.zeros(n, dtype=np.uint8); ratio = np.zeros(n)
    fx, fy, cx, cy = [float(v) for v in K]
    if which == "ref":
        assert T_wc is None
        ref_lib().ref_triangulate(n, _p(uvL, dbl_p), _p(uvR, dbl_p), C.c_double(fx), C.c_double(fy),
                                  C.c_double(cx), C.c_double(cy), C.c_double(baseline), _p(xyz, dbl_p),
                                  _p(ok, u8_p), _p(ratio, dbl_p))
    else:
        T = None if T_wc is None else np.ascontiguousarray(T_wc, dtype=np.float64)
        oracle_lib().orc_triangulate(n, _p(uvL, dbl_p), _p(uvR, dbl_p), C.c_double(fx), C.c_double(fy),
                                     C.c_double(cx), C.c_double(cy), C.c_double(baseline), _p(T, dbl_p),
                                     _p(xyz, dbl_p), _p(ok, u8_p), _p(ratio, dbl_p))
    return dict(xyz=xyz, ok=ok, ratio=ratio)


# ------------------------------------------------------------------ ORB / stereo
def _img(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    assert a.ndim == 2
    return a


def orb_params(nfeatures=2000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7):
    return OrbParams(nfeatures, scale_factor, nlevels, ini_th, min_th)


def fast_roi(img, threshold, cap=1 << 16):
    img = _img(img)
    xs = np.zeros(cap, np.int32); ys = np.zeros(cap, np.int32); sc = np.zeros(cap, np.int32)
    n = oracle_lib().orc_fast_roi(_p(img, u8_p), img.strides[0], img.shape[0], img.shape[1], threshold, cap,
                                  _p(xs, i32_p), _p(ys, i32_p), _p(sc, i32_p))
    assert n <= cap
    return xs[:n].copy(), ys[:n].copy(), sc[:n].copy()


def is_fast_corner(img, x, y, threshold):
    img = _img(img)
    return bool(oracle_lib().orc_is_fast_corner(_p(img, u8_p), img.strides[0], int(x), int(y), int(threshold)))


def orb_grid_fast(img, mask=None, ini_th=20, min_th=7, cap=1 << 18):
    img = _img(img)
    mask = None if mask is None else _img(mask)
    out = np.zeros(cap, dtype=KP_DTYPE)
    n = oracle_lib().orc_orb_grid_fast(_p(img, u8_p), img.strides[0], img.shape[0], img.shape[1],
                                       _p(mask, u8_p), 0 if mask is None else mask.strides[0], ini_th, min_th, cap,
                                       out.ctypes.data_as(C.c_void_p))
    assert n <= cap
    return out[:n].copy()


def octree(cand, minX, maxX, minY, maxY, N):
    cand = np.ascontiguousarray(cand, dtype=KP_DTYPE)
    out = np.zeros(max(len(cand), 1), dtype=KP_DTYPE)
    n = oracle_lib().orc_octree(cand.ctypes.data_as(C.c_void_p), len(cand), minX, maxX, minY, maxY, N, len(out),
                                out.ctypes.data_as(C.c_void_p))
    return out[:n].copy()


def orb_detect(img, mask=None, prm=None, cap=1 << 16):
    img = _img(img)
    mask = None if mask is None else _img(mask)
    prm = prm or orb_params()
    out = np.zeros(cap, dtype=KP_DTYPE)
    n = oracle_lib().orc_orb_detect(_p(img, u8_p), img.strides[0], img.shape[0], img.shape[1], _p(mask, u8_p),
                                    0 if mask is None else mask.strides[0], C.byref(prm), cap,
                                    out.ctypes.data_as(C.c_void_p))
    assert n <= cap
    return out[:n].copy()


def orb_extract(img, mask=None, prm=None, cap=1 << 16):
    img = _img(img)
    mask = None if mask is None else _img(mask)
    prm = prm or orb_params()
    kps = np.zeros(cap, dtype=KP_DTYPE); desc = np.zeros((cap, 32), dtype=np.uint8)
    n = oracle_lib().orc_orb_extract(_p(img, u8_p), img.strides[0], img.shape[0], img.shape[1], _p(mask, u8_p),
                                     0 if mask is None else mask.strides[0], C.byref(prm), cap,
                                     kps.ctypes.data_as(C.c_void_p), _p(desc, u8_p))
    assert n <= cap
    return kps[:n].copy(), desc[:n].copy()


def orb_describe_at(img, kps_in, prm=None):
    img = _img(img)
    prm = prm or orb_params()
    kps_in = np.ascontiguousarray(kps_in, dtype=KP_DTYPE)
    kps = np.zeros(len(kps_in), dtype=KP_DTYPE); desc = np.zeros((len(kps_in), 32), dtype=np.uint8)
    n = oracle_lib().orc_orb_describe_at(_p(img, u8_p), img.strides[0], img.shape[0], img.shape[1], C.byref(prm),
                                         kps_in.ctypes.data_as(C.c_void_p), len(kps_in),
                                         kps.ctypes.data_as(C.c_void_p), _p(desc, u8_p))
    return kps[:n].copy(), desc[:n].copy()


def level_sizes(rows, cols, scale_factor=1.2, nlevels=8):
    r = np.zeros(nlevels, np.int32); c = np.zeros(nlevels, np.int32)
    oracle_lib().orc_level_sizes(rows, cols, C.c_float(scale_factor), nlevels, _p(r, i32_p), _p(c, i32_p))
    return r, c


def features_per_level(nfeatures, scale_factor=1.2, nlevels=8):
    o = np.zeros(nlevels, np.int32)
    oracle_lib().orc_features_per_level(nfeatures, C.c_float(scale_factor), nlevels, _p(o, i32_p))
    return o


def umax():
    o = np.zeros(16, np.int32)
    oracle_lib().orc_umax(_p(o, i32_p))
    return o


def resize_linear(src, drows, dcols):
    src = _img(src)
    drows, dcols = int(drows), int(dcols)
    dst = np.zeros((drows, dcols), np.uint8)
    oracle_lib().orc_resize_linear(_p(src, u8_p), src.strides[0], src.shape[0], src.shape[1], _p(dst, u8_p),
                                   dst.strides[0], drows, dcols)
    return dst


def gauss7(src):
    src = _img(src)
    dst = np.zeros_like(src)
    oracle_lib().orc_gauss7(_p(src, u8_p), src.strides[0], src.shape[0], src.shape[1], _p(dst, u8_p), dst.strides[0])
    return dst


def ic_angle(img, x, y):
    img = _img(img)
    return float(oracle_lib().orc_ic_angle(_p(img, u8_p), img.strides[0], C.c_float(x), C.c_float(y)))


def fast_atan2(y, x):
    return float(oracle_lib().orc_fast_atan2(C.c_float(y), C.c_float(x)))


def sincos_deg(a):
    c = C.c_float(0); s = C.c_float(0)
    oracle_lib().orc_sincos_deg(C.c_float(a), C.byref(c), C.byref(s))
    return c.value, s.value


def brief(blurred, x, y, angle):
    blurred = _img(blurred)
    d = np.zeros(32, np.uint8)
    oracle_lib().orc_brief(_p(blurred, u8_p), blurred.strides[0], C.c_float(x), C.c_float(y), C.c_float(angle),
                           _p(d, u8_p))
    return d


def brief_libm_census(blurred, xya, mode=0):
    """-> (descriptor bits that differ, descriptors with a differing bit, (cos, sin) pairs that differ) between the deterministic
    sincos_deg and libm (mode 0: cosf / sinf, 1: cos / sin of the widened angle rounded to float) over the keypoints xya [n, 3]"""
    blurred = _img(blurred)
    xya = np.ascontiguousarray(xya, dtype=np.float32)
    out = np.zeros(3, np.int64)
    oracle_lib().orc_brief_libm_census(_p(blurred, u8_p), blurred.strides[0], len(xya), _p(xya, C.POINTER(C.c_float)), int(mode),
                                       _p(out, C.POINTER(C.c_int64)))
    return int(out[0]), int(out[1]), int(out[2])


def brief_pattern():
    p = oracle_lib().orc_brief_pattern()
    return np.ctypeslib.as_array(p, shape=(256, 4)).copy()


def match_params(band_px=2.0, min_disp=0.0, max_disp=120.0, max_dist=80, max_octave_diff=1, scale_factor=1.2):
    return MatchParams(band_px, min_disp, max_disp, max_dist, max_octave_diff, scale_factor)


def stereo_match(kL, dL, kR, dR, prm=None):
    prm = prm or match_params()
    kL = np.ascontiguousarray(kL, dtype=KP_DTYPE); kR = np.ascontiguousarray(kR, dtype=KP_DTYPE)
    dL = np.ascontiguousarray(dL, dtype=np.uint8); dR = np.ascontiguousarray(dR, dtype=np.uint8)
    idx = np.zeros(len(kL), np.int32); dist = np.zeros(len(kL), np.int32)
    oracle_lib().orc_stereo_match(kL.ctypes.data_as(C.c_void_p), _p(dL, u8_p), len(kL),
                                  kR.ctypes.data_as(C.c_void_p), _p(dR, u8_p), len(kR), C.byref(prm),
                                  _p(idx, i32_p), _p(dist, i32_p))
    return idx, dist


def bf_match(dq, dt):
    dq = np.ascontiguousarray(dq, dtype=np.uint8); dt = np.ascontiguousarray(dt, dtype=np.uint8)
    idx = np.zeros(len(dq), np.int32); dist = np.zeros(len(dq), np.int32)
    oracle_lib().orc_bf_match(_p(dq, u8_p), len(dq), _p(dt, u8_p), len(dt), _p(idx, i32_p), _p(dist, i32_p))
    return idx, dist


# ---------------- pyramidal LK (N1) ----------------
class LkParams(C.Structure):
    _fields_ = [("win", C.c_int), ("max_level", C.c_int), ("max_iters", C.c_int), ("eps", C.c_double),
                ("min_eig_threshold", C.c_float), ("use_initial_flow", C.c_int)]


def lk_params(win=11, max_level=3, max_iters=30, eps=0.01, min_eig_threshold=1e-4, use_initial_flow=1):
    return LkParams(win, max_level, max_iters, eps, min_eig_threshold, use_initial_flow)


def lk_pyr_down(img):
    img = _img(img)
    out = np.zeros(((img.shape[0] + 1) // 2, (img.shape[1] + 1) // 2), np.uint8)
    oracle_lib().orc_lk_pyr_down(_p(img, u8_p), img.strides[0], img.shape[0], img.shape[1], _p(out, u8_p), out.strides[0])
    return out


def lk_scharr(img):
    img = _img(img)
    out = np.zeros((img.shape[0], img.shape[1], 2), np.int16)
    oracle_lib().orc_lk_scharr(_p(img, u8_p), img.strides[0], img.shape[0], img.shape[1], out.ctypes.data_as(C.c_void_p))
    return out


def lk_track(prev, nxt, prev_pts, next_pts_init=None, prm=None):
    """cv::calcOpticalFlowPyrLK as frontend.cpp:156-166 / 374-384 call it; returns (next_pts, status, err, top_level)."""
    prev = _img(prev); nxt = _img(nxt)
    assert prev.shape == nxt.shape
    prm = prm or lk_params()
    pp = np.ascontiguousarray(prev_pts, dtype=np.float32).reshape(-1, 2)
    npts = (pp.copy() if next_pts_init is None else np.ascontiguousarray(next_pts_init, dtype=np.float32).reshape(-1, 2).copy())
    n = len(pp)
    status = np.zeros(n, np.uint8); err = np.zeros(n, np.float32)
    f32_p = C.POINTER(C.c_float)
    lib = oracle_lib()
    lib.orc_lk_track.restype = C.c_int
    top = lib.orc_lk_track(_p(prev, u8_p), prev.strides[0], _p(nxt, u8_p), nxt.strides[0], prev.shape[0], prev.shape[1], n,
                           _p(pp, f32_p), _p(npts, f32_p), _p(status, u8_p), _p(err, f32_p), C.byref(prm))
    return npts, status, err, top


# ---------------- pose-graph optimisation (N3) ----------------
def se3_log(pose7):
    out = np.zeros(6)
    oracle_lib().orc_se3_log(_p(np.ascontiguousarray(pose7, dtype=np.float64), dbl_p), _p(out, dbl_p))
    return out


def pg_edge_eval(meas7, T0, T1):
    e = np.zeros(6); Ji = np.zeros(36); Jj = np.zeros(36)
    oracle_lib().orc_pg_edge_eval(_p(np.ascontiguousarray(meas7, dtype=np.float64), dbl_p), _p(np.ascontiguousarray(T0, dtype=np.float64), dbl_p),
                                  _p(np.ascontiguousarray(T1, dtype=np.float64), dbl_p), _p(e, dbl_p), _p(Ji, dbl_p), _p(Jj, dbl_p))
    return e, Ji.reshape(6, 6), Jj.reshape(6, 6)


def pose_graph_opt(pr, which="oracle", iters=20):
    """LoopClosing::PoseGraphOptimization on a flat problem; which = "oracle" (restatement) or "ref" (real g2o)."""
    poses = np.ascontiguousarray(pr["poses"], dtype=np.float64).copy()
    fixed = np.ascontiguousarray(pr["fixed"], dtype=np.uint8)
    ei = np.ascontiguousarray(pr["ei"], dtype=np.int32); ej = np.ascontiguousarray(pr["ej"], dtype=np.int32)
    meas = np.ascontiguousarray(pr["meas"], dtype=np.float64)
    E = len(ei)
    err = np.zeros((E, 6)); cap = iters + 2
    n = C.c_int(0); chi = np.zeros(cap); lam = np.zeros(cap); tr = np.zeros(cap, np.int32)
    lib = oracle_lib() if which == "oracle" else ref_lib()
    fn = lib.orc_pose_graph_opt if which == "oracle" else lib.ref_pose_graph
    fn.restype = C.c_int
    done = fn(len(poses), _p(poses, dbl_p), _p(fixed, u8_p), E, _p(ei, i32_p), _p(ej, i32_p), _p(meas, dbl_p), iters,
              _p(err, dbl_p), cap, C.byref(n), _p(chi, dbl_p), _p(lam, dbl_p), _p(tr, i32_p))
    k = n.value
    return dict(poses=poses, edge_err=err, n_iters=done, chi2=chi[:k].copy(), lambdas=lam[:k].copy(), trials=tr[:k].copy())


# ---------------- bag of words (N2) ----------------
def voc_transform_features(voc, feat):
    """per-descriptor (word id, weight) by the restated tree descent of TemplatedVocabulary::transform"""
    parent = np.ascontiguousarray(voc["parent"], dtype=np.int32); leaf = np.ascontiguousarray(voc["is_leaf"], dtype=np.uint8)
    desc = np.ascontiguousarray(voc["desc"], dtype=np.uint8); weight = np.ascontiguousarray(voc["weight"], dtype=np.float64)
    feat = np.ascontiguousarray(feat, dtype=np.uint8).reshape(-1, 32)
    word = np.zeros(len(feat), np.int32); w = np.zeros(len(feat), np.float64)
    oracle_lib().orc_voc_transform_features(len(parent), _p(parent, i32_p), _p(leaf, u8_p), _p(desc, u8_p), _p(weight, dbl_p), _p(feat, u8_p),
                                            len(feat), _p(word, i32_p), _p(w, dbl_p))
    return word, w


def bow_vector(word, weight, weighting=0, which="oracle"):
    """which = "ref": the reference's own DBoW2::BowVector (BowVector.cpp compiled into oracle/_ref)"""
    word = np.ascontiguousarray(word, dtype=np.int32); weight = np.ascontiguousarray(weight, dtype=np.float64)
    cap = max(len(word), 1)
    ids = np.zeros(cap, np.int32); vals = np.zeros(cap, np.float64)
    if which == "ref":
        m = ref_lib().ref_bow_vector(len(word), _p(word, i32_p), _p(weight, dbl_p), int(weighting), cap, _p(ids, i32_p), _p(vals, dbl_p))
        return ids[:m].copy(), vals[:m].copy()
    m = oracle_lib().orc_bow_vector(len(word), _p(word, i32_p), _p(weight, dbl_p), int(weighting), cap, _p(ids, i32_p), _p(vals, dbl_p))
    return ids[:m].copy(), vals[:m].copy()


def voc_transform(voc, feat, weighting=0):
    word, w = voc_transform_features(voc, feat)
    return bow_vector(word, w, weighting)


def bow_score_l1(a, b):
    fn = oracle_lib().orc_bow_score_l1
    fn.restype = C.c_double
    ia, va = np.ascontiguousarray(a[0], dtype=np.int32), np.ascontiguousarray(a[1], dtype=np.float64)
    ib, vb = np.ascontiguousarray(b[0], dtype=np.int32), np.ascontiguousarray(b[1], dtype=np.float64)
    return float(fn(len(ia), _p(ia, i32_p), _p(va, dbl_p), len(ib), _p(ib, i32_p), _p(vb, dbl_p)))
