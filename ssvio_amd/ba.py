"""Local bundle adjustment through libssx.so -- host-side marshalling for Backend::OptimizeActiveMap
(/root/reference/src/ssvio/backend.cpp:78-245).

`problem` is a dict of flat arrays in the layout of ssx_ba_problem (include/ssx.h); see
tools.synth.make_ba_problem.  Every function here calls the HIP library; nothing is computed in
Python.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import BaOptions, BaProblem, BaResult, Context, dbl_p, i32_p, ptr, u8_p

JAC_ANALYTIC = 0
JAC_NUMERIC_G2O = 1


def _problem_struct(pr, keep):
    poses = np.ascontiguousarray(pr["poses"], dtype=np.float64)
    points = np.ascontiguousarray(pr["points"], dtype=np.float64)
    pose_fixed = None if pr.get("pose_fixed") is None else np.ascontiguousarray(pr["pose_fixed"], dtype=np.uint8)
    point_fixed = None if pr.get("point_fixed") is None else np.ascontiguousarray(pr["point_fixed"], dtype=np.uint8)
    edge_pose = np.ascontiguousarray(pr["edge_pose"], dtype=np.int32)
    edge_point = np.ascontiguousarray(pr["edge_point"], dtype=np.int32)
    edge_uv = np.ascontiguousarray(pr["edge_uv"], dtype=np.float64)
    edge_cam = None if pr.get("edge_cam") is None else np.ascontiguousarray(pr["edge_cam"], dtype=np.uint8)
    keep.extend([poses, points, pose_fixed, point_fixed, edge_pose, edge_point, edge_uv, edge_cam])
    s = BaProblem()
    s.P = int(poses.shape[0]); s.poses = ptr(poses, dbl_p); s.pose_fixed = ptr(pose_fixed, u8_p)
    s.L = int(points.shape[0]); s.points = ptr(points, dbl_p); s.point_fixed = ptr(point_fixed, u8_p)
    s.E = int(edge_pose.shape[0]); s.edge_pose = ptr(edge_pose, i32_p); s.edge_point = ptr(edge_point, i32_p)
    s.edge_uv = ptr(edge_uv, dbl_p); s.edge_cam = ptr(edge_cam, u8_p)
    for i, v in enumerate(np.asarray(pr["K"], dtype=np.float64).ravel()[:4]):
        s.K[i] = float(v)
    for i, v in enumerate(np.asarray(pr["cam_ext"], dtype=np.float64).ravel()[:14]):
        s.cam_ext[i] = float(v)
    return s


def ba_solve(ctx: Context, pr, outer_rounds=5, iters=10, chi2_th=5.891, huber_delta=5.891,
             inlier_ratio=0.7, jac_mode=JAC_ANALYTIC, allreduce=None, rank=0, world_size=1,
             want_edges=True, comm=None, collect_stats=False, large_solver=0):
    """Backend::OptimizeActiveMap's optimisation (defaults = backend.cpp:109,163,175,178,195).
    comm = an ssx_comm handle (dist_ba.init_native_comm): landmark shard of a multi-GPU solve over RCCL."""
    keep = []
    s = _problem_struct(pr, keep)
    opt = BaOptions()
    ctx.lib.ssx_ba_default_options(C.byref(opt))
    opt.outer_rounds = outer_rounds; opt.iters = iters; opt.chi2_th = chi2_th
    opt.huber_delta = huber_delta; opt.inlier_ratio = inlier_ratio; opt.jac_mode = jac_mode
    opt.rank = rank; opt.world_size = world_size
    if allreduce is not None:
        cb = _lib.ALLREDUCE_FN(allreduce)
        keep.append(cb)
        opt.allreduce = cb
    if comm is not None:
        opt.comm = comm
    opt.collect_stats = 1 if collect_stats else 0
    opt.large_solver = int(large_solver)                  # 0 auto, 1 tiles, 2 band (windows beyond 16 free keyframes)
    res = BaResult()
    poses = np.zeros((s.P, 7)); points = np.zeros((s.L, 3))
    chi2 = np.zeros(s.E) if want_edges else None
    outl = np.zeros(s.E, dtype=np.uint8) if want_edges else None
    res.poses_out = ptr(poses, dbl_p); res.points_out = ptr(points, dbl_p)
    res.edge_chi2 = ptr(chi2, dbl_p); res.edge_outlier = ptr(outl, u8_p)
    ctx.check(ctx.lib.ssx_ba_solve(ctx.handle, C.byref(s), C.byref(opt), C.byref(res)))
    k = min(res.n_iters, _lib.SSX_BA_MAX_STATS)
    return dict(rounds=res.rounds, n_iters=res.n_iters, poses=poses, points=points, edge_chi2=chi2,
                edge_outlier=outl, chi2=np.array(res.iter_chi2[:k]), lam=np.array(res.iter_lambda[:k]),
                trials=np.array(res.iter_trials[:k]), n_inliers=res.n_inliers, n_outliers=res.n_outliers,
                ms_total=res.ms_total,
                phase_ms=(dict(linearize=res.ms_linearize, schur=res.ms_schur, linear_solution=res.ms_linear_solution,
                               update=res.ms_update, reduce=res.ms_reduce, comm=res.ms_comm) if collect_stats else None))


def ba_linearize(ctx: Context, pr, huber_delta=5.891, jac_mode=JAC_ANALYTIC):
    """One linearisation (blocks of the normal equations) -- kernel-level parity hook."""
    keep = []
    s = _problem_struct(pr, keep)
    P, L, E = s.P, s.L, s.E
    Hpp = np.zeros((P, 6, 6)); bp = np.zeros((P, 6)); Hll = np.zeros((L, 3, 3)); bl = np.zeros((L, 3))
    Hpl = np.zeros((E, 6, 3)); err = np.zeros((E, 2)); chi = C.c_double(0)
    ctx.check(ctx.lib.ssx_ba_linearize(ctx.handle, C.byref(s), C.c_double(huber_delta), jac_mode,
                                       ptr(Hpp, dbl_p), ptr(bp, dbl_p), ptr(Hll, dbl_p), ptr(bl, dbl_p),
                                       ptr(Hpl, dbl_p), ptr(err, dbl_p), C.byref(chi)))
    return dict(Hpp=Hpp, bp=bp, Hll=Hll, bl=bl, Hpl=Hpl, err=err, chi2=chi.value)


def pose_only_opt(ctx: Context, pose, K, xyz, uv, rounds=4, iters=10, chi2_th=5.991, huber_delta=1.0):
    """FrontEnd::EstimateCurrentPose's optimisation (frontend.cpp:184-270) in one kernel launch.
    -> dict(pose, inliers (uint8 per feature), n_inliers)."""
    pose = np.ascontiguousarray(pose, dtype=np.float64).copy()
    K = np.ascontiguousarray(K, dtype=np.float64)
    xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
    uv = np.ascontiguousarray(uv, dtype=np.float64).reshape(-1, 2)
    M = xyz.shape[0]
    inl = np.zeros(M, dtype=np.uint8)
    n = C.c_int32(0)
    ctx.check(ctx.lib.ssx_pose_only_opt(ctx.handle, ptr(pose, dbl_p), ptr(K, dbl_p), M, ptr(xyz, dbl_p), ptr(uv, dbl_p),
                                        rounds, iters, C.c_double(chi2_th), C.c_double(huber_delta), ptr(inl, u8_p),
                                        C.byref(n)))
    return dict(pose=pose, inliers=inl, n_inliers=n.value)


class PoseOnlyJob(C.Structure):
    _fields_ = [("pose_io", dbl_p), ("K4", dbl_p), ("M", C.c_int32), ("xyz", dbl_p), ("uv", dbl_p), ("rounds", C.c_int32), ("iters", C.c_int32),
                ("chi2_th", C.c_double), ("huber_delta", C.c_double), ("inlier_out", u8_p), ("n_inliers", C.POINTER(C.c_int32))]


def pose_only_opt_batch(ctx: Context, problems, rounds=4, iters=10, chi2_th=5.991, huber_delta=1.0, prepared=False):
    """ssx_pose_only_opt_batch: problems = [dict(pose, K, xyz, uv)] -> [dict(pose, inliers, n_inliers)], one launch for all.
    prepared=True: -> a callable that restores the initial poses and makes the library call alone (the job structs built once, as a C
    caller holds them)."""
    n = len(problems)
    arr = (PoseOnlyJob * n)()
    keep, outs = [], []
    for i, pr in enumerate(problems):
        pose = np.ascontiguousarray(pr["pose"], dtype=np.float64).copy()
        K = np.ascontiguousarray(pr["K"], dtype=np.float64)
        xyz = np.ascontiguousarray(pr["xyz"], dtype=np.float64).reshape(-1, 3)
        uv = np.ascontiguousarray(pr["uv"], dtype=np.float64).reshape(-1, 2)
        inl = np.zeros(len(xyz), dtype=np.uint8)
        cnt = (C.c_int32 * 1)(0)
        a = arr[i]
        a.pose_io = ptr(pose, dbl_p); a.K4 = ptr(K, dbl_p); a.M = len(xyz); a.xyz = ptr(xyz, dbl_p); a.uv = ptr(uv, dbl_p)
        a.rounds = rounds; a.iters = iters; a.chi2_th = chi2_th; a.huber_delta = huber_delta
        a.inlier_out = ptr(inl, u8_p); a.n_inliers = C.cast(cnt, C.POINTER(C.c_int32))
        keep.append((K, xyz, uv)); outs.append((pose, inl, cnt))
    ctx.lib.ssx_pose_only_opt_batch.restype = C.c_int32
    ctx.lib.ssx_pose_only_opt_batch.argtypes = [C.c_void_p, C.c_int32, C.POINTER(PoseOnlyJob)]
    if prepared:
        init = [np.ascontiguousarray(pr["pose"], dtype=np.float64).copy() for pr in problems]

        def run():
            for (p_, _, _), p0 in zip(outs, init):
                p_[:] = p0
            ctx.check(ctx.lib.ssx_pose_only_opt_batch(ctx.handle, n, arr))
            return [dict(pose=p_, inliers=i_, n_inliers=int(c_[0])) for p_, i_, c_ in outs]
        run.keep = (keep, arr, outs)
        return run
    ctx.check(ctx.lib.ssx_pose_only_opt_batch(ctx.handle, n, arr))
    return [dict(pose=p_, inliers=i_, n_inliers=int(c_[0])) for p_, i_, c_ in outs]


class PoseGraphProblem(C.Structure):
    _fields_ = [("n_poses", C.c_int32), ("n_edges", C.c_int32), ("poses", C.POINTER(C.c_double)),
                ("pose_fixed", C.POINTER(C.c_ubyte)), ("edge_i", C.POINTER(C.c_int32)), ("edge_j", C.POINTER(C.c_int32)),
                ("edge_meas", C.POINTER(C.c_double)), ("edge_err_out", C.POINTER(C.c_double)), ("stats_cap", C.c_int32),
                ("stats_chi2", C.POINTER(C.c_double)), ("stats_lambda", C.POINTER(C.c_double)),
                ("stats_trials", C.POINTER(C.c_int32))]


class PoseGraphResult(C.Structure):
    _fields_ = [("n_iters", C.c_int32), ("stats_n", C.c_int32), ("chi2_initial", C.c_double), ("chi2_final", C.c_double)]


def pose_graph_opt(ctx, pr, iters=20):
    """LoopClosing::PoseGraphOptimization (loopclosing.cpp:458-539) on a flat problem:
    pr = dict(poses [P,7], fixed [P], ei [E], ej [E], meas [E,7])."""
    poses = np.ascontiguousarray(pr["poses"], dtype=np.float64).copy()
    fixed = np.ascontiguousarray(pr["fixed"], dtype=np.uint8)
    ei = np.ascontiguousarray(pr["ei"], dtype=np.int32); ej = np.ascontiguousarray(pr["ej"], dtype=np.int32)
    meas = np.ascontiguousarray(pr["meas"], dtype=np.float64)
    E = len(ei)
    err = np.zeros((max(E, 1), 6)); cap = iters + 2
    chi = np.zeros(cap); lam = np.zeros(cap); tr = np.zeros(cap, np.int32)
    dp = C.POINTER(C.c_double)
    prob = PoseGraphProblem(len(poses), E, poses.ctypes.data_as(dp), fixed.ctypes.data_as(C.POINTER(C.c_ubyte)),
                            ei.ctypes.data_as(C.POINTER(C.c_int32)), ej.ctypes.data_as(C.POINTER(C.c_int32)),
                            meas.ctypes.data_as(dp), err.ctypes.data_as(dp), cap, chi.ctypes.data_as(dp),
                            lam.ctypes.data_as(dp), tr.ctypes.data_as(C.POINTER(C.c_int32)))
    res = PoseGraphResult()
    ctx.lib.ssx_pose_graph_opt.restype = C.c_int
    ctx.check(ctx.lib.ssx_pose_graph_opt(ctx.handle, C.byref(prob), int(iters), C.byref(res)))
    k = res.stats_n
    return dict(poses=poses, edge_err=err[:E], n_iters=res.n_iters, chi2=chi[:k].copy(), lambdas=lam[:k].copy(),
                trials=tr[:k].copy(), chi2_initial=res.chi2_initial, chi2_final=res.chi2_final)


class BaBatch:
    """Many local windows per call -- ssx_ba_solve_batch (include/ssx.h): one window per stereo pair of a batch / per
    stream of BASELINE configs[4].  The problems are marshalled into ctypes once; solve() uploads, optimises and downloads
    ALL of them in one library call (the kernels run once for all windows)."""

    def __init__(self, ctx: Context, problems, outer_rounds=5, iters=10, chi2_th=5.891, huber_delta=5.891, inlier_ratio=0.7,
                 jac_mode=JAC_ANALYTIC, resident=False, with_edge_errors=True):
        """resident=True: ssx_ba_batch_create -- the windows are uploaded once and stay in HBM; solve() re-optimises them
        from the uploaded state (solve(download=False) moves nothing but the LM control words across PCIe)."""
        self.ctx = ctx
        self.handle = None
        self.n = len(problems)
        self._keep = []
        self.structs = (BaProblem * self.n)()
        for i, pr in enumerate(problems):
            self.structs[i] = _problem_struct(pr, self._keep)
        self.opt = BaOptions()
        ctx.lib.ssx_ba_default_options(C.byref(self.opt))
        self.opt.outer_rounds = outer_rounds; self.opt.iters = iters; self.opt.chi2_th = chi2_th
        self.opt.huber_delta = huber_delta; self.opt.inlier_ratio = inlier_ratio; self.opt.jac_mode = jac_mode
        self.res = (BaResult * self.n)()
        self.poses = [np.zeros((s.P, 7)) for s in self.structs]
        self.points = [np.zeros((s.L, 3)) for s in self.structs]
        self.chi2 = [np.zeros(s.E) for s in self.structs]
        self.outl = [np.zeros(s.E, dtype=np.uint8) for s in self.structs]
        self.with_edge_errors = with_edge_errors
        if resident:
            h = C.c_void_p()
            ctx.check(ctx.lib.ssx_ba_batch_create(ctx.handle, self.n, self.structs, C.byref(self.opt), 1 if with_edge_errors else 0, C.byref(h)))
            self.handle = h

    @property
    def groups(self):
        """groups of windows the library runs side by side (each batched kernel is launched once per group)"""
        if self.handle is None:
            return 1
        self.ctx.lib.ssx_ba_batch_groups.restype = C.c_int32
        self.ctx.lib.ssx_ba_batch_groups.argtypes = [C.c_void_p]
        return int(self.ctx.lib.ssx_ba_batch_groups(self.handle))

    def set_groups(self, groups):
        """1 .. 4 groups of windows side by side for the following solves, 0 = the library's default"""
        if self.handle is not None:
            self.ctx.lib.ssx_ba_batch_set_groups.restype = None
            self.ctx.lib.ssx_ba_batch_set_groups.argtypes = [C.c_void_p, C.c_int32]
            self.ctx.lib.ssx_ba_batch_set_groups(self.handle, int(groups))

    def solve(self, want_edges=True, download=True, summaries=True, points=True):
        """summaries=False: the per-window dicts are not built (poses / points are in self.poses / self.points, the counters
        in self.res): what a C caller pays -- building 128 dicts with their LM histories costs Python ~2 ms.
        points=False (with want_edges=False): only the keyframe poses are downloaded (ssx_ba_result.points_out NULL)."""
        if self.handle is not None and not download:
            tot = C.c_int32(0)
            self.ctx.check(self.ctx.lib.ssx_ba_batch_solve(self.handle, None, C.byref(tot)))
            return dict(results=None, n_iters_total=tot.value)
        want_edges = want_edges and (self.handle is None or self.with_edge_errors)
        if getattr(self, "_res_edges", None) != (want_edges, points):   # (the output pointers of the result structs: set once)
            for i in range(self.n):
                r = self.res[i]
                r.poses_out = ptr(self.poses[i], dbl_p); r.points_out = ptr(self.points[i], dbl_p) if points else None
                r.edge_chi2 = ptr(self.chi2[i], dbl_p) if want_edges else None
                r.edge_outlier = ptr(self.outl[i], u8_p) if want_edges else None
            self._res_edges = (want_edges, points)
        if self.handle is not None:
            self.ctx.check(self.ctx.lib.ssx_ba_batch_solve(self.handle, self.res, None))
        else:
            self.ctx.check(self.ctx.lib.ssx_ba_solve_batch(self.ctx.handle, self.n, self.structs, C.byref(self.opt), self.res))
        if not summaries:
            return dict(results=None, n_iters_total=None)
        out = []
        for i in range(self.n):
            r = self.res[i]
            k = min(r.n_iters, _lib.SSX_BA_MAX_STATS)
            out.append(dict(rounds=r.rounds, n_iters=r.n_iters, poses=self.poses[i], points=self.points[i],
                            edge_chi2=self.chi2[i] if want_edges else None, edge_outlier=self.outl[i] if want_edges else None,
                            chi2=np.array(r.iter_chi2[:k]), lam=np.array(r.iter_lambda[:k]), trials=np.array(r.iter_trials[:k]),
                            n_inliers=r.n_inliers, n_outliers=r.n_outliers, ms_total=r.ms_total))
        return dict(results=out, n_iters_total=int(sum(o["n_iters"] for o in out)))

    def close(self):
        """destroy the resident batch (before its Context is closed: the library frees device memory of that device)"""
        if self.handle is not None:
            self.ctx.lib.ssx_ba_batch_destroy.restype = None
            self.ctx.lib.ssx_ba_batch_destroy.argtypes = [C.c_void_p]
            self.ctx.lib.ssx_ba_batch_destroy(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:                                   # noqa: BLE001 -- interpreter shutdown: the library may be gone
            pass


i64_p = C.POINTER(C.c_int64)


class BaWindow:
    """A sliding local-BA window resident in HBM -- ssx_ba_window (include/ssx.h): push / pop keyframes, solve in place.
    What Backend::OptimizeActiveMap sees at consecutive keyframes (backend.cpp:88-169, map.cpp:27-56, 89-160)."""

    def __init__(self, ctx: Context, K, cam_ext, outer_rounds=5, iters=10, chi2_th=5.891, huber_delta=5.891, inlier_ratio=0.7,
                 jac_mode=JAC_ANALYTIC, fix_rule=0):
        """fix_rule=1: the window keeps backend.cpp:125-130 itself (a landmark is fixed while the keyframe of its first remaining
        observation is outside the window)"""
        self.ctx = ctx
        self.handle = None
        self.opt = BaOptions()
        ctx.lib.ssx_ba_default_options(C.byref(self.opt))
        self.opt.outer_rounds = outer_rounds; self.opt.iters = iters; self.opt.chi2_th = chi2_th
        self.opt.huber_delta = huber_delta; self.opt.inlier_ratio = inlier_ratio; self.opt.jac_mode = jac_mode
        self.K = np.ascontiguousarray(np.asarray(K, dtype=np.float64).ravel()[:4])
        self.cam_ext = np.ascontiguousarray(np.asarray(cam_ext, dtype=np.float64).ravel()[:14])
        lib = ctx.lib
        lib.ssx_ba_window_create.argtypes = [C.c_void_p, C.POINTER(BaOptions), dbl_p, dbl_p, C.POINTER(C.c_void_p)]
        lib.ssx_ba_window_destroy.argtypes = [C.c_void_p]; lib.ssx_ba_window_destroy.restype = None
        lib.ssx_ba_window_push_keyframe.argtypes = [C.c_void_p, C.c_int64, dbl_p, C.c_int32, C.c_int32, i64_p, dbl_p, u8_p, C.c_int32, i64_p, dbl_p, u8_p]
        lib.ssx_ba_window_push_keyframe_slots.argtypes = [C.c_void_p, C.c_int64, dbl_p, C.c_int32, C.c_int32, i64_p, dbl_p, u8_p, i32_p, C.c_int32, i32_p,
                                                          dbl_p, u8_p]
        lib.ssx_ba_window_pop_keyframe.argtypes = [C.c_void_p, C.c_int64]
        lib.ssx_ba_window_remove_flagged.argtypes = [C.c_void_p, C.c_int32, u8_p, i32_p]
        lib.ssx_ba_window_remove_observations.argtypes = [C.c_void_p, C.c_int64, C.c_int32, i64_p, u8_p, i32_p]
        lib.ssx_ba_window_remove_landmarks.argtypes = [C.c_void_p, C.c_int32, i64_p, i32_p]
        lib.ssx_ba_window_set_fix_rule.argtypes = [C.c_void_p, C.c_int32]
        lib.ssx_ba_window_set_pose.argtypes = [C.c_void_p, C.c_int64, dbl_p, C.c_int32]
        lib.ssx_ba_window_set_landmark.argtypes = [C.c_void_p, C.c_int64, dbl_p, C.c_int32]
        lib.ssx_ba_window_size.argtypes = [C.c_void_p, i32_p, i32_p, i32_p]
        lib.ssx_ba_window_export.argtypes = [C.c_void_p, i64_p, dbl_p, u8_p, i64_p, dbl_p, u8_p, i32_p, i32_p, dbl_p, u8_p]
        lib.ssx_ba_window_solve.argtypes = [C.c_void_p, C.POINTER(BaResult)]
        lib.ssx_ba_window_solve_batch.argtypes = [C.c_int32, C.POINTER(C.c_void_p), C.POINTER(BaResult)]
        h = C.c_void_p()
        ctx.check(lib.ssx_ba_window_create(ctx.handle, C.byref(self.opt), ptr(self.K, dbl_p), ptr(self.cam_ext, dbl_p), C.byref(h)))
        self.handle = h
        if fix_rule:
            ctx.check(lib.ssx_ba_window_set_fix_rule(h, int(fix_rule)))

    def remove_flagged(self, flags):
        """ssx_ba_window_remove_flagged: flags in export order (the edge_outlier of the solve that just ran); -> observations removed"""
        flags = np.ascontiguousarray(flags, dtype=np.uint8).ravel()
        n = C.c_int32(0)
        self.ctx.check(self.ctx.lib.ssx_ba_window_remove_flagged(self.handle, len(flags), ptr(flags, u8_p), C.byref(n)))
        return n.value

    def remove_observations(self, kf_id, lm_ids, cams=None):
        lm_ids = np.ascontiguousarray(lm_ids, dtype=np.int64).ravel()
        cams = None if cams is None else np.ascontiguousarray(cams, dtype=np.uint8)
        n = C.c_int32(0)
        self.ctx.check(self.ctx.lib.ssx_ba_window_remove_observations(self.handle, int(kf_id), len(lm_ids), ptr(lm_ids, i64_p), ptr(cams, u8_p), C.byref(n)))
        return n.value

    def remove_landmarks(self, lm_ids):
        lm_ids = np.ascontiguousarray(lm_ids, dtype=np.int64).ravel()
        n = C.c_int32(0)
        self.ctx.check(self.ctx.lib.ssx_ba_window_remove_landmarks(self.handle, len(lm_ids), ptr(lm_ids, i64_p), C.byref(n)))
        return n.value

    def push(self, kf_id, pose, new_ids=(), new_xyz=(), new_fixed=None, obs_lm=(), obs_uv=(), obs_cam=None, pose_fixed=False):
        pose = np.ascontiguousarray(pose, dtype=np.float64).ravel()
        new_ids = np.ascontiguousarray(new_ids, dtype=np.int64).ravel()
        new_xyz = np.ascontiguousarray(new_xyz, dtype=np.float64).reshape(-1, 3)
        new_fixed = None if new_fixed is None else np.ascontiguousarray(new_fixed, dtype=np.uint8)
        obs_lm = np.ascontiguousarray(obs_lm, dtype=np.int64).ravel()
        obs_uv = np.ascontiguousarray(obs_uv, dtype=np.float64).reshape(-1, 2)
        obs_cam = None if obs_cam is None else np.ascontiguousarray(obs_cam, dtype=np.uint8)
        self.ctx.check(self.ctx.lib.ssx_ba_window_push_keyframe(
            self.handle, int(kf_id), ptr(pose, dbl_p), 1 if pose_fixed else 0, len(new_ids), ptr(new_ids, i64_p), ptr(new_xyz, dbl_p),
            ptr(new_fixed, u8_p), len(obs_lm), ptr(obs_lm, i64_p), ptr(obs_uv, dbl_p), ptr(obs_cam, u8_p)))

    def push_slots(self, kf_id, pose, new_ids=(), new_xyz=(), new_fixed=None, obs_slot=(), obs_uv=(), obs_cam=None, pose_fixed=False):
        """ssx_ba_window_push_keyframe_slots: observations name their landmark by window slot (>= 0) or as -1 - i = the i-th new
        landmark of this call; returns the slots given to the new landmarks"""
        pose = np.ascontiguousarray(pose, dtype=np.float64).ravel()
        new_ids = np.ascontiguousarray(new_ids, dtype=np.int64).ravel()
        new_xyz = np.ascontiguousarray(new_xyz, dtype=np.float64).reshape(-1, 3)
        new_fixed = None if new_fixed is None else np.ascontiguousarray(new_fixed, dtype=np.uint8)
        obs_slot = np.ascontiguousarray(obs_slot, dtype=np.int32).ravel()
        obs_uv = np.ascontiguousarray(obs_uv, dtype=np.float64).reshape(-1, 2)
        obs_cam = None if obs_cam is None else np.ascontiguousarray(obs_cam, dtype=np.uint8)
        slots = np.zeros(len(new_ids), dtype=np.int32)
        self.ctx.check(self.ctx.lib.ssx_ba_window_push_keyframe_slots(
            self.handle, int(kf_id), ptr(pose, dbl_p), 1 if pose_fixed else 0, len(new_ids), ptr(new_ids, i64_p), ptr(new_xyz, dbl_p),
            ptr(new_fixed, u8_p), ptr(slots, i32_p), len(obs_slot), ptr(obs_slot, i32_p), ptr(obs_uv, dbl_p), ptr(obs_cam, u8_p)))
        return slots

    def pop(self, kf_id):
        self.ctx.check(self.ctx.lib.ssx_ba_window_pop_keyframe(self.handle, int(kf_id)))

    def set_pose(self, kf_id, pose, fixed=-1):
        pose = np.ascontiguousarray(pose, dtype=np.float64).ravel()
        self.ctx.check(self.ctx.lib.ssx_ba_window_set_pose(self.handle, int(kf_id), ptr(pose, dbl_p), int(fixed)))

    def set_landmark(self, lm_id, xyz=None, fixed=-1):
        xyz = None if xyz is None else np.ascontiguousarray(xyz, dtype=np.float64).ravel()
        self.ctx.check(self.ctx.lib.ssx_ba_window_set_landmark(self.handle, int(lm_id), ptr(xyz, dbl_p), int(fixed)))

    def size(self):
        a, b, c = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        self.ctx.check(self.ctx.lib.ssx_ba_window_size(self.handle, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def export(self):
        """the window as an ordinary problem dict (current estimate) + the ids of its keyframes / landmarks"""
        P, L, E = self.size()
        o = dict(P=P, L=L, E=E, kf_ids=np.zeros(P, np.int64), poses=np.zeros((P, 7)), pose_fixed=np.zeros(P, np.uint8),
                 lm_ids=np.zeros(L, np.int64), points=np.zeros((L, 3)), point_fixed=np.zeros(L, np.uint8),
                 edge_pose=np.zeros(E, np.int32), edge_point=np.zeros(E, np.int32), edge_uv=np.zeros((E, 2)), edge_cam=np.zeros(E, np.uint8),
                 K=self.K.copy(), cam_ext=self.cam_ext.copy())
        self.ctx.check(self.ctx.lib.ssx_ba_window_export(
            self.handle, ptr(o["kf_ids"], i64_p), ptr(o["poses"], dbl_p), ptr(o["pose_fixed"], u8_p), ptr(o["lm_ids"], i64_p),
            ptr(o["points"], dbl_p), ptr(o["point_fixed"], u8_p), ptr(o["edge_pose"], i32_p), ptr(o["edge_point"], i32_p),
            ptr(o["edge_uv"], dbl_p), ptr(o["edge_cam"], u8_p)))
        return o

    def _result_buffers(self, want_edges):
        P, L, E = self.size()
        res = BaResult()
        bufs = dict(poses=np.zeros((P, 7)), points=np.zeros((L, 3)), chi2=np.zeros(E) if want_edges else None,
                    outl=np.zeros(E, dtype=np.uint8) if want_edges else None)
        res.poses_out = ptr(bufs["poses"], dbl_p); res.points_out = ptr(bufs["points"], dbl_p)
        res.edge_chi2 = ptr(bufs["chi2"], dbl_p); res.edge_outlier = ptr(bufs["outl"], u8_p)
        return res, bufs

    @staticmethod
    def _result_dict(res, bufs):
        k = min(res.n_iters, _lib.SSX_BA_MAX_STATS)
        return dict(rounds=res.rounds, n_iters=res.n_iters, poses=bufs["poses"], points=bufs["points"], edge_chi2=bufs["chi2"],
                    edge_outlier=bufs["outl"], chi2=np.array(res.iter_chi2[:k]), lam=np.array(res.iter_lambda[:k]),
                    trials=np.array(res.iter_trials[:k]), n_inliers=res.n_inliers, n_outliers=res.n_outliers, ms_total=res.ms_total)

    def solve(self, want_edges=True):
        res, bufs = self._result_buffers(want_edges)
        self.ctx.check(self.ctx.lib.ssx_ba_window_solve(self.handle, C.byref(res)))
        return self._result_dict(res, bufs)

    @staticmethod
    def solve_batch(windows, want_edges=True):
        """ssx_ba_window_solve_batch: the windows (of one Context) in one call"""
        n = len(windows)
        arr = (BaResult * n)()
        bufs = []
        for i, w in enumerate(windows):
            r, b = w._result_buffers(want_edges)
            arr[i] = r
            bufs.append(b)
        hs = (C.c_void_p * n)(*[w.handle for w in windows])
        ctx = windows[0].ctx
        ctx.check(ctx.lib.ssx_ba_window_solve_batch(n, hs, arr))
        return [BaWindow._result_dict(arr[i], bufs[i]) for i in range(n)]

    @staticmethod
    def update_batch(windows, updates):
        """ssx_ba_window_update_batch: one keyframe replaced in each window (of one Context) in one call, on the library's host
        threads.  updates[i] = dict(remove_flags=per-observation flags or None, remove_lm=landmark ids or None, pop=kf id or None,
        push=kf id or None, pose, new_ids, new_xyz, new_fixed, obs_lm | obs_slot, obs_uv, obs_cam, pose_fixed), applied in that
        order; returns the slots of the new landmarks per window (slot form) or None."""
        n = len(windows)
        arr = (_lib.BaWindowUpdate * n)()
        keep, slots = [], []
        i64p = C.POINTER(C.c_int64)
        for i, u in enumerate(updates):
            a = arr[i]
            a.pop = 0 if u.get("pop") is None else 1
            a.pop_kf_id = 0 if u.get("pop") is None else int(u["pop"])
            a.push = 0 if u.get("push") is None else 1
            slots.append(None)
            if u.get("remove_flags") is not None:
                rf = np.ascontiguousarray(u["remove_flags"], dtype=np.uint8).ravel()
                a.n_remove_flags = len(rf); a.remove_flags = ptr(rf, u8_p)
                keep.append(rf)
            if u.get("remove_lm") is not None:
                rl = np.ascontiguousarray(u["remove_lm"], dtype=np.int64).ravel()
                a.n_remove_lm = len(rl); a.remove_lm_ids = ptr(rl, i64p)
                keep.append(rl)
            if not a.push:
                continue
            a.kf_id = int(u["push"])
            pose = np.ascontiguousarray(u["pose"], dtype=np.float64).ravel()
            new_ids = np.ascontiguousarray(u.get("new_ids", ()), dtype=np.int64).ravel()
            new_xyz = np.ascontiguousarray(u.get("new_xyz", ()), dtype=np.float64).reshape(-1, 3)
            new_fixed = None if u.get("new_fixed") is None else np.ascontiguousarray(u["new_fixed"], dtype=np.uint8)
            obs_uv = np.ascontiguousarray(u.get("obs_uv", ()), dtype=np.float64).reshape(-1, 2)
            obs_cam = None if u.get("obs_cam") is None else np.ascontiguousarray(u["obs_cam"], dtype=np.uint8)
            a.pose7 = ptr(pose, dbl_p); a.pose_fixed = 1 if u.get("pose_fixed") else 0
            a.n_new = len(new_ids); a.new_ids = ptr(new_ids, i64p); a.new_xyz = ptr(new_xyz, dbl_p); a.new_fixed = ptr(new_fixed, u8_p)
            a.n_obs = len(obs_uv); a.obs_uv = ptr(obs_uv, dbl_p); a.obs_cam = ptr(obs_cam, u8_p)
            if u.get("obs_slot") is not None:
                obs_slot = np.ascontiguousarray(u["obs_slot"], dtype=np.int32).ravel()
                so = np.zeros(len(new_ids), dtype=np.int32)
                a.obs_slot = ptr(obs_slot, i32_p); a.new_slots_out = ptr(so, i32_p)
                keep.append(obs_slot); slots[i] = so
            else:
                obs_lm = np.ascontiguousarray(u.get("obs_lm", ()), dtype=np.int64).ravel()
                a.obs_lm = ptr(obs_lm, i64p)
                keep.append(obs_lm)
            keep.extend([pose, new_ids, new_xyz, new_fixed, obs_uv, obs_cam])
        hs = (C.c_void_p * n)(*[w.handle for w in windows])
        ctx = windows[0].ctx
        ctx.lib.ssx_ba_window_update_batch.restype = C.c_int32
        ctx.lib.ssx_ba_window_update_batch.argtypes = [C.c_int32, C.POINTER(C.c_void_p), C.POINTER(_lib.BaWindowUpdate), C.POINTER(C.c_int32)]
        ctx.check(ctx.lib.ssx_ba_window_update_batch(n, hs, arr, None))
        return slots

    def close(self):
        if self.handle is not None:
            self.ctx.lib.ssx_ba_window_destroy(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:                                   # noqa: BLE001
            pass
