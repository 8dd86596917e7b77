"""The C ABI under misuse: null pointers, negative / zero / inconsistent sizes, out-of-range indices, NaN inputs.
Every call must come back with an error status (or a documented empty result) -- never crash, never hang, and the
context must stay usable afterwards (the reference signals such conditions by assert / LOG(FATAL); ssx.h promises
status codes)."""
import ctypes as C

import numpy as np
import pytest

from ssvio_amd import _lib, ba, lk, orb
from ssvio_amd._lib import SsxError
from tools.synth import make_ba_problem, make_pose_graph_problem, make_pose_only_problem, make_stereo_pair

pytestmark = pytest.mark.gpu

NULL = None


def _usable(ctx, po):
    L = make_stereo_pair(seed=3, h=120, w=200, n_blobs=200)[0]
    k, d = orb.ORBextractor(ctx, 100, 1.2, 3).DetectAndCompute(L)
    ko, do = po.orb_extract(L, prm=po.orb_params(nfeatures=100, nlevels=3))
    assert k.tobytes() == ko.tobytes() and np.array_equal(d, do)


def test_orb_entry_points_reject_bad_arguments(ctx, po):
    lib, h = ctx.lib, ctx.handle
    img = make_stereo_pair(seed=1, h=100, w=160, n_blobs=150)[0]
    prm = orb.OrbParams(100, 1.2, 3, 20, 7)
    kps = np.zeros(600, dtype=orb.KP_DTYPE); desc = np.zeros((600, 32), np.uint8); n = C.c_int32(0)
    p_img, p_k, p_d = img.ctypes.data_as(_lib.u8_p), kps.ctypes.data_as(C.c_void_p), desc.ctypes.data_as(_lib.u8_p)
    bad = [
        lib.ssx_orb_extract(NULL, p_img, 160, 100, 160, NULL, 0, C.byref(prm), 600, p_k, p_d, C.byref(n)),          # no context
        lib.ssx_orb_extract(h, p_img, 100, 100, 160, NULL, 0, C.byref(prm), 600, p_k, p_d, C.byref(n)),             # stride < cols
        lib.ssx_orb_extract(h, p_img, 160, 100, 160, NULL, 0, NULL, 600, p_k, p_d, C.byref(n)),                     # no parameters
        lib.ssx_orb_extract(h, p_img, 160, 100, 160, NULL, 0, C.byref(prm), 600, p_k, p_d, NULL),                   # no count
        lib.ssx_orb_extract(h, p_img, 160, 100, 160, NULL, 0, C.byref(prm), 3, p_k, p_d, C.byref(n)),               # capacity too small
        lib.ssx_orb_extract(h, p_img, 160, 100, 160, NULL, 0, C.byref(orb.OrbParams(100, 1.2, 0, 20, 7)), 600, p_k, p_d, C.byref(n)),   # 0 levels
        lib.ssx_orb_extract(h, p_img, 160, 100, 160, NULL, 0, C.byref(orb.OrbParams(100, 0.9, 3, 20, 7)), 600, p_k, p_d, C.byref(n)),   # scale < 1
        lib.ssx_orb_extract(h, p_img, 160, 100, 160, NULL, 0, C.byref(orb.OrbParams(-5, 1.2, 3, 20, 7)), 600, p_k, p_d, C.byref(n)),    # negative budget
        lib.ssx_orb_extract(h, p_img, 160, 100, 160, NULL, 0, C.byref(orb.OrbParams(100, 1.2, 40, 20, 7)), 600, p_k, p_d, C.byref(n)),  # too many levels
        lib.ssx_orb_detect(h, p_img, 160, 100000, 160, NULL, 0, C.byref(prm), 600, p_k, C.byref(n)),                # absurd height
    ]
    assert all(st != _lib.SSX_OK for st in bad), [int(x) for x in bad]
    assert len(ctx.lib.ssx_last_error(h)) > 0
    # empty image (no rows or no data): OK with n = 0, like the reference's silent return on image.empty()
    assert lib.ssx_orb_detect(h, p_img, 160, 0, 0, NULL, 0, C.byref(prm), 600, p_k, C.byref(n)) == _lib.SSX_OK and n.value == 0
    assert lib.ssx_orb_extract(h, NULL, 160, 100, 160, NULL, 0, C.byref(prm), 600, p_k, p_d, C.byref(n)) == _lib.SSX_OK and n.value == 0
    _usable(ctx, po)


def test_matching_and_triangulation_reject_bad_arguments(ctx, po):
    lib, h = ctx.lib, ctx.handle
    k = np.zeros(8, dtype=orb.KP_DTYPE); d = np.zeros((8, 32), np.uint8); idx = np.zeros(8, np.int32); dist = np.zeros(8, np.int32)
    mp = orb.match_params(); rig = orb.stereo_rig()
    pk, pd = k.ctypes.data_as(C.c_void_p), d.ctypes.data_as(_lib.u8_p)
    pi, pdist = idx.ctypes.data_as(_lib.i32_p), dist.ctypes.data_as(_lib.i32_p)
    assert lib.ssx_stereo_match(h, NULL, pd, 8, pk, pd, 8, C.byref(mp), pi, pdist) != _lib.SSX_OK
    assert lib.ssx_stereo_match(h, pk, pd, -1, pk, pd, 8, C.byref(mp), pi, pdist) != _lib.SSX_OK
    assert lib.ssx_stereo_match(h, pk, pd, 8, pk, pd, 8, NULL, pi, pdist) != _lib.SSX_OK
    assert lib.ssx_bf_match(h, NULL, 8, pd, 8, pi, pdist) != _lib.SSX_OK
    assert lib.ssx_bf_match(h, pd, 8, pd, -3, pi, pdist) != _lib.SSX_OK
    uv = np.zeros((4, 2)); xyz = np.zeros((4, 3)); ok = np.zeros(4, np.uint8)
    pu, px, po_ = uv.ctypes.data_as(_lib.dbl_p), xyz.ctypes.data_as(_lib.dbl_p), ok.ctypes.data_as(_lib.u8_p)
    assert lib.ssx_triangulate(h, 4, NULL, pu, C.byref(rig), NULL, px, po_) != _lib.SSX_OK
    assert lib.ssx_triangulate(h, -4, pu, pu, C.byref(rig), NULL, px, po_) != _lib.SSX_OK
    assert lib.ssx_triangulate(h, 4, pu, pu, NULL, NULL, px, po_) != _lib.SSX_OK
    # NaN / inf pixels must not be reported as valid points
    uvn = np.array([[np.nan, 10.0], [np.inf, 5.0], [100.0, 50.0], [200.0, np.nan]]); uvr = np.array([[1.0, 10.0], [2.0, 5.0], [np.nan, 50.0], [190.0, 60.0]])
    xyz2, ok2 = orb.triangulate(ctx, uvn, uvr)
    assert not np.asarray(ok2).any()
    _usable(ctx, po)


def test_lk_rejects_bad_arguments(ctx, po):
    lib, h = ctx.lib, ctx.handle
    img = make_stereo_pair(seed=2, h=100, w=160, n_blobs=150)[0]
    p_img = img.ctypes.data_as(_lib.u8_p)
    pts = np.array([[50.0, 40.0], [np.nan, 10.0], [1e9, -1e9], [np.inf, 20.0]], np.float32); out = pts.copy()
    st = np.zeros(4, np.uint8)
    pp, pn, ps = pts.ctypes.data_as(_lib.f32_p), out.ctypes.data_as(_lib.f32_p), st.ctypes.data_as(_lib.u8_p)
    lib.ssx_lk_track.restype = C.c_int
    assert lib.ssx_lk_track(h, NULL, 160, p_img, 160, 100, 160, 4, pp, pn, ps, NULL, NULL, NULL) != _lib.SSX_OK
    assert lib.ssx_lk_track(h, p_img, 160, p_img, 160, 100, 160, -1, pp, pn, ps, NULL, NULL, NULL) != _lib.SSX_OK
    assert lib.ssx_lk_track(h, p_img, 100, p_img, 160, 100, 160, 4, pp, pn, ps, NULL, NULL, NULL) != _lib.SSX_OK      # stride < cols
    assert lib.ssx_lk_track(h, p_img, 160, p_img, 160, 1, 160, 4, pp, pn, ps, NULL, NULL, NULL) != _lib.SSX_OK        # 1-row image
    assert lib.ssx_lk_track(h, p_img, 160, p_img, 160, 100, 160, 4, pp, NULL, ps, NULL, NULL, NULL) != _lib.SSX_OK
    # non-finite / absurd points: status 0, no crash, finite points still tracked as the oracle tracks them
    g = lk.calcOpticalFlowPyrLK(ctx, img, img, pts, pts)
    o = po.lk_track(img, img, pts, pts)
    assert np.array_equal(g[1], o[1]) and g[1][0] == 1 and not g[1][1:].any()
    _usable(ctx, po)


def test_ba_and_pose_graph_reject_bad_problems(ctx, po):
    pr = make_ba_problem(P=4, L=60, obs_per_lm=3, seed=1)
    keep = []
    for field, value in (("P", 0), ("P", -3), ("L", -1), ("E", -5)):           # the counts of the raw C struct
        s = ba._problem_struct(pr, keep); setattr(s, field, value)
        res = _lib.BaResult()
        assert ctx.lib.ssx_ba_solve(ctx.handle, C.byref(s), None, C.byref(res)) != _lib.SSX_OK, (field, value)
    s = ba._problem_struct(pr, keep); s.poses = None
    assert ctx.lib.ssx_ba_solve(ctx.handle, C.byref(s), None, C.byref(_lib.BaResult())) != _lib.SSX_OK
    assert ctx.lib.ssx_ba_solve(ctx.handle, None, None, C.byref(_lib.BaResult())) != _lib.SSX_OK
    assert ctx.lib.ssx_ba_solve(ctx.handle, C.byref(ba._problem_struct(pr, keep)), None, None) != _lib.SSX_OK
    q = dict(pr); q["edge_pose"] = pr["edge_pose"].copy(); q["edge_pose"][7] = 99            # pose index out of range
    with pytest.raises(SsxError):
        ba.ba_solve(ctx, q)
    q = dict(pr); q["edge_point"] = pr["edge_point"].copy(); q["edge_point"][3] = -1         # negative landmark index
    with pytest.raises(SsxError):
        ba.ba_solve(ctx, q)
    # NaN measurements / poses: no crash, no hang, the call returns (the LM rejects every trial or reports failure)
    q = dict(pr); q["edge_uv"] = pr["edge_uv"].copy(); q["edge_uv"][5] = np.nan
    try:
        r = ba.ba_solve(ctx, q)
        assert r["n_iters"] <= 50
    except SsxError:
        pass
    q = dict(pr); q["poses"] = pr["poses"].copy(); q["poses"][1, 5] = np.inf
    try:
        ba.ba_solve(ctx, q)
    except SsxError:
        pass
    # a healthy solve still matches the oracle afterwards
    g = ba.ba_solve(ctx, pr); o = po.ba_solve(pr, "oracle", jac_mode=0)
    assert np.array_equal(g["trials"], o["trials"]) and np.abs(np.sqrt(g["edge_chi2"]) - np.sqrt(o["edge_chi2"])).max() < 1e-4
    pp = make_pose_only_problem(M=50, seed=2)
    bad_uv = pp["uv"].copy(); bad_uv[3] = np.nan
    try:
        ba.pose_only_opt(ctx, pp["pose"], pp["K"], pp["xyz"], bad_uv)
    except SsxError:
        pass
    pg = make_pose_graph_problem(P=12, n_loops=1, seed=3, n_active=3)
    q = dict(pg); q["ei"] = np.asarray(pg["ei"]).copy(); q["ei"][0] = 500
    with pytest.raises(Exception):
        ba.pose_graph_opt(ctx, q)
    g = ba.pose_graph_opt(ctx, pg, iters=5); o = po.pose_graph_opt(pg, iters=5)
    assert np.abs(g["poses"] - o["poses"]).max() < 2e-4
    _usable(ctx, po)


def test_vocabulary_entry_points_reject_bad_arguments(ctx, po):
    from ssvio_amd import voc as svoc
    from tools.synth import make_vocabulary
    lib, h = ctx.lib, ctx.handle
    vv = make_vocabulary(k=4, L=2, seed=1)
    out = C.c_void_p()
    par, leaf = vv["parent"], vv["is_leaf"]
    pp, pl, pd, pw = par.ctypes.data_as(_lib.i32_p), leaf.ctypes.data_as(_lib.u8_p), vv["desc"].ctypes.data_as(_lib.u8_p), vv["weight"].ctypes.data_as(_lib.dbl_p)
    n = len(par)
    assert lib.ssx_voc_create(None, 4, 2, 0, 0, n, pp, pl, pd, pw, C.byref(out)) != _lib.SSX_OK
    assert lib.ssx_voc_create(h, 4, 2, 0, 0, n, None, pl, pd, pw, C.byref(out)) != _lib.SSX_OK
    assert lib.ssx_voc_create(h, 4, 2, 0, 0, 0, pp, pl, pd, pw, C.byref(out)) != _lib.SSX_OK
    assert lib.ssx_voc_create(h, 4, 2, 0, 7, n, pp, pl, pd, pw, C.byref(out)) != _lib.SSX_OK            # unknown weighting
    assert lib.ssx_voc_create(h, 4, 2, 0, 0, n, pp, pl, pd, pw, None) != _lib.SSX_OK
    wrong_leaf = leaf.copy(); wrong_leaf[1] = 1                                                            # an inner node flagged as a leaf
    assert lib.ssx_voc_create(h, 4, 2, 0, 0, n, pp, wrong_leaf.ctypes.data_as(_lib.u8_p), pd, pw, C.byref(out)) != _lib.SSX_OK
    assert lib.ssx_voc_load_text(h, None, C.byref(out)) != _lib.SSX_OK
    V = svoc.Vocabulary.from_arrays(ctx, 4, 2, par, leaf, vv["desc"], vv["weight"])
    d = np.zeros((5, 32), np.uint8); ids = np.zeros(5, np.int32); vals = np.zeros(5); m = C.c_int32(0)
    pdd, pi, pv = d.ctypes.data_as(_lib.u8_p), ids.ctypes.data_as(_lib.i32_p), vals.ctypes.data_as(_lib.dbl_p)
    assert lib.ssx_voc_transform(None, pdd, 5, None, None, 5, pi, pv, C.byref(m)) != _lib.SSX_OK
    assert lib.ssx_voc_transform(V.handle, None, 5, None, None, 5, pi, pv, C.byref(m)) != _lib.SSX_OK
    assert lib.ssx_voc_transform(V.handle, pdd, -2, None, None, 5, pi, pv, C.byref(m)) != _lib.SSX_OK
    assert lib.ssx_voc_transform(V.handle, pdd, 5, None, None, 5, None, pv, C.byref(m)) != _lib.SSX_OK
    feats = np.random.default_rng(0).integers(0, 256, (400, 32), dtype=np.uint8)
    assert lib.ssx_voc_transform(V.handle, feats.ctypes.data_as(_lib.u8_p), 400, None, None, 1, pi, pv, C.byref(m)) == _lib.SSX_ERR_CAPACITY and m.value > 1
    assert lib.ssx_voc_transform(V.handle, feats.ctypes.data_as(_lib.u8_p), 400, None, None, 0, None, None, C.byref(m)) == _lib.SSX_OK and m.value > 1   # size query
    lib.ssx_bow_score_l1.restype = C.c_double
    assert lib.ssx_bow_score_l1(-1, None, None, 3, pi, pv) == 0.0 and lib.ssx_bow_score_l1(3, None, None, 3, pi, pv) == 0.0
    lib.ssx_voc_destroy(None)                                                                              # no-op
    gi, gv = V.transform(feats); oi, ov = po.voc_transform(vv, feats)
    assert np.array_equal(gi, oi) and gv.tobytes() == ov.tobytes()
    V.close()
    _usable(ctx, po)


def test_batched_entry_points_reject_bad_arguments(ctx, po):
    """ssx_lk_track_batch / ssx_pose_only_opt_batch / ssx_orb_detect_boxes_batch / ssx_triangulate_batch: a bad job anywhere in the
    table fails the CALL with a status before anything is launched; an empty table is OK; the slots of an earlier good call stay
    usable afterwards."""
    lib, h = ctx.lib, ctx.handle
    L, R = make_stereo_pair(seed=5, h=120, w=200, n_blobs=200)[:2]
    pts = np.array([[60.0, 50.0], [100.0, 70.0], [150.0, 90.0]], np.float32)
    good = lk.track_batch(ctx, [dict(slot=0, prev=L, next=R, prev_pts=pts), dict(slot=1, prev=R, next=L, prev_pts=pts)])
    lib.ssx_lk_track_batch.restype = C.c_int
    lib.ssx_lk_track_batch.argtypes = [C.c_void_p, C.c_int32, C.POINTER(lk.LkJob), C.c_int32, C.c_int32, C.POINTER(lk.LkParams), C.c_int32]
    prm = lk.LkParams(11, 3, 30, 0.01, 1e-4, 0)
    out = pts.copy(); st = np.zeros(3, np.uint8)

    def job(slot, prev, nxt, n=3, stride=200, status=st):
        q = lk.LkJob()
        q.slot = slot; q.prev = None if prev is None else prev.ctypes.data_as(_lib.u8_p); q.prev_stride = 0 if prev is None else stride
        q.next = None if nxt is None else nxt.ctypes.data_as(_lib.u8_p); q.next_stride = stride; q.n = n
        q.prev_pts = pts.ctypes.data_as(_lib.f32_p); q.next_pts = out.ctypes.data_as(_lib.f32_p)
        q.status = None if status is None else status.ctypes.data_as(_lib.u8_p); q.err = None
        return q

    def call(jobs, n=None, rows=120, cols=200):
        arr = (lk.LkJob * max(len(jobs), 1))(*jobs)
        return lib.ssx_lk_track_batch(h, len(jobs) if n is None else n, arr, rows, cols, C.byref(prm), 0)

    bad = [
        call([job(0, L, R)], n=-1),                              # negative job count
        lib.ssx_lk_track_batch(h, 2, None, 120, 200, C.byref(prm), 0),   # no table
        lib.ssx_lk_track_batch(None, 0, None, 120, 200, C.byref(prm), 0),  # no context
        call([job(0, L, R), job(0, R, L)]),                      # one slot twice
        call([job(0, L, R), job(-1, R, L)]),                     # slot below 0
        call([job(0, L, R), job(4096, R, L)]),                   # slot above the limit
        call([job(0, L, R), job(7, None, L)]),                   # chained job on a slot that never tracked
        call([job(0, L, None)]),                                 # no image
        call([job(0, L, R, stride=100)]),                        # stride < cols
        call([job(0, L, R, n=-3)]),                              # negative point count
        call([job(0, L, R, status=None)]),                       # points but no status array
        call([job(0, L, R)], rows=1),                            # 1-row image
    ]
    assert all(s_ != _lib.SSX_OK for s_ in bad), [int(x) for x in bad]
    assert call([], n=0) == _lib.SSX_OK
    # the slots of the good call are untouched by the rejected ones: the chained continuation equals a fresh call
    nxt = lk.track_batch(ctx, [dict(slot=0, prev=None, next=L, prev_pts=pts), dict(slot=1, prev=None, next=R, prev_pts=pts)])
    ref = lk.track_batch(ctx, [dict(slot=2, prev=R, next=L, prev_pts=pts), dict(slot=3, prev=L, next=R, prev_pts=pts)])
    for a, b in zip(nxt, ref):
        assert a[0].tobytes() == b[0].tobytes() and np.array_equal(a[1], b[1])
    assert len(good) == 2

    # pose-only table
    pr = make_pose_only_problem(seed=4, M=120)
    lib.ssx_pose_only_opt_batch.restype = C.c_int32
    lib.ssx_pose_only_opt_batch.argtypes = [C.c_void_p, C.c_int32, C.POINTER(ba.PoseOnlyJob)]
    pose = np.ascontiguousarray(pr["pose"], dtype=np.float64).copy(); K = np.ascontiguousarray(pr["K"], dtype=np.float64)
    xyz = np.ascontiguousarray(pr["xyz"], dtype=np.float64); uv = np.ascontiguousarray(pr["uv"], dtype=np.float64)

    def pjob(M=120, pose_=pose, K_=K, xyz_=xyz, rounds=4, iters=10):
        q = ba.PoseOnlyJob()
        q.pose_io = None if pose_ is None else pose_.ctypes.data_as(_lib.dbl_p); q.K4 = None if K_ is None else K_.ctypes.data_as(_lib.dbl_p)
        q.M = M; q.xyz = None if xyz_ is None else xyz_.ctypes.data_as(_lib.dbl_p); q.uv = uv.ctypes.data_as(_lib.dbl_p)
        q.rounds = rounds; q.iters = iters; q.chi2_th = 5.991; q.huber_delta = 1.0; q.inlier_out = None; q.n_inliers = None
        return q

    def pcall(jobs, n=None):
        arr = (ba.PoseOnlyJob * max(len(jobs), 1))(*jobs)
        return lib.ssx_pose_only_opt_batch(h, len(jobs) if n is None else n, arr)

    p0 = pose.copy()
    bad = [pcall([pjob()], n=-1), lib.ssx_pose_only_opt_batch(h, 1, None), lib.ssx_pose_only_opt_batch(None, 0, None),
           pcall([pjob(), pjob(pose_=None)]), pcall([pjob(), pjob(K_=None)]), pcall([pjob(), pjob(M=-1)]), pcall([pjob(), pjob(xyz_=None)]),
           pcall([pjob(), pjob(rounds=-1)]), pcall([pjob(), pjob(iters=-1)])]
    assert all(s_ != _lib.SSX_OK for s_ in bad), [int(x) for x in bad]
    assert np.array_equal(pose, p0), "a rejected table must not have run its valid jobs"
    assert pcall([], n=0) == _lib.SSX_OK
    # an empty problem and a NaN problem beside a good one: the good one's result is the single call's
    one = ba.pose_only_opt(ctx, pr["pose"], pr["K"], pr["xyz"], pr["uv"])
    nanp = dict(pr, xyz=np.full_like(xyz, np.nan))
    emp = dict(pr, xyz=np.zeros((0, 3)), uv=np.zeros((0, 2)))
    res = ba.pose_only_opt_batch(ctx, [nanp, pr, emp])
    assert res[1]["pose"].tobytes() == one["pose"].tobytes() and np.array_equal(res[1]["inliers"], one["inliers"])
    assert res[2]["n_inliers"] == 0 and np.array_equal(res[2]["pose"], np.asarray(pr["pose"], dtype=np.float64))
    single_nan = ba.pose_only_opt(ctx, nanp["pose"], nanp["K"], nanp["xyz"], nanp["uv"])
    assert res[0]["pose"].tobytes() == single_nan["pose"].tobytes() and res[0]["n_inliers"] == single_nan["n_inliers"]

    # detection and triangulation tables
    class DJob(C.Structure):
        _fields_ = [("img", _lib.u8_p), ("stride", C.c_int32), ("boxes", C.POINTER(C.c_int32)), ("n_boxes", C.c_int32), ("cap", C.c_int32),
                    ("kps_out", C.c_void_p), ("n_out", C.POINTER(C.c_int32))]
    lib.ssx_orb_detect_boxes_batch.restype = C.c_int32
    lib.ssx_orb_detect_boxes_batch.argtypes = [C.c_void_p, C.c_int32, C.POINTER(DJob), C.c_int32, C.c_int32, C.POINTER(orb.OrbParams), C.c_int32]
    oprm = orb.OrbParams(100, 1.2, 3, 20, 7)
    kps = np.zeros(600, dtype=orb.KP_DTYPE); cnt = (C.c_int32 * 1)(0)

    def djob(img=L, stride=200, cap=600, n_boxes=0, n_out=cnt):
        q = DJob()
        q.img = None if img is None else img.ctypes.data_as(_lib.u8_p); q.stride = stride; q.boxes = None; q.n_boxes = n_boxes; q.cap = cap
        q.kps_out = kps.ctypes.data_as(C.c_void_p); q.n_out = None if n_out is None else C.cast(n_out, C.POINTER(C.c_int32))
        return q

    def dcall(jobs, n=None, rows=120, cols=200, prm_=oprm):
        arr = (DJob * max(len(jobs), 1))(*jobs)
        return lib.ssx_orb_detect_boxes_batch(h, len(jobs) if n is None else n, arr, rows, cols, None if prm_ is None else C.byref(prm_), 0)

    bad = [dcall([djob()], n=-1), dcall([djob()], prm_=None), dcall([djob()], rows=0), dcall([djob(), djob(img=None)]), dcall([djob(), djob(stride=100)]),
           dcall([djob(), djob(cap=-1)]), dcall([djob(), djob(n_boxes=2)]), dcall([djob(), djob(n_out=None)]), dcall([djob(stride=200), djob(stride=256)]),
           lib.ssx_orb_detect_boxes_batch(h, 2, None, 120, 200, C.byref(oprm), 0)]
    assert all(s_ != _lib.SSX_OK for s_ in bad), [int(x) for x in bad]
    assert dcall([], n=0) == _lib.SSX_OK
    assert dcall([djob(cap=1)]) == _lib.SSX_ERR_CAPACITY                     # the image has more than one corner

    class TJob(C.Structure):
        _fields_ = [("n", C.c_int32), ("uvL", _lib.dbl_p), ("uvR", _lib.dbl_p), ("rig", C.POINTER(orb.StereoRig)), ("T_wc", _lib.dbl_p),
                    ("xyz_out", _lib.dbl_p), ("ok_out", _lib.u8_p)]
    lib.ssx_triangulate_batch.restype = C.c_int32
    lib.ssx_triangulate_batch.argtypes = [C.c_void_p, C.c_int32, C.POINTER(TJob)]
    rig = orb.stereo_rig(); uvL = np.array([[100.0, 60.0], [np.nan, 5.0]]); uvR = np.array([[90.0, 60.0], [1.0, 5.0]])
    xo = np.zeros((2, 3)); ok = np.zeros(2, np.uint8)

    def tjob(n=2, rig_=rig, uvR_=uvR, ok_=ok):
        q = TJob()
        q.n = n; q.uvL = uvL.ctypes.data_as(_lib.dbl_p); q.uvR = None if uvR_ is None else uvR_.ctypes.data_as(_lib.dbl_p)
        q.rig = None if rig_ is None else C.pointer(rig_); q.T_wc = None; q.xyz_out = xo.ctypes.data_as(_lib.dbl_p)
        q.ok_out = None if ok_ is None else ok_.ctypes.data_as(_lib.u8_p)
        return q

    def tcall(jobs, n=None):
        arr = (TJob * max(len(jobs), 1))(*jobs)
        return lib.ssx_triangulate_batch(h, len(jobs) if n is None else n, arr)

    bad = [tcall([tjob()], n=-1), lib.ssx_triangulate_batch(h, 1, None), lib.ssx_triangulate_batch(None, 0, None), tcall([tjob(), tjob(n=-1)]),
           tcall([tjob(), tjob(rig_=None)]), tcall([tjob(), tjob(uvR_=None)]), tcall([tjob(), tjob(ok_=None)])]
    assert all(s_ != _lib.SSX_OK for s_ in bad), [int(x) for x in bad]
    assert tcall([], n=0) == _lib.SSX_OK and tcall([tjob(n=0, uvR_=None, ok_=None)]) == _lib.SSX_OK
    assert tcall([tjob()]) == _lib.SSX_OK and ok[0] == 1 and ok[1] == 0          # the NaN point is flagged, not propagated
    _usable(ctx, po)
