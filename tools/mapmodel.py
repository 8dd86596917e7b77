"""The reference's active-map bookkeeping around Backend::OptimizeActiveMap, restated in plain Python so that tests and the
bench can drive an ssx_ba_window exactly like ssvio's backend drives its map -- and check the window against a solve of the
re-marshalled map after every keyframe.

    Map::InsertKeyFrame / RemoveOldActiveKeyframe / RemoveOldActiveMapPoints / RemoveAllOutlierMapPoints
                                           /root/reference/src/ssvio/map.cpp:18-58, 89-160, 175-194
    MapPoint observation lists             /root/reference/src/ssvio/mappoint.cpp:22-79
    KeyFrame::CreateKF                     /root/reference/src/ssvio/keyframe.cpp:11-50
    Backend::OptimizeActiveMap             /root/reference/src/ssvio/backend.cpp:88-169 (graph), 205-244 (what it does to the map)

Nothing here computes: `ActiveMap.problem()` lists the graph the reference would hand to g2o (keyframes and map points ascending
by id, as g2o orders its vertices), `ActiveMap.apply()` writes a result back the way backend.cpp:205-244 does, and every change
of the window between two optimisations is recorded as an edit (`ActiveMap.take_edits()`) that `apply_edits` replays on a
`ssvio_amd.ba.BaWindow`.  `make_window_scenario` generates a synthetic drive (no GPU, no oracle, no reference).
"""
from __future__ import annotations

import numpy as np

from tools import synth


class _Feature:
    __slots__ = ("kf", "lm", "uv", "cam", "outlier")

    def __init__(self, kf, lm, uv, cam=0):
        self.kf, self.lm, self.uv, self.cam, self.outlier = kf, lm, (float(uv[0]), float(uv[1])), int(cam), False


class _MapPoint:
    __slots__ = ("id", "pos", "outlier", "obs", "active_obs")

    def __init__(self, id_, pos):
        self.id, self.pos, self.outlier, self.obs, self.active_obs = id_, np.array(pos, dtype=np.float64), False, [], []


class ActiveMap:
    """all_map_points_ / activate_map_points_ / all_active_key_frames_ of the reference's Map plus the parts of Backend that
    edit them.  Keyframe and map-point ids are the caller's (the reference numbers them with growing counters)."""

    def __init__(self, n_active, K=synth.KITTI_K, cam_ext=None):
        self.n_active = int(n_active)
        self.K = np.array(K, dtype=np.float64)
        self.cam_ext = synth.stereo_cam_ext() if cam_ext is None else np.array(cam_ext, dtype=np.float64)
        self.kfs = {}            # id -> dict(pose, feats)            (all_key_frames_)
        self.active_kfs = []     # ids, insertion order               (all_active_key_frames_)
        self.mps = {}            # id -> _MapPoint                    (all_map_points_)
        self.active_mps = {}     # id -> _MapPoint                    (activate_map_points_)
        self.outlier_list = []   # list_outlier_map_points_
        self.in_window = set()   # landmark ids an ssx_ba_window fed with take_edits() holds
        self._edits = []
        self.stats = dict(reentered=0, fixed_by_rule=0, condemned=0, outlier_edges=0)

    # ---- frontend side -------------------------------------------------------------------------------------------------
    def condemn(self, lm_id):
        """FrontEnd::EstimateCurrentPose, frontend.cpp:283-288: the map point is an outlier; it leaves the map with the next
        RemoveAllOutlierMapPoints (end of the next optimisation) and is skipped by the graph until then (backend.cpp:116)"""
        mp = self.mps.get(lm_id)
        if mp is None or mp.outlier:
            return
        mp.outlier = True
        self.outlier_list.append(lm_id)
        self.stats["condemned"] += 1
        if lm_id in self.in_window:
            self.in_window.discard(lm_id)
            self._edits.append(("remove_lm", [lm_id]))

    def insert_keyframe(self, kf_id, pose, obs, new_points, victim=None):
        """FrontEnd::InsertKeyFrame -> KeyFrame::CreateKF -> Backend::InsertKeyFrame -> Map::InsertKeyFrame.
        obs: [(lm_id, (u, v))] left-image features that carry a map point; new_points: {lm_id: xyz} triangulated with this
        keyframe (frontend.cpp:500-544; they are in `obs` too).  victim: the keyframe RemoveOldActiveKeyframe drops when the
        window overflows (None = the oldest; the reference picks by pose distance, map.cpp:96-133)."""
        for lm_id, xyz in new_points.items():
            self.mps[lm_id] = _MapPoint(lm_id, xyz)                                   # Map::InsertMapPoint
        feats = []
        for lm_id, uv in obs:
            mp = self.mps.get(lm_id)
            if mp is None:                                                            # weak_ptr expired: the point was deleted
                continue
            f = _Feature(kf_id, lm_id, np.float32(uv).astype(np.float64))             # cv::KeyPoint::pt is a Point2f
            feats.append(f)
            mp.obs.append(f)                                                          # keyframe.cpp:48 AddObservation
        self.kfs[kf_id] = dict(pose=np.array(pose, dtype=np.float64), feats=feats)
        self.active_kfs.append(kf_id)
        for f in feats:                                                               # map.cpp:41-49
            mp = self.mps[f.lm]
            mp.active_obs.append(f)
            self.active_mps[mp.id] = mp
        # the window's view of the push: observations of live, not condemned map points; landmarks it does not hold come in
        # with the flag of backend.cpp:125-130 evaluated on the map (a point that comes BACK has an observer that left long ago)
        w_obs = [f for f in feats if not self.mps[f.lm].outlier]
        new_ids = []
        for f in w_obs:
            if f.lm not in self.in_window and f.lm not in new_ids:
                new_ids.append(f.lm)
        active = set(self.active_kfs)
        new_fixed = [0 if self.mps[l].obs[0].kf in active else 1 for l in new_ids]
        self.stats["reentered"] += sum(1 for l in new_ids if l not in new_points)
        self.in_window.update(new_ids)
        self._edits.append(("push", dict(kf_id=kf_id, pose=self.kfs[kf_id]["pose"].copy(), new_ids=np.array(new_ids, dtype=np.int64),
                                         new_xyz=np.array([self.mps[l].pos for l in new_ids]).reshape(-1, 3),
                                         new_fixed=np.array(new_fixed, dtype=np.uint8),
                                         obs_lm=np.array([f.lm for f in w_obs], dtype=np.int64),
                                         obs_uv=np.array([f.uv for f in w_obs]).reshape(-1, 2),
                                         obs_cam=np.array([f.cam for f in w_obs], dtype=np.uint8))))
        if len(self.active_kfs) > self.n_active:                                      # map.cpp:52-56
            v = self.active_kfs[0] if victim is None else victim
            assert v in self.active_kfs and v != kf_id
            self.active_kfs.remove(v)
            for f in self.kfs[v]["feats"]:                                            # map.cpp:137-145 RemoveActiveObservation
                mp = self.mps.get(f.lm) if f.lm is not None else None
                if mp is not None and f in mp.active_obs:
                    mp.active_obs.remove(f)
            self._remove_old_active_map_points()
            self._edits.append(("pop", v))

    def _remove_old_active_map_points(self):                                          # map.cpp:148-166
        for lm_id in [i for i, mp in self.active_mps.items() if not mp.active_obs]:
            del self.active_mps[lm_id]
            self.in_window.discard(lm_id)                                             # (the window drops it in the same edit)

    # ---- backend side ---------------------------------------------------------------------------------------------------
    def problem(self):
        """The graph of backend.cpp:88-169 as an ssx_ba_problem dict: keyframes ascending by id (none fixed), map points ascending
        by id (active, not condemned, with at least one edge; fixed = the keyframe of observations.front() is not active), edges
        per map point in active-observation order.  Also returns kf_ids, lm_ids and the edges' features."""
        kf_ids = sorted(self.active_kfs)
        kf_index = {k: i for i, k in enumerate(kf_ids)}
        lm_ids, points, fixed, e_pose, e_point, e_uv, e_cam, e_feat = [], [], [], [], [], [], [], []
        for lm_id in sorted(self.active_mps):
            mp = self.active_mps[lm_id]
            if mp.outlier:
                continue
            edges = [f for f in mp.active_obs if f.kf in kf_index and not f.outlier]
            if not edges:
                continue
            fx = 0 if (mp.obs and mp.obs[0].kf in kf_index) else 1
            if fx:
                self.stats["fixed_by_rule"] += 1
            for f in edges:
                e_pose.append(kf_index[f.kf]); e_point.append(len(lm_ids)); e_uv.append(f.uv); e_cam.append(f.cam); e_feat.append(f)
            lm_ids.append(lm_id); points.append(mp.pos); fixed.append(fx)
        pr = dict(P=len(kf_ids), L=len(lm_ids), E=len(e_pose),
                  poses=np.array([self.kfs[k]["pose"] for k in kf_ids]).reshape(-1, 7), pose_fixed=np.zeros(len(kf_ids), dtype=np.uint8),
                  points=np.array(points).reshape(-1, 3), point_fixed=np.array(fixed, dtype=np.uint8),
                  edge_pose=np.array(e_pose, dtype=np.int32), edge_point=np.array(e_point, dtype=np.int32),
                  edge_uv=np.array(e_uv).reshape(-1, 2), edge_cam=np.array(e_cam, dtype=np.uint8), K=self.K.copy(), cam_ext=self.cam_ext.copy())
        return pr, kf_ids, lm_ids, e_feat

    def apply(self, kf_ids, lm_ids, e_feat, poses, points, edge_outlier):
        """backend.cpp:205-244: outlier edges lose their observation (a map point left without any is condemned), poses and
        positions are written back, condemned map points are deleted, map points nobody observes actively leave the window."""
        removed = []
        for f, out in zip(e_feat, edge_outlier):
            if not out:
                f.outlier = False
                continue
            f.outlier = True
            mp = self.mps[f.lm]
            mp.active_obs.remove(f)
            mp.obs.remove(f)
            removed.append((f.kf, f.lm, f.cam))
            if not mp.obs:
                mp.outlier = True
                self.outlier_list.append(mp.id)
            f.lm = None
        self.stats["outlier_edges"] += len(removed)
        for k, p in zip(kf_ids, poses):
            self.kfs[k]["pose"] = np.array(p, dtype=np.float64)
        for l, x in zip(lm_ids, points):
            self.mps[l].pos = np.array(x, dtype=np.float64)
        for lm_id in self.outlier_list:                                               # map.cpp:175-194
            self.mps.pop(lm_id, None)
            self.active_mps.pop(lm_id, None)
            self.in_window.discard(lm_id)
        self.outlier_list = []
        self._remove_old_active_map_points()
        if removed:
            self._edits.append(("remove_obs", removed))

    def take_edits(self):
        e, self._edits = self._edits, []
        return e


def apply_edits(win, edits):
    """replay ActiveMap.take_edits() on a ssvio_amd.ba.BaWindow (created with fix_rule=1)"""
    for kind, arg in edits:
        if kind == "push":
            win.push(arg["kf_id"], arg["pose"], new_ids=arg["new_ids"], new_xyz=arg["new_xyz"], new_fixed=arg["new_fixed"], obs_lm=arg["obs_lm"],
                     obs_uv=arg["obs_uv"], obs_cam=arg["obs_cam"])
        elif kind == "pop":
            win.pop(arg)
        elif kind == "remove_lm":
            win.remove_landmarks(arg)
        elif kind == "remove_obs":
            by_kf = {}
            for kf, lm, cam in arg:
                by_kf.setdefault(kf, []).append((lm, cam))
            for kf, lst in by_kf.items():
                win.remove_observations(kf, [l for l, _ in lst], [c for _, c in lst])
        else:
            raise ValueError(kind)


def make_window_scenario(n_kf=14, n_active=5, new_per_kf=140, track_len=7, seed=0, frac_gross=0.04, pix_sigma=0.4, step=0.8,
                         pose_t_noise=0.02, pose_r_noise=0.002, point_noise=0.05, K=synth.KITTI_K):
    """A synthetic forward drive as the sequence of keyframes a frontend would hand to the backend: keyframe i (ids 100, 101, ...)
    brings `new_per_kf` new map points (ids grow with time) and re-observes the map points earlier keyframes introduced while
    their track lasts (1 .. track_len keyframes); `frac_gross` of the observations are gross outliers.  Returns a list of
    dict(kf_id, pose, obs, new_points, victim, condemn): `victim` names the keyframe the map drops when the window overflows
    (mostly the oldest, now and then a younger one: the reference picks by pose distance), `condemn` map points the frontend
    declares outliers BEFORE this keyframe is inserted (they are then no longer observed)."""
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy = K
    frames = []
    tracks = []                                                  # (lm_id, xyz_gt, first kf index, length)
    next_lm = 0
    active = []
    condemned = set()
    for i in range(n_kf):
        kf_id = 100 + i
        gt = np.array([0, 0, 0, 1, 0.05 * np.sin(0.7 * i), 0, -step * i], dtype=np.float64)       # T_cw of a camera at z = step * i
        q = synth.small_rot_quat(rng.uniform(-pose_r_noise, pose_r_noise, 3))
        pose = gt.copy()
        pose[:4] = q / np.linalg.norm(q)
        pose[4:] = synth.quat_rot(q, gt[4:]) + rng.uniform(-pose_t_noise, pose_t_noise, 3)
        condemn = []
        live = [t for t in tracks if t[2] + t[3] > i and t[0] not in condemned]
        if i >= 3 and live:
            for t in live:
                if rng.random() < 0.01:
                    condemn.append(t[0]); condemned.add(t[0])
        obs, new_points = [], {}
        for _ in range(new_per_kf):
            local = np.array([rng.uniform(-12, 12), rng.uniform(-3, 3), rng.uniform(6, 40)])
            xyz = local + np.array([-gt[4], -gt[5], -gt[6]])
            tracks.append((next_lm, xyz, i, int(rng.integers(1, track_len + 1))))
            new_points[next_lm] = xyz + rng.normal(0, point_noise, 3)
            next_lm += 1
        for lm_id, xyz, i0, ln in tracks:
            if not (i0 <= i < i0 + ln) or lm_id in condemned:
                continue
            pc = xyz + gt[4:]
            if pc[2] < 1.0:
                continue
            uv = np.array([fx * pc[0] / pc[2] + cx, fy * pc[1] / pc[2] + cy]) + rng.normal(0, pix_sigma, 2)
            if rng.random() < frac_gross:
                uv += rng.normal(0, 30.0, 2)
            obs.append((lm_id, uv))
        active.append(kf_id)
        victim = None
        if len(active) > n_active:
            victim = active[0] if rng.random() < 0.6 else active[int(rng.integers(0, n_active - 1))]
            active.remove(victim)
        frames.append(dict(kf_id=kf_id, pose=pose, obs=obs, new_points=new_points, victim=victim, condemn=condemn))
    return frames
