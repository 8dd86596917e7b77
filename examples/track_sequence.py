"""A compact stereo visual-odometry loop in the structure of ssvio's FrontEnd (StereoInit / Track / keyframe
insertion, /root/reference/src/ssvio/frontend.cpp:24-128) + Backend::OptimizeActiveMap, written ONLY against the
compute entry points this repository provides:

    Detect                      ORBextractor::Detect              (frontend.cpp:302-344 DetectFeatures)
    calcOpticalFlowPyrLK        TrackLastFrame / FindFeaturesInRight (frontend.cpp:130-182, 346-428)
    pose_only                   EstimateCurrentPose               (frontend.cpp:184-300)
    triangulate                 BuidInitMap / TriangulateNewPoints (frontend.cpp:448-544)
    ba                          Backend::OptimizeActiveMap        (backend.cpp:78-245)

The same host logic runs on two providers -- the GPU library and the CPU oracle -- so tests/test_track_gpu.py can
compare a whole tracked sequence (system-level parity), and it reports frames/s.

    python examples/track_sequence.py [--frames 20] [--provider gpu|oracle]
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from tools import synth  # noqa: E402
from tools.synth import KITTI_BASELINE, KITTI_K, pose_inv, pose_mul, quat_rot  # noqa: E402


class GpuProvider:
    def __init__(self, ctx):
        from ssvio_amd import ba, lk, orb
        self.ctx, self.ba_mod, self.lk_mod, self.orb = ctx, ba, lk, orb

    def detect(self, img, mask, n):
        k = self.orb.ORBextractor(self.ctx, nfeatures=n).Detect(img, mask)
        return np.stack([k["x"], k["y"]], 1).astype(np.float32)

    def lk(self, a, b, pts, init):
        p, st, _, _ = self.lk_mod.calcOpticalFlowPyrLK(self.ctx, a, b, pts, init)
        return p, st

    def pose_only(self, pose, xyz, uv):
        r = self.ba_mod.pose_only_opt(self.ctx, pose, np.array(KITTI_K), xyz, uv)
        return r["pose"], r["inliers"]

    def triangulate(self, uvL, uvR, T_wc):
        return self.orb.triangulate(self.ctx, uvL, uvR, T_wc=T_wc)

    def ba(self, pr):
        r = self.ba_mod.ba_solve(self.ctx, pr, want_edges=False)
        return r["poses"], r["points"]


class OracleProvider:
    def __init__(self, po):
        self.po = po

    def detect(self, img, mask, n):
        k = self.po.orb_detect(img, mask=mask, prm=self.po.orb_params(nfeatures=n))
        return np.stack([k["x"], k["y"]], 1).astype(np.float32)

    def lk(self, a, b, pts, init):
        p, st, _, _ = self.po.lk_track(a, b, pts, init)
        return p, st

    def pose_only(self, pose, xyz, uv):
        r = self.po.pose_only(dict(pose=pose, M=len(xyz), xyz=np.ascontiguousarray(xyz), uv=np.ascontiguousarray(uv), K=np.array(KITTI_K)))
        return r["pose"], r["inliers"]

    def triangulate(self, uvL, uvR, T_wc):
        t = self.po.triangulate(uvL, uvR, KITTI_K, KITTI_BASELINE, T_wc=T_wc)
        return t["xyz"], t["ok"]

    def ba(self, pr):
        r = self.po.ba_solve(pr, "oracle", jac_mode=0)
        return r["poses"], r["points"]


def project(T_cw, xyz):
    fx, fy, cx, cy = KITTI_K
    pc = np.array([quat_rot(T_cw[:4], p) + T_cw[4:] for p in xyz]).reshape(-1, 3)
    return np.stack([fx * pc[:, 0] / pc[:, 2] + cx, fy * pc[:, 1] / pc[:, 2] + cy], 1)


def run(provider, frames, n_init=300, n_new=100, kf_below=230, window=7):
    """Returns dict(poses [n,7] T_cw per frame, keyframes [frame ids], n_points, tracked [per frame])."""
    H, W = frames[0][0].shape
    ident = np.array([0, 0, 0, 1, 0, 0, 0], dtype=np.float64)
    points = {}                                   # map point id -> xyz (world)
    kfs = []                                      # keyframes: dict(frame, pose, obs=[(mp, uvL, uvR or None)])
    next_id = 0
    poses, tracked = [], []

    def stereo_and_triangulate(L, R, pts, mp, pose):
        nonlocal next_id
        rp, st = provider.lk(L, R, pts, pts)                                   # FindFeaturesInRight
        right = [rp[i] if st[i] else None for i in range(len(pts))]
        new = [i for i in range(len(pts)) if mp[i] < 0 and st[i]]
        if new:
            xyz, ok = provider.triangulate(pts[new].astype(np.float64), rp[new].astype(np.float64), pose_inv(pose))
            for j, i in enumerate(new):
                if ok[j]:
                    points[next_id] = xyz[j].copy(); mp[i] = next_id; next_id += 1
        return right

    def insert_keyframe(fid, pose, pts, mp, right):
        kfs.append(dict(frame=fid, pose=pose.copy(),
                        obs=[(int(mp[i]), pts[i].astype(np.float64), None if right[i] is None else right[i].astype(np.float64))
                             for i in range(len(pts)) if mp[i] >= 0]))

    def local_ba():
        act = kfs[-window:]
        ids = sorted({o[0] for kf in act for o in kf["obs"] if o[0] in points})
        if len(act) < 2 or not ids:
            return
        col = {m: j for j, m in enumerate(ids)}
        ep, el, uv, cam = [], [], [], []
        for a, kf in enumerate(act):
            for m, uL, uR in kf["obs"]:
                if m not in col:
                    continue
                ep.append(a); el.append(col[m]); uv.append(uL); cam.append(0)
                if uR is not None:
                    ep.append(a); el.append(col[m]); uv.append(uR); cam.append(1)
        pr = dict(P=len(act), L=len(ids), E=len(ep), poses=np.array([kf["pose"] for kf in act]), pose_fixed=None,
                  points=np.array([points[m] for m in ids]), point_fixed=None, edge_pose=np.array(ep, np.int32),
                  edge_point=np.array(el, np.int32), edge_uv=np.array(uv, np.float64), edge_cam=np.array(cam, np.uint8),
                  K=np.array(KITTI_K), cam_ext=synth.stereo_cam_ext())
        new_poses, new_pts = provider.ba(pr)
        for a, kf in enumerate(act):
            kf["pose"] = new_poses[a].copy()
        for m, j in col.items():
            points[m] = new_pts[j].copy()

    # ---- StereoInit (frontend.cpp:430-446) ----
    L, R = frames[0]
    pts = provider.detect(L, None, n_init)
    mp = -np.ones(len(pts), dtype=np.int64)
    pose = ident.copy()
    right = stereo_and_triangulate(L, R, pts, mp, pose)
    insert_keyframe(0, pose, pts, mp, right)
    poses.append(pose.copy()); tracked.append(int((mp >= 0).sum()))
    rel = ident.copy()
    last_L, last_pose = L, pose
    for fid in range(1, len(frames)):
        L, R = frames[fid]
        pred = pose_mul(rel, last_pose)                                        # constant-velocity model (:84-88)
        has = mp >= 0
        guess = pts.copy()
        if has.any():
            guess[has] = project(pred, [points[m] for m in mp[has]]).astype(np.float32)   # TrackLastFrame initial flow
        cur, st = provider.lk(last_L, L, pts, guess)
        keep = (st > 0) & has
        pts, mp = cur[keep], mp[keep]
        pose, inl = provider.pose_only(pred, np.array([points[m] for m in mp]), pts.astype(np.float64))
        pts, mp = pts[inl > 0], mp[inl > 0]                                    # outliers lose their map point (:279-293)
        rel = pose_mul(pose, pose_inv(last_pose))
        tracked.append(len(pts))
        if len(pts) < kf_below:                                                # TRACKING_BAD -> new keyframe (:119-126)
            mask = np.full((H, W), 255, np.uint8)
            for x, y in pts:                                                   # DetectFeatures' 21x21 boxes (:304-312)
                mask[max(int(y) - 10, 0):int(y) + 11, max(int(x) - 10, 0):int(x) + 11] = 0
            new = provider.detect(L, mask, n_new)
            pts = np.concatenate([pts, new]); mp = np.concatenate([mp, -np.ones(len(new), dtype=np.int64)])
            right = stereo_and_triangulate(L, R, pts, mp, pose)
            insert_keyframe(fid, pose, pts, mp, right)
            local_ba()
            pose = kfs[-1]["pose"].copy()
        poses.append(pose.copy())
        last_L, last_pose = L, pose
    return dict(poses=np.array(poses), keyframes=[kf["frame"] for kf in kfs], n_points=len(points), tracked=tracked)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--provider", default="gpu", choices=["gpu", "oracle"])
    args = ap.parse_args()
    frames, gt, _ = synth.make_lateral_sequence(n_frames=args.frames)
    if args.provider == "gpu":
        import ssvio_amd
        prov = GpuProvider(ssvio_amd.Context(0))
        run(prov, frames[:3])                                                  # warm-up (plans, allocations)
    else:
        from oracle import pyoracle as po
        po.build()
        prov = OracleProvider(po)
    t = time.perf_counter()
    r = run(prov, frames)
    dt = time.perf_counter() - t
    err = np.abs(r["poses"][:, 4:] - gt[:, 4:]).max()
    print(f"{args.provider}: {args.frames} frames in {dt:.3f} s = {args.frames / dt:.1f} frames/s (host loop in Python); "
          f"keyframes {r['keyframes']}, {r['n_points']} map points, tracked {r['tracked']}; max |t - t_gt| = {err:.4f} m")


if __name__ == "__main__":
    main()
