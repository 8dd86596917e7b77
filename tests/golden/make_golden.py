"""Generate tests/golden/*.npz.

Run in the BUILD container (where /root/reference is mounted and oracle/_ref/libssvio_ref.so can be built):

    python tests/golden/make_golden.py

ref_*.npz hold outputs of the REAL reference arithmetic (oracle/ref_driver.cpp linked against the
reference's g2otypes.hpp / algorithm.hpp / g2o / Sophus / Eigen): they pin the CPU oracle and travel to the
GPU box, where /root/reference does not exist.  Inputs are regenerated from seeds by tools.synth (the
arrays that cannot be regenerated bit-for-bit are stored).  self_orb.npz holds outputs of OUR restatement
of the ORB path (no reference implementation of that half can run anywhere: OpenCV is absent) -- it is a
regression pin of the oracle, not a reference vector.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402
from tools import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

BA_CASES = {
    "tiny": dict(P=4, L=60, obs_per_lm=4, seed=2),
    "mid": dict(P=10, L=400, seed=3),
    "C3": dict(P=10, L=4000, seed=1),
    "gauge": dict(P=6, L=200, obs_per_lm=3, seed=9, fix_first_pose=True),
    # the reference's own window: Map.activeMap.size 12 (config/kitti_00.yaml:30), ~1500 landmarks; and the largest small window
    "win12": dict(P=12, L=1500, obs_per_lm=4, seed=5),
    "win16": dict(P=16, L=1200, obs_per_lm=5, seed=6),
}


PG_CASES = {
    "pg60": dict(P=60, n_loops=2, seed=11, meas_noise=0.02, drift=0.05),
    "pg200": dict(P=200, n_loops=3, seed=12, meas_noise=0.01, drift=0.03, n_active=7),
    "pg12": dict(P=12, n_loops=1, seed=13, meas_noise=0.03, drift=0.05, n_active=2),
}


def make_pose_graph_golden():
    """ref_pg.npz: LoopClosing::PoseGraphOptimization (loopclosing.cpp:458-539) run by the REAL reference classes
    (VertexPose, EdgePoseGraph, BlockSolver<6,6>, LinearSolverEigen, LM) + single-edge error / numeric Jacobians."""
    import ctypes as C
    g = {}
    for name, cfg in PG_CASES.items():
        pr = synth.make_pose_graph_problem(**cfg)
        r = po.pose_graph_opt(pr, "ref", iters=20)
        g[f"{name}_cfg"] = np.array([cfg["P"], cfg["n_loops"], cfg["seed"]])
        g[f"{name}_input_sum"] = np.array([pr["poses"].sum(), pr["meas"].sum(), float(pr["fixed"].sum())])
        for k in ("poses", "chi2", "lambdas", "trials", "edge_err"):
            g[f"{name}_{k}"] = r[k]
        g[f"{name}_n_iters"] = np.array(r["n_iters"])
    rng = np.random.default_rng(77)
    dbl_p = C.POINTER(C.c_double)
    ref = po.ref_lib()
    M, T0, T1, E, JI, JJ = [], [], [], [], [], []
    for _ in range(64):
        def rp():
            q = synth.small_rot_quat(rng.normal(0, 0.8, 3)); q /= np.linalg.norm(q)
            return np.concatenate([q, rng.normal(0, 5, 3)])
        m, a, b = rp(), rp(), rp()
        e = np.zeros(6); ji = np.zeros(36); jj = np.zeros(36)
        ref.ref_pg_edge_eval(m.ctypes.data_as(dbl_p), a.ctypes.data_as(dbl_p), b.ctypes.data_as(dbl_p), e.ctypes.data_as(dbl_p),
                             ji.ctypes.data_as(dbl_p), jj.ctypes.data_as(dbl_p))
        M.append(m); T0.append(a); T1.append(b); E.append(e); JI.append(ji.reshape(6, 6)); JJ.append(jj.reshape(6, 6))
    g.update(edge_M=np.array(M), edge_T0=np.array(T0), edge_T1=np.array(T1), edge_err=np.array(E), edge_Ji=np.array(JI), edge_Jj=np.array(JJ))
    np.savez_compressed(os.path.join(OUT, "ref_pg.npz"), **g)
    print("ref_pg.npz:", os.path.getsize(os.path.join(OUT, "ref_pg.npz")), "bytes,", len(g), "arrays")


def bow_inputs(case):
    """per-feature (word, weight) pairs as a tree descent would deliver them: repeated words, zero weights (stop words)"""
    rng = np.random.default_rng(900 + case)
    n = [1500, 40, 1, 700][case]
    word = rng.integers(0, [1000, 25, 5, 100000][case], n).astype(np.int32)
    weight = rng.uniform(0.0, 3.0, n)
    weight[rng.random(n) < 0.1] = 0.0
    return word, weight


def make_bow_golden():
    """ref_bow.npz: the BowVector of TemplatedVocabulary::transform (TemplatedVocabulary.h:1083-1122) built on the
    reference's OWN DBoW2::BowVector (BowVector.cpp compiled into oracle/_ref): ids ascending, L1-normalised values, for
    the four weighting types.  The tree descent and L1Scoring::score include OpenCV headers and stay unpinned."""
    g = {}
    for case in range(4):
        word, weight = bow_inputs(case)
        for weighting in range(4):
            ids, vals = po.bow_vector(word, weight, weighting, which="ref")
            g[f"bow{case}_w{weighting}_ids"] = ids; g[f"bow{case}_w{weighting}_vals"] = vals
    np.savez_compressed(os.path.join(OUT, "ref_bow.npz"), **g)
    print("ref_bow.npz:", os.path.getsize(os.path.join(OUT, "ref_bow.npz")), "bytes,", len(g), "arrays")


WINDOW_SCENARIO = dict(n_kf=14, n_active=5, new_per_kf=80, track_len=7, seed=6)


def make_window_golden():
    """ref_window.npz: the reference's backend over a DRIVE -- fourteen keyframes through a five-keyframe active map, every
    window optimised by the REAL g2o classes (po.ba_solve "ref"), every result written back the way backend.cpp:205-244 does
    (outlier observations unlinked, condemned map points deleted, the first-observer rule of backend.cpp:125-130 deciding which
    map points are fixed): the map bookkeeping is tools.mapmodel.ActiveMap, the arithmetic is the reference's.  The GPU test
    replays the same drive on an ssx_ba_window and must take the same decisions at every keyframe."""
    from tools.mapmodel import ActiveMap, make_window_scenario
    frames = make_window_scenario(**WINDOW_SCENARIO)
    m = ActiveMap(WINDOW_SCENARIO["n_active"])
    g = {"cfg": np.array([WINDOW_SCENARIO[k] for k in ("n_kf", "n_active", "new_per_kf", "track_len", "seed")])}
    g["input_sum"] = np.array([sum(f["pose"].sum() for f in frames), sum(float(np.sum([uv for _, uv in f["obs"]])) for f in frames),
                               sum(float(np.sum(list(f["new_points"].values()))) for f in frames)])
    margin = np.inf
    for r, fr in enumerate(frames):
        for l in fr["condemn"]:
            m.condemn(l)
        m.insert_keyframe(fr["kf_id"], fr["pose"], fr["obs"], fr["new_points"], fr["victim"])
        pr, kf_ids, lm_ids, e_feat = m.problem()
        res = po.ba_solve(pr, "ref")
        margin = min(margin, float(np.abs(res["edge_chi2"] - 5.891).min()))
        g[f"w{r}_sizes"] = np.array([pr["P"], pr["L"], pr["E"], int(pr["point_fixed"].sum())])
        g[f"w{r}_kf_ids"] = np.array(kf_ids); g[f"w{r}_lm_ids"] = np.array(lm_ids, dtype=np.int32)
        g[f"w{r}_fixed"] = np.packbits(pr["point_fixed"])
        g[f"w{r}_poses"] = res["poses"]; g[f"w{r}_points"] = res["points"]      # (complete: the open-loop test advances its map with them)
        g[f"w{r}_outlier"] = np.packbits(res["edge_outlier"]); g[f"w{r}_edge_chi2"] = res["edge_chi2"][::5]
        g[f"w{r}_chi2"] = res["chi2"]; g[f"w{r}_trials"] = res["trials"]; g[f"w{r}_rounds"] = np.array(res["rounds"])
        m.apply(kf_ids, lm_ids, e_feat, res["poses"], res["points"], res["edge_outlier"])
    g["stats"] = np.array([m.stats[k] for k in ("reentered", "fixed_by_rule", "condemned", "outlier_edges")])
    g["margin"] = np.array(margin)                 # smallest |chi2 - 5.891| over all edges of all rounds: how safe the decisions are
    assert margin > 5e-3 and m.stats["reentered"] > 0, (margin, m.stats)
    np.savez_compressed(os.path.join(OUT, "ref_window.npz"), **g)
    print("ref_window.npz:", os.path.getsize(os.path.join(OUT, "ref_window.npz")), "bytes,", len(g), "arrays; decision margin", margin, m.stats)


def main():
    assert po.have_ref(), "needs /root/reference (or a prebuilt oracle/_ref/libssvio_ref.so)"
    rng = np.random.default_rng(1234)
    g = {}
    # ---- BA end-to-end (Backend::OptimizeActiveMap defaults) ----
    for name, cfg in BA_CASES.items():
        pr = synth.make_ba_problem(**cfg)
        r = po.ba_solve(pr, "ref")
        # an edge between a fixed pose and a fixed landmark is inactive in g2o: its chi2 is never computed and the driver returns
        # uninitialised memory for it -- stored as 0 so that the file is reproducible (the tests skip these edges)
        inactive = pr["pose_fixed"][pr["edge_pose"]].astype(bool) & pr["point_fixed"][pr["edge_point"]].astype(bool)
        r["edge_chi2"] = np.where(inactive, 0.0, r["edge_chi2"])
        g[f"ba_{name}_cfg"] = np.array([cfg.get("P", 10), cfg.get("L", 4000), cfg.get("obs_per_lm", 5), cfg.get("seed", 1),
                                        int(cfg.get("fix_first_pose", False))])
        g[f"ba_{name}_rounds"] = np.array(r["rounds"])
        g[f"ba_{name}_chi2"] = r["chi2"]; g[f"ba_{name}_lam"] = r["lam"]; g[f"ba_{name}_trials"] = r["trials"]
        g[f"ba_{name}_poses"] = r["poses"]
        if pr["L"] <= 400:
            g[f"ba_{name}_points"] = r["points"]; g[f"ba_{name}_edge_chi2"] = r["edge_chi2"]
        else:
            sel = np.arange(0, pr["E"], 37)
            g[f"ba_{name}_edge_sel"] = sel; g[f"ba_{name}_edge_chi2"] = r["edge_chi2"][sel]
            g[f"ba_{name}_points"] = r["points"][::41]
        # checksum of the regenerated inputs so that a drift of the generator is detected
        g[f"ba_{name}_input_sum"] = np.array([pr["poses"].sum(), pr["points"].sum(), pr["edge_uv"].sum()])
    # ---- single-edge known answers (EdgeProjection error, g2o numeric Jacobian, Huber) ----
    n = 24
    poses = np.zeros((n, 7)); pts = np.zeros((n, 3)); uvs = np.zeros((n, 2)); cams = np.zeros(n, dtype=np.int64)
    E = np.zeros((n, 2)); JI = np.zeros((n, 2, 6)); JJ = np.zeros((n, 2, 3)); CH = np.zeros(n); RHO = np.zeros((n, 3))
    ext = synth.stereo_cam_ext()
    for i in range(n):
        q = synth.small_rot_quat(rng.uniform(-0.3, 0.3, 3)); q /= np.linalg.norm(q)
        poses[i] = np.concatenate([q, rng.uniform(-1, 1, 3)])
        pts[i] = [rng.uniform(-8, 8), rng.uniform(-3, 3), rng.uniform(5, 40)]
        pc = synth.quat_rot(q, pts[i]) + poses[i, 4:]
        uvs[i] = [718.856 * pc[0] / pc[2] + 607.1928, 718.856 * pc[1] / pc[2] + 185.2157]
        uvs[i] += rng.normal(0, 1.0 if i % 3 else 12.0, 2)     # every third edge lands beyond the Huber delta
        cams[i] = i % 2
        r = po.edge_eval(poses[i], pts[i], uvs[i], synth.KITTI_K, ext[cams[i]], which="ref")
        E[i] = r["e"]; JI[i] = r["Ji"]; JJ[i] = r["Jj"]; CH[i] = r["chi2"]; RHO[i] = r["rho"]
    g.update(edge_pose=poses, edge_pt=pts, edge_uv=uvs, edge_cam=cams, edge_e=E, edge_Ji=JI, edge_Jj=JJ,
             edge_chi2=CH, edge_rho=RHO)
    # ---- SE3 exp / oplus / action ----
    tang = [np.zeros(6), np.array([0.1, -0.2, 0.3, 0, 0, 0]), np.array([0.1, -0.2, 0.3, 1e-12, 0, 0]),
            np.array([1, 2, 3, 1e-6, -2e-6, 1e-6]), np.array([0.5, -0.1, 0.2, 0.3, -0.4, 0.5]),
            np.array([0, 0, 0, np.pi - 1e-6, 0, 0]), np.array([1e-9, 0, 0, 0, 1e-9, 0]),
            np.array([-2, 1, 0.5, 1.0, 1.0, -1.0])]
    g["se3_tangent"] = np.array(tang)
    g["se3_exp"] = np.array([po.se3_exp(t, "ref") for t in tang])
    base = np.concatenate([synth.small_rot_quat(np.array([0.2, -0.1, 0.3])), [0.3, -0.2, 1.5]])
    base[:4] /= np.linalg.norm(base[:4])
    g["se3_base"] = base
    g["se3_oplus"] = np.array([po.pose_oplus(base, t, "ref") for t in tang])
    p3 = np.array([1.5, -0.7, 12.0])
    g["se3_act_p"] = p3
    g["se3_act"] = np.array([po.se3_act(po.se3_exp(t, "ref"), p3, "ref") for t in tang])
    # ---- triangulation (algorithm.hpp:23-45) incl. rejects ----
    m = 96
    uL = np.stack([rng.uniform(50, 1190, m), rng.uniform(30, 340, m)], 1)
    d = rng.uniform(1, 200, m)
    uR = uL.copy(); uR[:, 0] -= d
    uR[:, 1] += rng.normal(0, 0.3, m)
    uR[::7, 1] += rng.uniform(3, 15, len(uR[::7]))     # gross vertical disparity -> sigma3/sigma2 reject
    uR[5::11, 0] = uL[5::11, 0] + rng.uniform(1, 30, len(uR[5::11]))  # negative disparity -> z <= 0
    uL = uL.astype(np.float32).astype(np.float64); uR = uR.astype(np.float32).astype(np.float64)
    t = po.triangulate(uL, uR, synth.KITTI_K, synth.KITTI_BASELINE, which="ref")
    g.update(tri_uvL=uL, tri_uvR=uR, tri_xyz=t["xyz"], tri_ok=t["ok"], tri_ratio=t["ratio"])
    # ---- pose-only (frontend.cpp:184-270) ----
    for name, cfg in {"po200": dict(M=200, seed=3), "po60": dict(M=60, seed=8, frac_gross=0.25)}.items():
        pp = synth.make_pose_only_problem(**cfg)
        r = po.pose_only(pp, "ref")
        g[f"{name}_cfg"] = np.array([cfg["M"], cfg["seed"], int(100 * cfg.get("frac_gross", 0.1))])
        g[f"{name}_pose"] = r["pose"]; g[f"{name}_inliers"] = r["inliers"]; g[f"{name}_n"] = np.array(r["n_inliers"])
        g[f"{name}_input_sum"] = np.array([pp["xyz"].sum(), pp["uv"].sum()])
    np.savez_compressed(os.path.join(OUT, "ref_golden.npz"), **g)
    print("ref_golden.npz:", os.path.getsize(os.path.join(OUT, "ref_golden.npz")), "bytes,", len(g), "arrays")
    make_pose_graph_golden()
    make_bow_golden()
    make_window_golden()

    # ---- self pins of the ORB restatement ----
    s = {}
    L, R, _ = synth.make_stereo_pair(seed=7, h=160, w=260, n_blobs=260)
    s["imgL"] = L; s["imgR"] = R
    prm = po.orb_params(nfeatures=300, nlevels=4)
    s["prm"] = np.array([300, 4, 20, 7]); s["prm_scale"] = np.array([1.2], dtype=np.float32)
    s["grid_cands"] = po.orb_grid_fast(L)
    s["detect"] = po.orb_detect(L, prm=prm)
    kL, dL = po.orb_extract(L, prm=prm); kR, dR = po.orb_extract(R, prm=prm)
    s["kL"] = kL; s["dL"] = dL; s["kR"] = kR; s["dR"] = dR
    idx, dist = po.stereo_match(kL, dL, kR, dR)
    s["match_idx"] = idx; s["match_dist"] = dist
    s["resize"] = po.resize_linear(L, 133, 217); s["gauss"] = po.gauss7(L)
    np.savez_compressed(os.path.join(OUT, "self_orb.npz"), **s)
    print("self_orb.npz:", os.path.getsize(os.path.join(OUT, "self_orb.npz")), "bytes")


if __name__ == "__main__":
    main()
