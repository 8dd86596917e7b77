"""bench.py's `next_rows`: driver-run lines for the rows SURVEY.md 8-F marks "next" and for A11, each with the work a call must do by
definition (algorithmic bytes / flops), what the GPU call takes, the fraction of the roof that implies, and the CPU beside it
(the reference's own classes where oracle/_ref travelled, else the oracle port) on ONE host core, bounded samples.

  N1  LK points/s          cv::calcOpticalFlowPyrLK as frontend.cpp:156-166 / 374-384 call it: 11x11 window, maxLevel 3, 30 / 0.01,
                           initial flow; 300 points (kitti_00.yaml num_features) on a 1241x376 pair
  A11 pose-only solves/s   FrontEnd::EstimateCurrentPose (frontend.cpp:184-300): 4 rounds x 10 LM iterations, M = 200 map points
  N2  vocabulary transform ORBVocabulary::transform (loopclosing.cpp:84, 633) on a vocabulary of ORBvoc's shape (k = 10, L = 6:
                           1 111 111 nodes), 2000 descriptors per call
  N3  pose-graph it/s      LoopClosing::PoseGraphOptimization (loopclosing.cpp:458-539): 500 keyframes, 20 LM iterations

Test / bench infrastructure, not part of the product."""
from __future__ import annotations

import time

import numpy as np

HBM_PEAK_GBS = 8000.0
F64_PEAK_TFLOPS = 78.6


def _time(fn, reps, warm=2):
    for _ in range(warm):
        fn()
    t = time.perf_counter()
    for _ in range(reps):
        r = fn()
    return (time.perf_counter() - t) / reps, r


def next_rows(ssvio_amd, ctx, cpu=True, voc_levels=6):
    from ssvio_amd import ba, lk, orb
    from ssvio_amd import voc as svoc
    from tools import synth
    out = {}
    po = None
    if cpu:
        from oracle import pyoracle as po
        po.build()
    have_ref = bool(po and po.have_ref())

    # ---- N1: LK -------------------------------------------------------------------------------------------------------------
    L, R, _ = synth.make_stereo_pair(seed=0)
    k, _ = orb.ORBextractor(ctx, nfeatures=300, nlevels=1).DetectAndCompute(L)
    pts = np.stack([k["x"], k["y"]], 1).astype(np.float32)
    guess = pts.copy(); guess[:, 0] -= 8.0                                  # a stereo disparity guess, as FindFeaturesInRight passes
    dt, r = _time(lambda: lk.calcOpticalFlowPyrLK(ctx, L, R, pts, guess), 30)
    dt_chain, _ = _time(lambda: lk.calcOpticalFlowPyrLK(ctx, None, R, pts, guess), 30)
    n = len(pts)
    px = [int(np.ceil(L.shape[0] / 2 ** l)) * int(np.ceil(L.shape[1] / 2 ** l)) for l in range(4)]
    # by definition: both pyramids read + written once (u8), Scharr derivatives of the previous pyramid written and read once
    # (2 x int16 per pixel), per point and level the 13x13 source patch + its derivatives and the search neighbourhood once
    lk_bytes = 2 * (sum(px) + sum(px[1:])) + 2 * 4 * sum(px) + n * 4 * (169 * 5 + 1024)
    row = {"points": n, "tracked": int(r[1].sum()), "ms_per_call": round(dt * 1e3, 4), "points_per_s": round(n / dt, 1),
           "ms_per_call_chained": round(dt_chain * 1e3, 4), "points_per_s_chained": round(n / dt_chain, 1),
           "algorithmic_bytes_per_call": int(lk_bytes), "hbm_frac": round(lk_bytes / dt / 1e9 / HBM_PEAK_GBS, 6),
           "bound": "latency: one wave per point, <= 4 levels x 30 dependent iterations; the pyramids of a 0.47 Mpx pair are three small launches",
           "what": "ssx_lk_track, host images in / host results out (chained: ssx_lk_track_next, the previous pyramid stays on the device)"}
    if po:
        tc, rc = _time(lambda: po.lk_track(L, R, pts, guess), 3, warm=1)
        row["cpu"] = {"ms_per_call": round(tc * 1e3, 3), "points_per_s": round(n / tc, 1), "cores": 1, "kind": "port",
                      "what": "oracle restatement of OpenCV's pyramidal LK (OpenCV itself cannot be built here)"}
        row["identical_to_cpu"] = bool(np.array_equal(rc[1], r[1]) and np.array_equal(rc[0][rc[1] > 0], r[0][r[1] > 0]))
    # the batched form (one frame of each of S streams in one call: ssx_lk_track_batch, chained jobs on S slots of the context)
    S_B = 64
    imgs = [np.ascontiguousarray(np.roll(R, s_ % 7, axis=1)) for s_ in range(S_B)]
    lk.track_batch(ctx, [dict(slot=s_, prev=L, next=imgs[s_], prev_pts=pts, next_pts=guess) for s_ in range(S_B)])          # fills the slots
    jobs_b = [dict(slot=s_, prev=None, next=imgs[(s_ + 1) % S_B], prev_pts=pts, next_pts=guess) for s_ in range(S_B)]
    prep = lk.PreparedTrackBatch(ctx, jobs_b)                                # job structs once, images pinned: what the StreamBatcher hands over
    dt_b, _ = _time(prep.run, 20, warm=2)
    prep.close()
    lk_bytes_chained = (sum(px) + sum(px[1:])) + 2 * 4 * sum(px) + n * 4 * (169 * 5 + 1024)      # one new pyramid per job
    row["batched"] = {"jobs_per_call": S_B, "ms_per_call": round(dt_b * 1e3, 4), "points_per_s": round(S_B * n / dt_b, 1),
                      "hbm_frac": round(S_B * lk_bytes_chained / dt_b / 1e9 / HBM_PEAK_GBS, 6),
                      "what": "ssx_lk_track_batch: 64 chained jobs (one frame of each of 64 streams) per call, images in pinned memory read by the "
                              "GPU over PCIe (images_on_device = 1, as ssvio_amd/host/stream_batcher.cpp calls it): the library call alone"}
    out["lk"] = row

    # ---- A11: pose-only LM --------------------------------------------------------------------------------------------------
    pp = synth.make_pose_only_problem(M=200, seed=1, frac_gross=0.05)
    dt, r = _time(lambda: ba.pose_only_opt(ctx, pp["pose"], pp["K"], pp["xyz"], pp["uv"]), 50)
    M = 200
    po_flops = 40 * M * 220.0                                               # 4 x 10 iterations x (error ~60 + J 2x6 ~70 + 27 accumulations x 2 + chi2 ~30) per edge
    row = {"map_points": M, "ms_per_solve": round(dt * 1e3, 4), "solves_per_s": round(1.0 / dt, 1), "lm_iterations_per_s": round(40.0 / dt, 1),
           "inliers": int(r["n_inliers"]), "algorithmic_flops_per_solve": int(po_flops),
           "flops_frac": round(po_flops / dt / 1e12 / F64_PEAK_TFLOPS, 8),
           "bound": "serial f64 latency: 40 dependent LM iterations of a 6x6 system inside ONE launch of one workgroup (k_pose_only)",
           "what": "ssx_pose_only_opt, host arrays in / pose + inlier flags out"}
    if po:
        which = "ref" if have_ref else "oracle"
        tc, rc = _time(lambda: po.pose_only(pp, which), 20, warm=1)
        row["cpu"] = {"ms_per_solve": round(tc * 1e3, 4), "solves_per_s": round(1.0 / tc, 1), "cores": 1,
                      "kind": "reference" if have_ref else "port",
                      "what": "the reference's own VertexPose + EdgeProjectionPoseOnly on g2o (oracle/_ref)" if have_ref else "oracle port"}
        row["max_pose_diff_vs_cpu"] = float(np.abs(rc["pose"] - r["pose"]).max())
    probs_b = [synth.make_pose_only_problem(M=200, seed=100 + s_, frac_gross=0.05) for s_ in range(64)]
    run_b = ba.pose_only_opt_batch(ctx, probs_b, prepared=True)
    dt_b, _ = _time(run_b, 30, warm=2)
    row["batched"] = {"problems_per_call": 64, "ms_per_call": round(dt_b * 1e3, 4), "solves_per_s": round(64 / dt_b, 1),
                      "flops_frac": round(64 * po_flops / dt_b / 1e12 / F64_PEAK_TFLOPS, 8),
                      "what": "ssx_pose_only_opt_batch: one workgroup per problem, one launch, one synchronisation (the library call alone)"}
    out["pose_only"] = row

    # ---- N2: vocabulary transform -------------------------------------------------------------------------------------------
    rng = np.random.default_rng(0)
    kk, Lv = 10, int(voc_levels)
    nn = sum(kk ** d for d in range(Lv + 1))
    parent = np.zeros(nn, np.int32); parent[0] = -1
    first = np.cumsum([0] + [kk ** d for d in range(Lv + 1)])
    for d in range(1, Lv + 1):
        ids = np.arange(first[d], first[d + 1]); parent[ids] = first[d - 1] + (ids - first[d]) // kk
    leaf = np.zeros(nn, np.uint8); leaf[first[Lv]:] = 1
    desc = rng.integers(0, 256, (nn, 32), dtype=np.uint8)
    weight = np.where(leaf, rng.uniform(0.5, 9, nn), 0.0)
    V = svoc.Vocabulary.from_arrays(ctx, kk, Lv, parent, leaf, desc, weight)
    feats = rng.integers(0, 256, (2000, 32), dtype=np.uint8)
    dt, r = _time(lambda: V.transform(feats), 30)
    nd = len(feats)
    voc_bytes = nd * (Lv * kk * 32 + 32 + 12)                                # L levels x k child descriptors + the query + (word, weight) out
    row = {"descriptors": nd, "vocabulary": {"k": kk, "L": Lv, "nodes": int(nn), "descriptor_MB": round(nn * 32 / 1e6, 1)},
           "ms_per_call": round(dt * 1e3, 4), "descriptors_per_s": round(nd / dt, 1), "words": int(len(r[0])),
           "algorithmic_bytes_per_call": int(voc_bytes), "hbm_frac": round(voc_bytes / dt / 1e9 / HBM_PEAK_GBS, 6),
           "bound": "latency: L dependent levels per descriptor (k_voc_words), then the BowVector assembled on the host in DBoW2's order",
           "what": "ssx_voc_transform, descriptors in / BowVector out; the vocabulary stays resident in HBM"}
    if po:
        vd = dict(parent=parent, is_leaf=leaf, desc=desc, weight=weight)
        tc, rc = _time(lambda: po.voc_transform(vd, feats), 3, warm=1)
        row["cpu"] = {"ms_per_call": round(tc * 1e3, 3), "descriptors_per_s": round(nd / tc, 1), "cores": 1, "kind": "port",
                      "what": "oracle restatement of TemplatedVocabulary::transform (DBoW2's own needs OpenCV)"}
        row["identical_to_cpu"] = bool(np.array_equal(rc[0], r[0]) and np.array_equal(rc[1], r[1]))
    V.close()
    out["voc_transform"] = row

    # ---- N3: pose-graph optimisation ----------------------------------------------------------------------------------------
    pg = synth.make_pose_graph_problem(P=500, n_loops=3, seed=12, meas_noise=0.01, drift=0.03, n_active=7)
    dt, r = _time(lambda: ba.pose_graph_opt(ctx, pg, iters=20), 3, warm=1)
    E = len(pg["ei"]); nit = max(int(r["n_iters"]), 1)
    # per LM iteration and edge: the error + 24 perturbed evaluations of g2o's central differences (~25 x 700 flop), two 6x6 blocks
    # and three block products (~1300 flop)
    pg_flops = nit * E * (25 * 700.0 + 1300.0)
    row = {"keyframes": 500, "edges": int(E), "lm_iterations": nit, "ms_per_solve": round(dt * 1e3, 3), "iterations_per_s": round(nit / dt, 1),
           "chi2_first_last": [float(r["chi2_initial"]), float(r["chi2_final"])], "algorithmic_flops_per_solve": int(pg_flops),
           "flops_frac": round(pg_flops / dt / 1e12 / F64_PEAK_TFLOPS, 8),
           "bound": "latency: per LM trial 2 x 8 levels of block cyclic reduction (the free keyframes couple to their neighbours only: loop edges end at "
                    "fixed keyframes, loopclosing.cpp:484-489) + one host synchronisation for the LM decision",
           "what": "ssx_pose_graph_opt, host arrays in / poses out"}
    if po:
        which = "ref" if have_ref else "oracle"
        t0 = time.perf_counter(); rc = po.pose_graph_opt(pg, which, iters=20); tc = time.perf_counter() - t0
        nc = max(int(rc["n_iters"]), 1)
        row["cpu"] = {"ms_per_solve": round(tc * 1e3, 2), "iterations_per_s": round(nc / tc, 2), "lm_iterations": nc, "cores": 1,
                      "kind": "reference" if have_ref else "port",
                      "what": "the reference's own EdgePoseGraph on g2o + LinearSolverEigen (oracle/_ref)" if have_ref else "oracle port"}
        row["max_pose_diff_vs_cpu"] = float(np.abs(rc["poses"] - r["poses"]).max())
    out["pose_graph"] = row
    return out
