"""GPU parity of the bundle adjustment (ssx_ba_solve / ssx_ba_linearize) against the CPU oracle.

Tolerances: the north_star asks for reprojection residuals within 1e-4 px of the CPU reference; the HIP
path and the oracle run the same algorithm (analytic or numeric Jacobians) in f64 with different
summation orders, so the bars here are much tighter than that where the arithmetic allows.
"""
import ctypes as C
import os

import numpy as np
import pytest

from ssvio_amd import ba
from tools.synth import make_ba_problem

pytestmark = pytest.mark.gpu

RESID_TOL = 1e-4   # px, north_star tolerance for reprojection residuals

CASES = {
    "tiny": dict(P=4, L=60, obs_per_lm=4, seed=2),
    "mid": dict(P=10, L=400, seed=3),
    "C3": dict(P=10, L=4000, seed=1),
    "window12": dict(P=12, L=1500, obs_per_lm=4, seed=5),          # 72 unknowns: k_solve80 (5x5 tiles)
    "window16": dict(P=16, L=1200, obs_per_lm=5, seed=6),          # 96 unknowns: k_solve (6x6 tiles), the largest small window
    "window14fix": dict(P=15, L=900, obs_per_lm=4, seed=7, fix_first_pose=True),   # 84 unknowns
}


@pytest.mark.parametrize("name", list(CASES))
def test_linearize_blocks_match_oracle(ctx, po, name):
    pr = make_ba_problem(**CASES[name])
    g = ba.ba_linearize(ctx, pr, jac_mode=ba.JAC_ANALYTIC)
    o = po.ba_linearize(pr, jac_mode=0)
    for k in ("err", "Hll", "bl", "Hpl", "Hpp", "bp"):
        scale = max(np.abs(o[k]).max(), 1.0)
        assert np.abs(g[k] - o[k]).max() / scale < 1e-11, k
    assert abs(g["chi2"] - o["chi2"]) / o["chi2"] < 1e-12


@pytest.mark.parametrize("name", list(CASES))
def test_ba_solve_matches_oracle_analytic(ctx, po, name):
    pr = make_ba_problem(**CASES[name])
    g = ba.ba_solve(ctx, pr, jac_mode=ba.JAC_ANALYTIC)
    o = po.ba_solve(pr, "oracle", jac_mode=0)
    assert g["rounds"] == o["rounds"]
    assert g["n_iters"] == len(o["chi2"])
    assert (g["trials"] == o["trials"]).all()
    np.testing.assert_allclose(g["chi2"], o["chi2"], rtol=1e-8)
    np.testing.assert_allclose(g["lam"], o["lam"], rtol=1e-6)
    rg, ro = np.sqrt(g["edge_chi2"]), np.sqrt(o["edge_chi2"])
    assert np.abs(rg - ro).max() < RESID_TOL
    assert (g["edge_outlier"] == o["edge_outlier"]).all()
    assert np.abs(g["poses"] - o["poses"]).max() < 1e-7
    assert np.abs(g["points"] - o["points"]).max() < 1e-4


def test_ba_solve_numeric_mode_matches_oracle(ctx, po):
    """g2o-faithful mode (central differences, delta=1e-9): the Jacobian noise (~1e-6 relative) makes
    the trajectory less reproducible; chi2 still agrees to 1e-5 and every residual to 2e-3 px."""
    pr = make_ba_problem(P=10, L=400, seed=3)
    g = ba.ba_solve(ctx, pr, jac_mode=ba.JAC_NUMERIC_G2O)
    o = po.ba_solve(pr, "oracle", jac_mode=1)
    assert (g["trials"] == o["trials"]).all()
    np.testing.assert_allclose(g["chi2"], o["chi2"], rtol=1e-5)
    d = np.abs(np.sqrt(g["edge_chi2"]) - np.sqrt(o["edge_chi2"]))
    assert np.median(d) < 1e-5 and d.max() < 2e-3


def test_ba_fixed_pose_and_duplicates(ctx, po):
    """pose 0 fixed (gauge of a global BA) and landmarks seen by both cameras of one keyframe."""
    pr = make_ba_problem(P=6, L=200, obs_per_lm=3, seed=9, fix_first_pose=True)
    # add right-camera observations for the first 50 landmarks (same pose as their first edge)
    fx, fy, cx, cy = pr["K"]
    extra_pose, extra_pt, extra_uv = [], [], []
    for j in range(50):
        e = int(np.nonzero(pr["edge_point"] == j)[0][0])
        uv = pr["edge_uv"][e].copy()
        z = 20.0
        uv[0] -= fx * 0.537 / z
        extra_pose.append(pr["edge_pose"][e]); extra_pt.append(j); extra_uv.append(uv)
    pr["edge_pose"] = np.concatenate([pr["edge_pose"], np.array(extra_pose, dtype=np.int32)])
    pr["edge_point"] = np.concatenate([pr["edge_point"], np.array(extra_pt, dtype=np.int32)])
    pr["edge_uv"] = np.concatenate([pr["edge_uv"], np.array(extra_uv)])
    pr["edge_cam"] = np.concatenate([pr["edge_cam"], np.ones(50, dtype=np.uint8)])
    pr["E"] = len(pr["edge_pose"])
    g = ba.ba_solve(ctx, pr)
    o = po.ba_solve(pr, "oracle", jac_mode=0)
    assert (g["trials"] == o["trials"]).all()
    np.testing.assert_allclose(g["chi2"], o["chi2"], rtol=1e-8)
    assert np.abs(np.sqrt(g["edge_chi2"]) - np.sqrt(o["edge_chi2"])).max() < RESID_TOL
    np.testing.assert_array_equal(g["poses"][0], pr["poses"][0])


def test_ba_is_deterministic(ctx):
    pr = make_ba_problem(P=10, L=1000, seed=4)
    a = ba.ba_solve(ctx, pr)
    b = ba.ba_solve(ctx, pr)
    assert np.array_equal(a["poses"], b["poses"]) and np.array_equal(a["points"], b["points"])
    assert np.array_equal(a["edge_chi2"], b["edge_chi2"])


def test_ba_reduces_cost_and_recovers_truth(ctx):
    """size-independent property: robust chi2 decreases monotonically over accepted LM steps and the
    optimised poses are closer to ground truth than the noisy initial ones."""
    pr = make_ba_problem(P=10, L=4000, seed=11)
    g = ba.ba_solve(ctx, pr)
    assert (np.diff(g["chi2"]) <= 1e-9).all()
    e0 = np.abs(pr["poses"][:, 4:] - pr["gt_poses"][:, 4:]).max()
    e1 = np.abs(g["poses"][:, 4:] - pr["gt_poses"][:, 4:]).max()
    assert e1 < e0


def test_ba_errors(ctx):
    import ssvio_amd
    pr = make_ba_problem(P=4, L=20, obs_per_lm=3, seed=1)
    bad = dict(pr); bad["edge_pose"] = pr["edge_pose"].copy(); bad["edge_pose"][0] = 99
    with pytest.raises(ssvio_amd.SsxError):
        ba.ba_solve(ctx, bad)
    # empty edge set: nothing to optimise, state returned unchanged
    empty = dict(pr); empty["edge_pose"] = pr["edge_pose"][:0]; empty["edge_point"] = pr["edge_point"][:0]
    empty["edge_uv"] = pr["edge_uv"][:0]; empty["edge_cam"] = pr["edge_cam"][:0]; empty["E"] = 0
    r = ba.ba_solve(ctx, empty)
    np.testing.assert_array_equal(r["poses"], pr["poses"])
    np.testing.assert_array_equal(r["points"], pr["points"])


def test_allreduce_hook_world1_is_identity(ctx, po):
    """single GPU, world_size 1: the RCCL hook path (torch.distributed nccl all_reduce on zero-copy views of the
    library's device buffers, ctx on torch's current stream) must give exactly the plain result."""
    import os
    import torch
    import torch.distributed as dist
    import ssvio_amd
    from ssvio_amd import dist_ba
    pr = make_ba_problem(P=10, L=600, seed=6)
    plain = ba.ba_solve(ctx, pr)
    # the large-window path exchanges packed non-zero tiles of the reduced system (+ the tile pattern itself)
    pr_big = make_ba_problem(P=40, L=1500, obs_per_lm=5, seed=9, loop=False, fix_first_pose=True)
    plain_big = ba.ba_solve(ctx, pr_big, outer_rounds=1, iters=5)
    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29733")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
        created = True
    try:
        s = torch.cuda.Stream(device="cuda:0")
        with torch.cuda.stream(s):
            c2 = ssvio_amd.Context(0, stream=s.cuda_stream)
            hook = dist_ba.make_allreduce_hook(torch.device("cuda:0"))
            hooked = ba.ba_solve(c2, pr, allreduce=hook, rank=0, world_size=1)
            hooked_big = ba.ba_solve(c2, pr_big, outer_rounds=1, iters=5, allreduce=hook, rank=0, world_size=1)
            c2.close()
    finally:
        if created:
            dist.destroy_process_group()
    assert np.array_equal(plain["poses"], hooked["poses"]) and np.array_equal(plain["points"], hooked["points"])
    assert np.array_equal(plain["chi2"], hooked["chi2"]) and np.array_equal(plain["edge_chi2"], hooked["edge_chi2"])
    assert np.array_equal(plain_big["poses"], hooked_big["poses"]) and np.array_equal(plain_big["chi2"], hooked_big["chi2"])


@pytest.mark.parametrize("name", ["po200", "po60"])
def test_pose_only_matches_reference_golden_and_oracle(ctx, po, name):
    """FrontEnd::EstimateCurrentPose (frontend.cpp:184-270): the one-launch kernel against the vectors produced by the
    REAL reference (g2o + EdgeProjectionPoseOnly) and against the CPU oracle."""
    import os
    from tools.synth import make_pose_only_problem
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_golden.npz"))
    M, seed, fg = [int(v) for v in G[f"{name}_cfg"]]
    pp = make_pose_only_problem(M=M, seed=seed, frac_gross=fg / 100.0)
    g = ba.pose_only_opt(ctx, pp["pose"], pp["K"], pp["xyz"], pp["uv"])
    assert g["n_inliers"] == int(G[f"{name}_n"])
    np.testing.assert_array_equal(g["inliers"], G[f"{name}_inliers"])
    np.testing.assert_allclose(g["pose"], G[f"{name}_pose"], rtol=0, atol=1e-9)
    o = po.pose_only(pp)
    np.testing.assert_allclose(g["pose"], o["pose"], rtol=0, atol=2e-9)   # tree vs sequential f64 sums through 40 LM steps
    np.testing.assert_array_equal(g["inliers"], o["inliers"])


def test_pose_only_edge_cases(ctx, po):
    from tools.synth import make_pose_only_problem
    pp = make_pose_only_problem(M=700, seed=11, frac_gross=0.3)        # more edges than threads; many outliers
    g = ba.pose_only_opt(ctx, pp["pose"], pp["K"], pp["xyz"], pp["uv"])
    o = po.pose_only(pp)
    assert g["n_inliers"] == o["n_inliers"] and np.array_equal(g["inliers"], o["inliers"])
    np.testing.assert_allclose(g["pose"], o["pose"], rtol=0, atol=2e-9)
    assert np.abs(g["pose"] - pp["gt_pose"]).max() < 5e-3                 # recovers the true pose despite 30 % outliers
    for M in (1, 7, 256, 257, 512, 513, 1536, 1537, 2500):                # both register-resident variants and the generic kernel
        pp2 = make_pose_only_problem(M=M, seed=100 + M, frac_gross=0.1)
        g2 = ba.pose_only_opt(ctx, pp2["pose"], pp2["K"], pp2["xyz"], pp2["uv"])
        o2 = po.pose_only(pp2)
        assert g2["n_inliers"] == o2["n_inliers"] and np.array_equal(g2["inliers"], o2["inliers"]), M
        np.testing.assert_allclose(g2["pose"], o2["pose"], rtol=0, atol=1e-8 if M < 8 else 2e-9, err_msg=str(M))
    e = ba.pose_only_opt(ctx, pp["pose"], pp["K"], pp["xyz"][:0], pp["uv"][:0])
    assert e["n_inliers"] == 0 and np.array_equal(e["pose"], pp["pose"])
    z = ba.pose_only_opt(ctx, pp["pose"], pp["K"], pp["xyz"], pp["uv"], rounds=0)
    assert np.array_equal(z["pose"], pp["pose"])


def test_pose_only_batch_equals_single_calls(ctx):
    """ssx_pose_only_opt_batch -- one frame of each of n streams in one launch -- returns, per problem, the bits of
    ssx_pose_only_opt: both register-resident kernel classes, the generic kernel, an empty problem, 40 problems at once."""
    from tools.synth import make_pose_only_problem
    sizes = [300, 1, 512, 513, 0, 280, 1536, 1700, 64] + [250 + 3 * k for k in range(31)]
    probs = [make_pose_only_problem(M=max(M, 1), seed=500 + i, frac_gross=0.05 + 0.01 * (i % 20)) for i, M in enumerate(sizes)]
    for pr, M in zip(probs, sizes):
        if M == 0:
            pr["xyz"], pr["uv"] = pr["xyz"][:0], pr["uv"][:0]
    ones = [ba.pose_only_opt(ctx, pr["pose"], pr["K"], pr["xyz"], pr["uv"]) for pr in probs]
    got = ba.pose_only_opt_batch(ctx, probs)
    for i, (a, b) in enumerate(zip(got, ones)):
        assert np.array_equal(a["pose"], b["pose"]) and np.array_equal(a["inliers"], b["inliers"]) and a["n_inliers"] == b["n_inliers"], (i, sizes[i])
    assert ba.pose_only_opt_batch(ctx, []) == []


BIG_CASES = {
    "P24_gauge": dict(P=24, L=1200, obs_per_lm=5, seed=31, fix_first_pose=True),
    "P40": dict(P=40, L=3000, obs_per_lm=6, seed=32, fix_first_pose=True),
    "P70_npad_not_multiple": dict(P=70, L=2500, obs_per_lm=4, seed=33, fix_first_pose=True),
}


@pytest.mark.parametrize("name", list(BIG_CASES))
def test_large_window_linearize_and_solve_match_oracle(ctx, po, name):
    """windows beyond 16 free poses: pose-major pose blocks, block-sparse Schur, blocked Cholesky on the matrix cores
    (v_mfma_f64_16x16x4_f64), blocked back-substitution -- against the CPU oracle (dense Cholesky)."""
    pr = make_ba_problem(**BIG_CASES[name])
    g = ba.ba_linearize(ctx, pr)
    o = po.ba_linearize(pr, jac_mode=0)
    # edges whose two vertices are fixed are inactive (g2o never evaluates them; the oracle leaves their error at 0)
    act = ~(pr["pose_fixed"][pr["edge_pose"]].astype(bool) & pr["point_fixed"][pr["edge_point"]].astype(bool))
    assert np.abs(g["err"][act] - o["err"][act]).max() < 1e-9
    for k in ("Hll", "bl", "Hpl", "Hpp", "bp"):
        scale = max(np.abs(o[k]).max(), 1.0)
        assert np.abs(g[k] - o[k]).max() / scale < 1e-11, k
    g = ba.ba_solve(ctx, pr, outer_rounds=1)
    o = po.ba_solve(pr, "oracle", jac_mode=0, outer_rounds=1)
    assert g["n_iters"] == len(o["chi2"])
    assert (g["trials"] == o["trials"]).all()
    np.testing.assert_allclose(g["chi2"], o["chi2"], rtol=1e-7)
    np.testing.assert_allclose(g["lam"], o["lam"], rtol=1e-5)
    assert np.abs(np.sqrt(g["edge_chi2"]) - np.sqrt(o["edge_chi2"]))[act].max() < RESID_TOL
    assert np.abs(g["poses"] - o["poses"]).max() < 1e-6


def test_global_ba_c4_shape_properties(ctx):
    """BASELINE config 4 shape at full pose count on one GPU (500 keyframes on a 400 m loop, 6 observations per
    landmark, pose 0 fixed; 20 000 landmarks keep the test short): size-independent properties -- every accepted LM
    step lowers the robust cost, the result is deterministic, the fixed pose does not move, poses move towards truth."""
    pr = make_ba_problem(P=500, L=20000, obs_per_lm=6, seed=41, loop=True, fix_first_pose=True)
    a = ba.ba_solve(ctx, pr, outer_rounds=1, iters=5)
    assert a["n_iters"] >= 1 and (np.diff(a["chi2"]) <= 1e-6).all()
    np.testing.assert_array_equal(a["poses"][0], pr["poses"][0])
    e0 = np.abs(pr["poses"][:, 4:] - pr["gt_poses"][:, 4:]).mean()
    e1 = np.abs(a["poses"][:, 4:] - pr["gt_poses"][:, 4:]).mean()
    assert e1 < e0
    b = ba.ba_solve(ctx, pr, outer_rounds=1, iters=5)
    assert np.array_equal(a["poses"], b["poses"]) and np.array_equal(a["points"], b["points"])


# ---- directly against the REAL reference (tests/golden/ref_golden.npz = g2o + the reference's own g2otypes.hpp,
# produced by tests/golden/make_golden.py from oracle/_ref): no oracle in between -------------------------------------
GOLD_BA = ["tiny", "mid", "C3", "gauge"]


def _golden_problem(G, name):
    P, L, k, seed, fix = [int(v) for v in G[f"ba_{name}_cfg"]]
    pr = make_ba_problem(P=P, L=L, obs_per_lm=k, seed=seed, fix_first_pose=bool(fix))
    s = np.array([pr["poses"].sum(), pr["points"].sum(), pr["edge_uv"].sum()])
    np.testing.assert_allclose(s, G[f"ba_{name}_input_sum"], rtol=1e-13)        # the generator did not drift
    return pr


@pytest.mark.parametrize("name", GOLD_BA)
@pytest.mark.parametrize("jac", [ba.JAC_NUMERIC_G2O, ba.JAC_ANALYTIC])
def test_ba_solve_matches_reference_golden(ctx, name, jac, record_property):
    """Backend::OptimizeActiveMap (backend.cpp:175-227) on the GPU against the vectors of the compiled reference:
    identical outer rounds, LM iteration and trial counts, lambda and robust chi2 trajectory, final poses, and the
    DISTRIBUTION of the per-edge residual differences.  north_star's bar is 1e-4 px; the reference linearises with
    g2o's central differences (delta = 1e-9, base_binary_edge.hpp:144-212), whose ~1e-6 relative Jacobian noise is
    amplified on barely observable landmarks, so two faithful implementations agree on >= 99 % of the residuals to
    1e-4 px and on the worst one to ~1e-3 px (tests/test_oracle_ba.py header): the whole distribution is asserted."""
    import os
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_golden.npz"))
    pr = _golden_problem(G, name)
    g = ba.ba_solve(ctx, pr, jac_mode=jac)
    assert g["rounds"] == int(G[f"ba_{name}_rounds"])
    assert g["n_iters"] == len(G[f"ba_{name}_chi2"])
    np.testing.assert_array_equal(g["trials"], G[f"ba_{name}_trials"])
    # (chi2 / lambda trajectories, final poses and the residual distribution: _assert_within_noise_floor below)
    ec = g["edge_chi2"]
    if f"ba_{name}_edge_sel" in G:
        ec = ec[G[f"ba_{name}_edge_sel"]]
    d = np.abs(np.sqrt(ec) - np.sqrt(G[f"ba_{name}_edge_chi2"]))
    if f"ba_{name}_edge_sel" not in G:
        act = ~(pr["pose_fixed"][pr["edge_pose"]].astype(bool) & pr["point_fixed"][pr["edge_point"]].astype(bool))
        d = d[act]                                   # all-fixed edges are inactive in g2o: the reference never evaluates them
    frac = float((d <= RESID_TOL).mean())
    record_property("residual_diff_px", dict(median=float(np.median(d)), p99=float(np.percentile(d, 99)), max=float(d.max()),
                                             frac_le_1e_4=frac))
    print(f"[{name} jac={jac}] |r_gpu - r_ref| px: median {np.median(d):.2e} p99 {np.percentile(d, 99):.2e} max {d.max():.2e} "
          f"<=1e-4: {100 * frac:.2f} %")
    # The bars are not hand-set: tests/golden/ref_noise_floor.npz (make_noise_floor.py) holds, per golden window and Jacobian mode,
    # how far the CPU oracle -- a statement-by-statement restatement -- lands from the same reference vectors: the noise floor of
    # "two faithful implementations" (the reference's central differences, delta = 1e-9, amplified by barely constrained
    # landmarks).  The GPU must stay within K x that floor: K = 2 for the median, the 99th and 99.9th percentile and the
    # trajectories, K = 4 for the single worst residual (the maximum of a heavy-tailed sample of a few hundred to 20 000 values
    # fluctuates by more than the body of the distribution), plus a floor of 1e-6 px where the oracle's own figure is rounding.
    _assert_within_noise_floor(name, jac, g, G, d, record_property)
    if name == "C3" and jac == ba.JAC_ANALYTIC:
        assert d.max() < RESID_TOL                              # north_star: residuals within 1e-4 px of the CPU reference


_NOISE = None


def _assert_within_noise_floor(name, jac, g, G, d, record_property=None):
    global _NOISE
    if _NOISE is None:
        _NOISE = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_noise_floor.npz"))
    K, KM = float(_NOISE["K_body"]), float(_NOISE["K_max"])        # 2 and 4, stored with the floors (tests/golden/make_noise_floor.py)
    key = f"{name}_jac{1 if jac == ba.JAC_NUMERIC_G2O else 0}"
    med, p99, p999, mx, frac = _NOISE[key + "_resid"]
    got = dict(median=float(np.median(d)), p99=float(np.percentile(d, 99)), p999=float(np.percentile(d, 99.9)), max=float(d.max()),
               frac_le_1e_4=float((d <= RESID_TOL).mean()))
    if record_property:
        record_property(f"residual_diff_px_{key}", dict(got, oracle_vs_reference=dict(median=float(med), p99=float(p99), p999=float(p999), max=float(mx),
                                                                                        frac_le_1e_4=float(frac))))
    print(f"[{key}] |r_gpu - r_ref| px: median {got['median']:.2e} ({med:.2e}) p99 {got['p99']:.2e} ({p99:.2e}) p99.9 {got['p999']:.2e} ({p999:.2e}) "
          f"max {got['max']:.2e} ({mx:.2e}) <=1e-4: {100 * got['frac_le_1e_4']:.2f} % ({100 * frac:.2f} %)   [noise floor = oracle vs reference in brackets]")
    FLOOR = 1e-6
    assert got["median"] <= K * med + FLOOR, (key, "median", got["median"], med)
    assert got["p99"] <= K * (p99 if len(d) >= 1000 else p999) + FLOOR, (key, "p99", got["p99"], p99, p999)
    assert got["p999"] <= K * p999 + FLOOR, (key, "p99.9", got["p999"], p999)
    assert got["max"] <= KM * mx + FLOOR, (key, "max", got["max"], mx)
    if KM * mx < RESID_TOL:
        # where the restatement itself stays KM times inside north_star's tolerance, the GPU is held to the tolerance itself, per edge
        assert got["max"] < RESID_TOL, (key, "every residual within 1e-4 px of the reference", got["max"])
    assert got["frac_le_1e_4"] >= 1.0 - K * (1.0 - frac) - 1e-9, (key, "fraction within 1e-4 px", got["frac_le_1e_4"], frac)
    n = min(len(g["chi2"]), len(G[f"ba_{name}_chi2"]))
    chi2_rel = float(np.abs(g["chi2"][:n] / G[f"ba_{name}_chi2"][:n] - 1).max())
    poses = float(np.abs(g["poses"] - G[f"ba_{name}_poses"]).max())
    assert chi2_rel <= K * float(_NOISE[key + "_chi2_rel"]) + 1e-9, (key, "chi2 trajectory", chi2_rel, float(_NOISE[key + "_chi2_rel"]))
    # lambda follows 1 - (2 rho - 1)^3 of the gain ratio and is a PRODUCT of ten such factors: the last digits of every rho show up
    # amplified and compound (tiny: 0.25 % for the oracle, 0.9 % for the GPU after ten iterations) -- 2 K for this one
    lam_rel = float(np.abs(g["lam"][:n] / G[f"ba_{name}_lam"][:n] - 1).max())
    assert lam_rel <= 2 * K * float(_NOISE[key + "_lam_rel"]) + 1e-6, (key, "lambda trajectory", lam_rel, float(_NOISE[key + "_lam_rel"]))
    assert poses <= K * float(_NOISE[key + "_poses"]) + 1e-8, (key, "poses", poses, float(_NOISE[key + "_poses"]))


def test_ba_batch_matches_reference_golden(ctx, record_property):
    """The BENCH path (ssx_ba_solve_batch: every golden window in one call, one grid dimension = the window, device-driven LM)
    directly against the vectors of the compiled reference, including the reference's own window size (12 keyframes,
    config/kitti_00.yaml:30) and the largest small window (16): rounds, LM iterations, trial counts, chi2 / lambda trajectory,
    final poses, per-edge residuals, each within three times the noise floor of tests/golden/ref_noise_floor.npz (see
    _assert_within_noise_floor); with analytic Jacobians every residual of C3 and win12 inside north_star's 1e-4 px."""
    import os
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_golden.npz"))
    names = ["tiny", "mid", "C3", "gauge", "win12", "win16"]
    probs = [_golden_problem(G, n) for n in names]
    for jac in (ba.JAC_NUMERIC_G2O, ba.JAC_ANALYTIC):
        out = ba.BaBatch(ctx, probs, jac_mode=jac).solve()
        for name, pr, g in zip(names, probs, out["results"]):
            assert g["rounds"] == int(G[f"ba_{name}_rounds"]) and g["n_iters"] == len(G[f"ba_{name}_chi2"]), name
            np.testing.assert_array_equal(g["trials"], G[f"ba_{name}_trials"])
            ec = g["edge_chi2"]
            act = ~(pr["pose_fixed"][pr["edge_pose"]].astype(bool) & pr["point_fixed"][pr["edge_point"]].astype(bool))
            if f"ba_{name}_edge_sel" in G:
                sel = G[f"ba_{name}_edge_sel"]
                ec, act = ec[sel], act[sel]
            d = np.abs(np.sqrt(ec) - np.sqrt(G[f"ba_{name}_edge_chi2"]))[act]
            frac, p99 = float((d <= RESID_TOL).mean()), float(np.percentile(d, 99))
            record_property(f"{name}_jac{jac}", dict(median=float(np.median(d)), p99=p99, max=float(d.max()), frac_le_1e_4=frac))
            print(f"[batch {name} jac={jac}] |r_gpu - r_ref| px: median {np.median(d):.2e} p99 {p99:.2e} max {d.max():.2e} <=1e-4: {100 * frac:.2f} %")
            # the bars: K x the oracle's own distance from the same reference vectors (tests/golden/ref_noise_floor.npz), as above;
            # at the realistic windows with analytic Jacobians every residual additionally inside north_star's 1e-4 px where the
            # oracle's is (C3, win12; at win16 one of the 163 sampled residuals sits at 1.18e-4 px for the oracle too)
            _assert_within_noise_floor(name, jac, g, G, d)
            if jac == ba.JAC_ANALYTIC and name in ("C3", "win12"):
                assert d.max() < RESID_TOL, (name, d.max())


def test_global_ba_c4_full_size(ctx):
    """BASELINE configs[3] at its full size on ONE GPU: 500 keyframes on a loop x 80 000 landmarks x 480 000 edges,
    pose 0 fixed.  Size-independent properties: every accepted LM step lowers the robust cost, the fixed pose does not
    move, the poses move towards the truth, and two runs give identical bits."""
    pr = make_ba_problem(P=500, L=80000, obs_per_lm=6, seed=4, loop=True, fix_first_pose=True)
    assert pr["E"] == 480000
    a = ba.ba_solve(ctx, pr, outer_rounds=1, iters=6, want_edges=False)
    assert a["n_iters"] >= 2 and (np.diff(a["chi2"]) <= 1e-6).all()
    np.testing.assert_array_equal(a["poses"][0], pr["poses"][0])
    e0 = np.abs(pr["poses"][:, 4:] - pr["gt_poses"][:, 4:]).mean()
    e1 = np.abs(a["poses"][:, 4:] - pr["gt_poses"][:, 4:]).mean()
    assert e1 < e0
    b = ba.ba_solve(ctx, pr, outer_rounds=1, iters=6, want_edges=False)
    assert np.array_equal(a["poses"], b["poses"]) and np.array_equal(a["points"], b["points"])


def test_native_rccl_comm_world1_is_identity(ctx, po):
    """RCCL inside libssx.so (ssx_comm_*): a one-rank communicator created by the library itself; ssx_ba_solve with
    options.comm issues ncclAllReduce on the ctx stream at every exchange point of the small-window AND the large-window
    path.  With one rank the sum is the identity: results must be bit-identical to the plain call."""
    import ctypes as C
    import ssvio_amd
    from ssvio_amd import dist_ba
    pr = make_ba_problem(P=10, L=600, seed=6)
    pr_big = make_ba_problem(P=40, L=1500, obs_per_lm=5, seed=9, loop=False, fix_first_pose=True)
    plain = ba.ba_solve(ctx, pr)
    plain_big = ba.ba_solve(ctx, pr_big, outer_rounds=1, iters=5)
    c2 = ssvio_amd.Context(0)
    comm = dist_ba.init_native_comm(c2, 0, 1)
    r, w = C.c_int32(-1), C.c_int32(-1)
    c2.check(c2.lib.ssx_comm_info(comm, C.byref(r), C.byref(w)))
    assert (r.value, w.value) == (0, 1)
    with_comm = ba.ba_solve(c2, pr, comm=comm)
    with_comm_big = ba.ba_solve(c2, pr_big, outer_rounds=1, iters=5, comm=comm)
    # the raw collective on a device buffer of the caller
    import torch
    t = torch.arange(1000, dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    c2.check(c2.lib.ssx_comm_allreduce_sum(c2.handle, comm, C.c_void_p(t.data_ptr()), C.c_size_t(1000)))
    c2.synchronize()
    assert torch.equal(t.cpu(), torch.arange(1000, dtype=torch.float64))
    dist_ba.destroy_native_comm(c2, comm)
    c2.close()
    for a, b in ((plain, with_comm), (plain_big, with_comm_big)):
        assert np.array_equal(a["poses"], b["poses"]) and np.array_equal(a["points"], b["points"])
        assert np.array_equal(a["chi2"], b["chi2"]) and np.array_equal(a["trials"], b["trials"])


def test_phase_statistics(ctx):
    """ssx_ba_options.collect_stats: GPU time per phase (the G2OBatchStatistics fields), non-zero where work happened
    and adding up to less than the whole call."""
    pr = make_ba_problem(P=10, L=1000, seed=4)
    r = ba.ba_solve(ctx, pr, collect_stats=True)
    ph = r["phase_ms"]
    assert all(ph[k] > 0 for k in ("linearize", "schur", "linear_solution", "update", "reduce"))
    assert sum(ph.values()) <= r["ms_total"] * 1.05
    plain = ba.ba_solve(ctx, pr)
    assert plain["phase_ms"] is None and np.array_equal(plain["poses"], r["poses"])


# ---- large windows, the two reduced-system solvers (ssx_ba_options.large_solver) ------------------------------------------
BAND_CASES = {
    "chain_K1": dict(P=30, L=1500, obs_per_lm=5, seed=51, fix_first_pose=True),                       # 29 free poses: one workgroup
    "chain_dissected": dict(P=120, L=4000, obs_per_lm=6, seed=52, fix_first_pose=True),               # segments + separator system
    "closed_loop": dict(P=90, L=3500, obs_per_lm=5, seed=53, loop=True, wrap=True, fix_first_pose=True),   # the band wraps around
    "closed_loop_gauge_free": dict(P=64, L=2500, obs_per_lm=4, seed=54, loop=True, wrap=True),        # no fixed pose (as the reference)
    "narrow_band": dict(P=70, L=2000, obs_per_lm=2, seed=55, fix_first_pose=True),                    # w = 1
}


@pytest.mark.parametrize("name", list(BAND_CASES))
def test_band_solver_matches_oracle_and_tile_solver(ctx, po, name):
    """trajectory-shaped windows: the sliding-window block Cholesky with nested dissection (ba_band.inc) against the CPU
    oracle (dense Cholesky) and against the 64x64-tile solver of the same library."""
    pr = make_ba_problem(**BAND_CASES[name])
    kw = dict(outer_rounds=1, iters=6)
    g = ba.ba_solve(ctx, pr, large_solver=2, **kw)                   # 2 = band: fails if the structure is not banded
    t = ba.ba_solve(ctx, pr, large_solver=1, **kw)                   # 1 = tiles
    a = ba.ba_solve(ctx, pr, **kw)                                   # auto must pick the band solver: same bits
    assert np.array_equal(a["poses"], g["poses"]) and np.array_equal(a["chi2"], g["chi2"])
    o = po.ba_solve(pr, "oracle", jac_mode=0, **kw)
    act = ~(pr["pose_fixed"][pr["edge_pose"]].astype(bool) & pr["point_fixed"][pr["edge_point"]].astype(bool))
    gauged = bool(pr["pose_fixed"].any())
    for r, tag in ((g, "band"), (t, "tiles")):
        assert r["n_iters"] == len(o["chi2"]), tag
        assert (r["trials"] == o["trials"]).all(), tag
        # a gauge-free window has 7 unobservable directions, along which the step is set by rounding: three correct
        # solvers then drift apart by ~1e-6 relative in the cost after a few iterations (oracle vs tiles just the same)
        np.testing.assert_allclose(r["chi2"], o["chi2"], rtol=1e-7 if gauged else 2e-5, err_msg=tag)
        np.testing.assert_allclose(r["lam"], o["lam"], rtol=1e-5 if gauged else 1e-3, err_msg=tag)
        if gauged:
            assert np.abs(np.sqrt(r["edge_chi2"]) - np.sqrt(o["edge_chi2"]))[act].max() < RESID_TOL, tag
    if gauged:
        assert np.abs(g["poses"] - o["poses"]).max() < 1e-6 and np.abs(g["poses"] - t["poses"]).max() < 1e-7
    b = ba.ba_solve(ctx, pr, large_solver=2, **kw)
    assert np.array_equal(b["poses"], g["poses"]) and np.array_equal(b["points"], g["points"])       # deterministic


def test_unordered_keyframes_fall_back_to_the_tile_solver(ctx, po):
    """the same kind of graph with the keyframes numbered at random: no band in the pose order -> auto takes the tile
    solver, and asking for the band solver is refused"""
    import ssvio_amd
    pr = make_ba_problem(P=48, L=2000, obs_per_lm=5, seed=57, fix_first_pose=True, shuffle_poses=True)
    kw = dict(outer_rounds=1, iters=5)
    a = ba.ba_solve(ctx, pr, **kw)
    t = ba.ba_solve(ctx, pr, large_solver=1, **kw)
    assert np.array_equal(a["poses"], t["poses"]) and np.array_equal(a["chi2"], t["chi2"])
    o = po.ba_solve(pr, "oracle", jac_mode=0, **kw)
    assert (a["trials"] == o["trials"]).all()
    np.testing.assert_allclose(a["chi2"], o["chi2"], rtol=1e-7)
    with pytest.raises(ssvio_amd.SsxError):
        ba.ba_solve(ctx, pr, large_solver=2, **kw)


def test_c4_shape_band_equals_tiles(ctx):
    """BASELINE configs[3] shape (500 keyframes on a loop, pose 0 fixed) at 12 000 landmarks: both reduced-system solvers
    take the same LM decisions and end at the same poses"""
    pr = make_ba_problem(P=500, L=12000, obs_per_lm=6, seed=43, loop=True, fix_first_pose=True)
    g = ba.ba_solve(ctx, pr, outer_rounds=1, iters=5, want_edges=False, large_solver=2)
    t = ba.ba_solve(ctx, pr, outer_rounds=1, iters=5, want_edges=False, large_solver=1)
    assert np.array_equal(g["trials"], t["trials"])
    np.testing.assert_allclose(g["chi2"], t["chi2"], rtol=1e-9)
    assert np.abs(g["poses"] - t["poses"]).max() < 1e-8


@pytest.mark.parametrize("P,obs,wrap,fix", [(40, 2, False, True), (41, 3, True, False), (57, 4, True, True), (63, 5, False, True),
                                            (64, 6, True, True), (97, 6, False, False), (131, 6, True, True), (48, 6, True, True)])
def test_block_cyclic_reduction_shapes_equal_tiles(ctx, P, obs, wrap, fix):
    """The block cyclic reduction of the cyclic band (ba_bcr.inc) over its shapes: bandwidths w = 1 .. 5 (super-blocks of 24 and of 36
    unknowns), pose counts that leave 0 .. w - 1 poses over (super-blocks of w + 1 poses at the odd positions), odd and even numbers
    of super-blocks (the level with two blocks coupled in both directions, the wrap-around), open and closed trajectories, with and
    without a fixed keyframe -- against the 64 x 64-tile sparse Cholesky on the same system: identical LM decisions, the same cost
    and poses."""
    pr = make_ba_problem(P=P, L=60 * P, obs_per_lm=obs, seed=900 + P, loop=True, wrap=wrap, fix_first_pose=fix)
    g = ba.ba_solve(ctx, pr, outer_rounds=1, iters=4, want_edges=False, large_solver=2)
    t = ba.ba_solve(ctx, pr, outer_rounds=1, iters=4, want_edges=False, large_solver=1)
    assert np.array_equal(g["trials"], t["trials"])
    np.testing.assert_allclose(g["chi2"], t["chi2"], rtol=1e-8)
    # (obs = 2: every landmark ties just two neighbouring keyframes -- a chain whose reduced system is the worst conditioned of the
    # set; two exact solvers then differ by 1.4e-7 in a pose at equal cost)
    assert np.abs(g["poses"] - t["poses"]).max() < (3e-7 if obs == 2 else 1e-7)
    again = ba.ba_solve(ctx, pr, outer_rounds=1, iters=4, want_edges=False, large_solver=2)
    assert np.array_equal(again["poses"], g["poses"]) and np.array_equal(again["chi2"], g["chi2"])      # bit-deterministic


def test_block_cyclic_reduction_on_a_busy_chip_returns_the_idle_bits(ctx):
    """The BCR_S workgroups that eliminate one super-block share its stores; none of them may write what a sibling still reads (the
    coupling blocks E live in two buffers: bcr_e_buf).  On an idle GPU all workgroups of a level start together and a violation
    stays invisible, so the solve is repeated while another context keeps every CU busy with a resident batch of local windows on
    its own stream (the siblings of a super-block are then dispatched apart): every repeat returns the bits of the idle solve."""
    import threading
    import ssvio_amd
    big = [make_ba_problem(P=131, L=60 * 131, obs_per_lm=6, seed=1031, loop=True, wrap=True, fix_first_pose=True),
           make_ba_problem(P=500, L=12000, obs_per_lm=6, seed=43, loop=True, fix_first_pose=True)]
    kw = dict(outer_rounds=1, iters=4, want_edges=False, large_solver=2)
    idle = [ba.ba_solve(ctx, pr, **kw) for pr in big]
    ctx2 = ssvio_amd.Context(0)
    load = ba.BaBatch(ctx2, [make_ba_problem(P=10, L=2000, obs_per_lm=5, seed=70 + k) for k in range(96)], resident=True, with_edge_errors=False)
    stop, n_load, err = threading.Event(), [0], []

    def keep_busy():
        try:
            while not stop.is_set():
                load.solve(download=False)
                n_load[0] += 1
        except Exception as exc:                              # noqa: BLE001 -- reported by the test's thread
            err.append(exc)

    th = threading.Thread(target=keep_busy)
    th.start()
    try:
        for _ in range(6):
            for pr, ref in zip(big, idle):
                g = ba.ba_solve(ctx, pr, **kw)
                assert np.array_equal(g["trials"], ref["trials"]) and np.array_equal(g["chi2"], ref["chi2"])
                assert np.array_equal(g["poses"], ref["poses"]) and np.array_equal(g["points"], ref["points"])
    finally:
        stop.set()
        th.join()
        load.close()
        ctx2.close()
    assert not err and n_load[0] >= 2, (err, n_load)


def test_batch_groups_do_not_change_the_bits(ctx):
    """A batch of >= 8 windows runs as groups of windows on several streams (ssx_ba_batch_groups / _set_groups): every
    grouping returns, per window, the bits of ssx_ba_solve."""
    probs = [make_ba_problem(P=4 + (k % 7), L=120 + 90 * k, obs_per_lm=3 + (k % 3), seed=300 + k, fix_first_pose=(k % 4 == 0),
                             frac_gross=0.4 if k == 5 else 0.05) for k in range(11)]
    ones = [ba.ba_solve(ctx, pr) for pr in probs]
    res = ba.BaBatch(ctx, probs, resident=True)
    assert res.groups == 2                                   # the default for 8 or more windows
    for g in (0, 1, 2, 3, 4, 9):
        res.set_groups(g)
        assert res.groups == (2 if g == 0 else min(g, 4))
        out = res.solve()
        for b, one in zip(out["results"], ones):
            assert np.array_equal(b["poses"], one["poses"]) and np.array_equal(b["points"], one["points"])
            assert np.array_equal(b["chi2"], one["chi2"]) and np.array_equal(b["trials"], one["trials"])
            assert np.array_equal(b["edge_chi2"], one["edge_chi2"]) and b["rounds"] == one["rounds"]
    res.close()
    host = ba.BaBatch(ctx, probs).solve()                     # the one-call form takes the default grouping
    assert all(np.array_equal(b["poses"], one["poses"]) for b, one in zip(host["results"], ones))
    assert ba.BaBatch(ctx, probs[:3], resident=True).groups == 1


def test_trial_finish_litmus(ctx):
    """Litmus test of the ticket protocol that lets the LAST chunk of k_backsub_residual complete an LM trial (agent-scope stores of
    the chunk's three sums, s_waitcnt vmcnt(0), a relaxed ticket, agent-scope loads; no fence -- ba.hip, k_backsub_residual_body).
    A chunk's sums that were not visible to the last chunk would change chi2 / the LM decision of that trial.  Reference = the same
    solves with the trial completed by a launch of k_reduce_trial (ssx_debug_set_trial_finish(0): ordered by the kernel boundary).
    128 windows of 20 .. 80 chunks x 10 trials x REPS solves = REPS x 1280 protocol runs per round, beside a second context that
    keeps the chip busy half of the time (chunks of one window then finish far apart).  SSX_LITMUS_REPS raises the count; the
    default gives >= 10 000 launches-x-windows in a few seconds.  tools/jobs/litmus_O1.sh runs it on a library built at -O1."""
    import os
    import threading
    import ssvio_amd
    lib = ssvio_amd.load()
    lib.ssx_debug_set_trial_finish.argtypes = [C.c_int32]; lib.ssx_debug_set_trial_finish.restype = None
    reps = int(os.environ.get("SSX_LITMUS_REPS", "8"))
    probs = [make_ba_problem(P=6 + (k % 5), L=1000 + 97 * (k % 31), obs_per_lm=4 + (k % 2), seed=4000 + k, frac_gross=0.3 if k % 9 == 0 else 0.03)
             for k in range(128)]
    batch = ba.BaBatch(ctx, probs, resident=True, with_edge_errors=True)
    try:
        lib.ssx_debug_set_trial_finish(0)
        ref = [{k: np.copy(v) for k, v in r.items() if isinstance(v, np.ndarray)} for r in batch.solve()["results"]]   # (solve() reuses its arrays)
        lib.ssx_debug_set_trial_finish(1)
        ctx2 = ssvio_amd.Context(0)
        load = ba.BaBatch(ctx2, probs[:64], resident=True, with_edge_errors=False)
        stop = threading.Event()

        def keep_busy():
            while not stop.is_set():
                load.solve(download=False)

        n_runs = 0
        for rep in range(reps):
            th = None
            if rep % 2:
                stop.clear()
                th = threading.Thread(target=keep_busy)
                th.start()
            try:
                got = batch.solve()["results"]
            finally:
                if th:
                    stop.set()
                    th.join()
            for w, (a, b) in enumerate(zip(got, ref)):
                assert np.array_equal(a["trials"], b["trials"]) and np.array_equal(a["chi2"], b["chi2"]) and np.array_equal(a["lam"], b["lam"]), (rep, w)
                assert np.array_equal(a["poses"], b["poses"]) and np.array_equal(a["points"], b["points"]) and np.array_equal(a["edge_chi2"], b["edge_chi2"]), (rep, w)
                n_runs += int(np.sum(a["trials"]))
        assert n_runs >= 1200 * reps
        load.close()
        ctx2.close()
    finally:
        lib.ssx_debug_set_trial_finish(1)
        batch.close()


def test_sparse_slabs_equal_dense_slabs(ctx):
    """A chunk's workgroup writes only the blocks of the reduced system its landmarks contribute to and the pose blocks of the
    poses it holds edges of (BaDev::touch), and the reductions read only those; ssx_debug_set_dense_slabs(1) brings back round 3's
    dense slabs (every entry written, zeros included, every entry read).  The sums are formed in the same order either way: per
    window the same bits, single call and resident batch, every batch grouping -- windows of 1 .. 80 chunks, 4 .. 16 keyframes,
    fixed poses, rejected trials, several outer rounds, numeric Jacobians, landmarks that arrive sorted by keyframe or shuffled."""
    import ssvio_amd
    lib = ssvio_amd.load()
    lib.ssx_debug_set_dense_slabs.argtypes = [C.c_int32]; lib.ssx_debug_set_dense_slabs.restype = None
    probs = [make_ba_problem(P=10, L=4000, seed=601), make_ba_problem(P=10, L=380, seed=602), make_ba_problem(P=16, L=1700, obs_per_lm=5, seed=603),
             make_ba_problem(P=12, L=1500, obs_per_lm=4, seed=604), make_ba_problem(P=4, L=60, obs_per_lm=4, seed=605),
             make_ba_problem(P=10, L=2000, seed=606, pose_t_noise=0.3, pose_r_noise=0.03), make_ba_problem(P=10, L=600, seed=607, frac_gross=0.45),
             make_ba_problem(P=7, L=800, obs_per_lm=3, seed=608, fix_first_pose=True), make_ba_problem(P=10, L=4000, seed=609)]
    keys = ("poses", "points", "chi2", "lam", "trials", "edge_chi2", "edge_outlier")
    try:
        for jac in (ba.JAC_ANALYTIC, ba.JAC_NUMERIC_G2O):
            lib.ssx_debug_set_dense_slabs(1)
            dense = [ba.ba_solve(ctx, pr, jac_mode=jac) for pr in probs]
            lib.ssx_debug_set_dense_slabs(0)
            ones = [ba.ba_solve(ctx, pr, jac_mode=jac) for pr in probs]
            assert max(o["rounds"] for o in ones) >= 2
            for a, b in zip(dense, ones):
                for k in keys:
                    assert np.array_equal(a[k], b[k]), ("dense vs sparse", jac, k)
            for mode in (0, 1):
                lib.ssx_debug_set_dense_slabs(mode)
                res = ba.BaBatch(ctx, probs, resident=True, jac_mode=jac)
                for groups in (1, 2, 3):
                    res.set_groups(groups)
                    out = res.solve()
                    for b, one in zip(out["results"], ones):
                        for k in keys:
                            assert np.array_equal(b[k], one[k]), (mode, groups, k)
                        assert b["rounds"] == one["rounds"]
                res.close()
    finally:
        lib.ssx_debug_set_dense_slabs(-1)


_LISTS_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
import ssvio_amd
from ssvio_amd import ba
from tools.synth import make_ba_problem
ctx = ssvio_amd.Context(0)
out = {}
def with_duplicates(pr, n):   # right-camera observations of the first n landmarks from the pose of their first edge
    ep, pt, uv = [], [], []
    for j in range(n):
        e = int(np.nonzero(pr["edge_point"] == j)[0][0])
        u = pr["edge_uv"][e].copy(); u[0] -= pr["K"][0] * 0.537 / 20.0
        ep.append(pr["edge_pose"][e]); pt.append(j); uv.append(u)
    pr["edge_pose"] = np.concatenate([pr["edge_pose"], np.array(ep, dtype=np.int32)])
    pr["edge_point"] = np.concatenate([pr["edge_point"], np.array(pt, dtype=np.int32)])
    pr["edge_uv"] = np.concatenate([pr["edge_uv"], np.array(uv)])
    pr["edge_cam"] = np.concatenate([pr["edge_cam"], np.ones(n, dtype=np.uint8)])
    pr["E"] = len(pr["edge_pose"])
    return pr
for i, kw in enumerate(eval(sys.argv[3])):
    pr = make_ba_problem(**kw)
    if i == 5: pr = with_duplicates(pr, 80)
    if i in (1, 5, 6, 7, 8):                # the caller's edge order is arbitrary: shuffled observations
        q = np.random.default_rng(7 + i).permutation(pr["E"])
        for k in ("edge_pose", "edge_point", "edge_uv", "edge_cam"): pr[k] = np.ascontiguousarray(pr[k][q])
    r = ba.ba_solve(ctx, pr)
    for k in ("poses", "points", "chi2", "lam", "trials", "edge_chi2"):
        out[f"{i}_{k}"] = np.asarray(r[k])
probs = [make_ba_problem(**kw) for kw in eval(sys.argv[3])[:4]] * 3     # (small windows only: a batch)
rb = ba.BaBatch(ctx, probs, resident=True).solve()
for i, r in enumerate(rb["results"]):
    out[f"b{i}_poses"] = r["poses"]; out[f"b{i}_chi2"] = r["chi2"]
np.savez(sys.argv[2], **out)
"""


def test_device_built_lists_equal_host_built_lists(ctx, tmp_path):
    """Small windows are marshalled on the device: the edge sort by (landmark, pose), packed records, pose-major order
    (k_prep_scatter / k_prep_chunk) and the pair lists + work items (k_build_lists) from the caller's raw arrays.
    SSX_BA_HOST_PREP=1 keeps the sort and the records on the host, SSX_BA_HOST_LISTS=1 everything; and the two slab reductions
    of an LM slot run as one launch (k_reduce_both: the Schur reduction sums its share of Hpp / bp itself) or, with
    SSX_BA_SPLIT_REDUCE=1, as two; a single solve's results leave in one kernel-written pinned block (k_pack_one) or, with SSX_BA_NO_PACK=1,
    by the three copies of before.  All five must give the same bits: single solves (duplicate observations, fixed
    poses and landmarks, 4 .. 16 keyframes, sparse co-visibility) and a resident batch in two groups."""
    import subprocess, sys as _sys
    cases = [dict(P=10, L=700, seed=41), dict(P=12, L=500, obs_per_lm=4, seed=42), dict(P=16, L=600, obs_per_lm=5, seed=43, fix_first_pose=True),
             dict(P=4, L=60, obs_per_lm=4, seed=44), dict(P=7, L=300, obs_per_lm=2, seed=45), dict(P=10, L=500, seed=46, fix_first_pose=True, frac_fixed=0.3),
             dict(P=10, L=4000, seed=47),
             # windows beyond 16 keyframes (the large path: records + pose-major edge list built by big_records on the device)
             dict(P=40, L=1500, obs_per_lm=5, seed=48, fix_first_pose=True, loop=True),
             dict(P=48, L=1200, obs_per_lm=4, seed=49, fix_first_pose=True, shuffle_poses=True)]
    import inspect
    accepted = inspect.signature(make_ba_problem).parameters
    cases = [{k: v for k, v in kw.items() if k in accepted} for kw in cases]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for mode in ("device", "host", "hostprep", "splitreduce", "nopack"):
        env = dict(os.environ)
        env.pop("SSX_BA_HOST_LISTS", None)
        env.pop("SSX_BA_HOST_PREP", None)
        env.pop("SSX_BA_SPLIT_REDUCE", None)
        env.pop("SSX_BA_NO_PACK", None)
        if mode == "nopack":
            env["SSX_BA_NO_PACK"] = "1"                 # a single solve's control block / statistics / estimate / chi2 by three copies instead of k_pack_one's block
        if mode == "splitreduce":
            env["SSX_BA_SPLIT_REDUCE"] = "1"            # k_reduce_lin and k_reduce_schur as two launches per LM slot instead of k_reduce_both
        if mode == "host":
            env["SSX_BA_HOST_LISTS"] = "1"              # everything on the host: edge sort, records, pair lists, work items
        if mode == "hostprep":
            env["SSX_BA_HOST_PREP"] = "1"               # edge sort + records on the host, lists on the device (round 2's split)
        path = str(tmp_path / f"{mode}.npz")
        r = subprocess.run([_sys.executable, "-c", _LISTS_SCRIPT, root, path, repr(cases)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[mode] = np.load(path)
    assert sorted(outs["device"].files) == sorted(outs["host"].files) and len(outs["device"].files) > 40
    for k in outs["device"].files:
        assert np.array_equal(outs["device"][k], outs["host"][k]), k
        assert np.array_equal(outs["device"][k], outs["hostprep"][k]), k
        assert np.array_equal(outs["device"][k], outs["splitreduce"][k]), k
        assert np.array_equal(outs["device"][k], outs["nopack"][k]), k


def test_poses_only_download_equals_the_full_download(ctx):
    """BaBatch.solve(points=False, want_edges=False): only the keyframe poses cross PCIe -- requested behind the LM control words of
    every outer round, packed from the state buffer the window's own control block names (batch_run's speculative download).  Same
    poses, iteration counts and LM histories as the full download: windows of 4 .. 12 keyframes, one that needs several outer
    rounds (the speculative copies of the earlier rounds are overwritten), rejected trials, both batch groupings, solved twice."""
    probs = [make_ba_problem(P=10, L=800, seed=801), make_ba_problem(P=7, L=300, obs_per_lm=4, seed=802), make_ba_problem(P=12, L=900, obs_per_lm=4, seed=803),
             make_ba_problem(P=10, L=600, seed=804, frac_gross=0.45), make_ba_problem(P=4, L=60, obs_per_lm=4, seed=805),
             make_ba_problem(P=10, L=2000, seed=806, pose_t_noise=0.3, pose_r_noise=0.03)]
    full = ba.BaBatch(ctx, probs, resident=True)
    ref = full.solve()["results"]
    full.close()
    assert max(o["rounds"] for o in ref) >= 2 and max(int(o["trials"].max()) for o in ref) >= 2
    lean = ba.BaBatch(ctx, probs, resident=True, with_edge_errors=False)
    for groups in (1, 2):
        lean.set_groups(groups)
        for _ in range(2):
            lean.solve(want_edges=False, summaries=False, points=False)
            for i, one in enumerate(ref):
                assert np.array_equal(lean.poses[i][:probs[i]["P"]], one["poses"]), (groups, i)
                assert lean.res[i].n_iters == one["n_iters"] and lean.res[i].rounds == one["rounds"]
                k = min(one["n_iters"], len(one["chi2"]))
                assert np.array_equal(np.array(lean.res[i].iter_chi2[:k]), one["chi2"][:k])
    lean.close()


@pytest.mark.parametrize("knob", ["iters", "outer_rounds"])
def test_poses_only_download_without_any_lm_round_returns_the_input(ctx, knob):
    """No LM round runs (iters = 0 or outer_rounds = 0): batch_run's speculative poses download is never enqueued, and the call
    must fall back to the ordinary gather -- the INPUT poses come back, not what the staging block held after an earlier solve."""
    probs = [make_ba_problem(P=10, L=500, seed=811), make_ba_problem(P=6, L=200, obs_per_lm=4, seed=812)]
    warm = ba.BaBatch(ctx, probs, resident=True, with_edge_errors=False)
    warm.solve(want_edges=False, summaries=False, points=False)          # leaves optimised poses in the ctx's staging memory
    assert not np.array_equal(warm.poses[0], probs[0]["poses"])
    warm.close()
    kw = dict(iters=0) if knob == "iters" else dict(outer_rounds=0)
    idle = ba.BaBatch(ctx, probs, resident=True, with_edge_errors=False, **kw)
    one = ba.BaBatch(ctx, probs, **kw)
    for b in (idle, one):
        b.solve(want_edges=False, summaries=False, points=False)
        for i, pr in enumerate(probs):
            assert np.array_equal(b.poses[i], pr["poses"]), (knob, i)
            assert b.res[i].n_iters == 0
    idle.close()


def test_bench_size_batch_equals_single_calls(ctx):
    """The bench's own configuration (BASELINE configs[2]: 10 keyframes x 4000 landmarks x 20 000 edges per window), eight
    windows resident in two groups: every window returns the bits of ssx_ba_solve, and solving again from the uploaded
    state returns them again."""
    probs = [make_ba_problem(P=10, L=4000, seed=500 + k) for k in range(8)]
    res = ba.BaBatch(ctx, probs, resident=True)
    assert res.groups == 2
    first = res.solve()
    again = res.solve()
    for pr, a, b in zip(probs, first["results"], again["results"]):
        one = ba.ba_solve(ctx, pr)
        assert one["n_iters"] == 10 and a["n_iters"] == 10
        for k in ("poses", "points", "chi2", "lam", "trials", "edge_chi2", "edge_outlier"):
            assert np.array_equal(a[k], one[k]), k
            assert np.array_equal(b[k], one[k]), k
        assert a["chi2"][-1] < a["chi2"][0]
    res.close()


def test_batched_windows_equal_single_calls(ctx):
    """ssx_ba_solve_batch: many small windows in one call (one grid dimension = the window) give, per window, exactly
    the bits of ssx_ba_solve -- different sizes, a window that needs several outer rounds, rejected LM trials, a fixed
    pose, numeric Jacobians."""
    probs = [make_ba_problem(P=10, L=800, seed=21), make_ba_problem(P=7, L=300, obs_per_lm=4, seed=22),
             make_ba_problem(P=12, L=900, obs_per_lm=4, seed=23),                       # 72 unknowns: the k_solve path
             make_ba_problem(P=10, L=600, seed=24, frac_gross=0.45),                    # inlier ratio < 0.7: several rounds
             make_ba_problem(P=6, L=200, obs_per_lm=3, seed=25, fix_first_pose=True),
             make_ba_problem(P=10, L=2000, seed=26, pose_t_noise=0.3, pose_r_noise=0.03),   # far start: rejected trials
             make_ba_problem(P=4, L=60, obs_per_lm=4, seed=27)]
    for jac in (ba.JAC_ANALYTIC, ba.JAC_NUMERIC_G2O):
        batch = ba.BaBatch(ctx, probs, jac_mode=jac)
        out = batch.solve()
        assert max(o["rounds"] for o in out["results"]) >= 2 and max(o["trials"].max() for o in out["results"]) >= 2
        for pr, b in zip(probs, out["results"]):
            one = ba.ba_solve(ctx, pr, jac_mode=jac)
            assert b["rounds"] == one["rounds"] and b["n_iters"] == one["n_iters"]
            assert np.array_equal(b["trials"], one["trials"]) and np.array_equal(b["chi2"], one["chi2"]) and np.array_equal(b["lam"], one["lam"])
            assert np.array_equal(b["poses"], one["poses"]) and np.array_equal(b["points"], one["points"])
            assert np.array_equal(b["edge_chi2"], one["edge_chi2"]) and np.array_equal(b["edge_outlier"], one["edge_outlier"])
            assert (b["n_inliers"], b["n_outliers"]) == (one["n_inliers"], one["n_outliers"])
    # a RESIDENT batch (uploaded once): solved twice from the uploaded state, same bits both times; solve without download
    res = ba.BaBatch(ctx, probs, resident=True)
    r1 = res.solve(); p1 = [o["poses"].copy() for o in r1["results"]]; c1 = [o["edge_chi2"].copy() for o in r1["results"]]
    nd = res.solve(download=False)
    assert nd["results"] is None and nd["n_iters_total"] == r1["n_iters_total"]
    r2 = res.solve()
    for pr, a, pa, ca in zip(probs, r2["results"], p1, c1):
        one = ba.ba_solve(ctx, pr)
        assert np.array_equal(a["poses"], pa) and np.array_equal(a["poses"], one["poses"]) and np.array_equal(a["edge_chi2"], ca)
        assert np.array_equal(a["edge_chi2"], one["edge_chi2"]) and np.array_equal(a["trials"], one["trials"])
    res.close()
    # without per-edge outputs, and a batch that contains a large window (falls back to one call per window)
    again = ba.BaBatch(ctx, probs[:3]).solve(want_edges=False)
    assert all(np.array_equal(a["poses"], ba.ba_solve(ctx, p)["poses"]) for a, p in zip(again["results"], probs[:3]))
    # (nobody asks for per-edge results: the kernels do not write the per-edge error arrays -- nothing else may change)
    for p in probs[:2]:
        w, wo = ba.ba_solve(ctx, p), ba.ba_solve(ctx, p, want_edges=False)
        for k in ("poses", "points", "chi2", "lam", "trials"):
            assert np.array_equal(w[k], wo[k]), k
        assert (w["n_inliers"], w["n_outliers"], w["rounds"]) == (wo["n_inliers"], wo["n_outliers"], wo["rounds"])
    lean = ba.BaBatch(ctx, probs[:3], resident=True, with_edge_errors=False)
    lr = lean.solve()
    assert all(np.array_equal(a["poses"], ba.ba_solve(ctx, p)["poses"]) and a["n_outliers"] == ba.ba_solve(ctx, p)["n_outliers"]
               for a, p in zip(lr["results"], probs[:3]))
    lean.close()
    mixed = [probs[0], make_ba_problem(P=30, L=1200, obs_per_lm=5, seed=28, fix_first_pose=True)]
    mo = ba.BaBatch(ctx, mixed, outer_rounds=1, iters=4).solve()
    for pr, b in zip(mixed, mo["results"]):
        assert np.array_equal(b["poses"], ba.ba_solve(ctx, pr, outer_rounds=1, iters=4)["poses"])


def _window_feed(pr):
    """per keyframe of a synthetic problem: (pose, ids / xyz / fixed of the landmarks first seen there, its observations)"""
    first = np.full(pr["L"], 10 ** 9, dtype=np.int64)
    np.minimum.at(first, pr["edge_point"], pr["edge_pose"])
    feed = []
    for k in range(pr["P"]):
        new = np.nonzero(first == k)[0]
        e = np.nonzero(pr["edge_pose"] == k)[0]
        feed.append(dict(pose=pr["poses"][k], new_ids=1000 + new, new_xyz=pr["points"][new], new_fixed=pr["point_fixed"][new],
                         obs_lm=1000 + pr["edge_point"][e], obs_uv=pr["edge_uv"][e], obs_cam=pr["edge_cam"][e]))
    return feed


def _assert_same(a, b, what):
    for k in ("poses", "points", "chi2", "lam", "trials", "edge_chi2", "edge_outlier"):
        assert np.array_equal(a[k], b[k]), (what, k)
    assert a["rounds"] == b["rounds"] and a["n_iters"] == b["n_iters"], what


def _upload_format(ctx, pr):
    keep = []
    st = ba._problem_struct(pr, keep)
    ctx.lib.ssx_ba_debug_upload_format.restype = C.c_int32
    ctx.lib.ssx_ba_debug_upload_format.argtypes = [C.POINTER(type(st))]
    return int(ctx.lib.ssx_ba_debug_upload_format(C.byref(st)))


def test_compact_upload_is_lossless(ctx):
    """Observation arrays whose pixel coordinates are float values (as the reference's cv::KeyPoint measurements are) cross
    PCIe as bytes / 16-bit words / floats (13 instead of 26 bytes per observation).  The window object keeps WIDE device
    arrays (int / double) of the same observations: a fresh solve of what it exports (compact upload) and its own solve
    (wide arrays) must agree bit for bit -- alone, in a batch next to a window whose coordinates are not floats, and with
    right-camera observations."""
    pr = make_ba_problem(P=10, L=1500, obs_per_lm=5, seed=91, pose_t_noise=0.05, uv_f32=True)
    pr["edge_cam"] = (np.arange(len(pr["edge_pose"])) % 3 == 0).astype(np.uint8)
    pw = make_ba_problem(P=10, L=1500, obs_per_lm=5, seed=92, pose_t_noise=0.05)
    assert _upload_format(ctx, pr) == 7 and _upload_format(ctx, pw) == 3
    win = ba.BaWindow(ctx, pr["K"], pr["cam_ext"])
    for k, f in enumerate(_window_feed(pr)):
        win.push(k, **f)
    ex = win.export()
    assert _upload_format(ctx, ex) == 7
    fresh = ba.ba_solve(ctx, ex)
    both = ba.BaBatch(ctx, [ex, pw]).solve()["results"]
    alone = ba.ba_solve(ctx, pw)
    got = win.solve()
    _assert_same(got, fresh, "compact vs wide")
    for k in ("poses", "points", "edge_chi2", "edge_outlier"):
        assert np.array_equal(both[0][k], fresh[k]), k
        assert np.array_equal(both[1][k], alone[k]), k
    win.close()
    # one coordinate that is not a float's value: the window goes up wide, and nothing else changes
    pr2 = dict(pr); pr2["edge_uv"] = pr["edge_uv"].copy(); pr2["edge_uv"][17, 1] += 1e-9
    assert _upload_format(ctx, pr2) == 3
    r2 = ba.ba_solve(ctx, pr2)
    assert r2["n_iters"] == ba.ba_solve(ctx, pr)["n_iters"] and np.isfinite(r2["chi2"][-1])


def test_window_update_batch_equals_single_edits(ctx):
    """ssx_ba_window_update_batch (one keyframe replaced in each of n windows, on the library's host threads) leaves every window
    exactly as its own pop + push calls do -- by landmark id and by landmark slot -- and reports a failing window without
    touching the others."""
    from ssvio_amd._lib import SsxError
    prs = [make_ba_problem(P=14, L=1800, obs_per_lm=5, seed=300 + q, pose_t_noise=0.05) for q in range(3)]
    feeds = [_window_feed(p) for p in prs]
    n = 6
    ref = [ba.BaWindow(ctx, prs[i % 3]["K"], prs[i % 3]["cam_ext"]) for i in range(n)]
    got = [ba.BaWindow(ctx, prs[i % 3]["K"], prs[i % 3]["cam_ext"]) for i in range(n)]
    slot_of = [dict() for _ in range(n)]
    for k in range(14):
        ups = []
        for i in range(n):
            f = feeds[i % 3][k]
            if k >= 10:
                ref[i].pop(100 + k - 10)
            ref[i].push(100 + k, **f)
            u = dict(pop=100 + k - 10 if k >= 10 else None, push=100 + k, pose=f["pose"], new_ids=f["new_ids"], new_xyz=f["new_xyz"],
                     new_fixed=f["new_fixed"], obs_uv=f["obs_uv"], obs_cam=f["obs_cam"])
            if i % 2 == 0:
                u["obs_lm"] = f["obs_lm"]
            else:
                new_index = {int(x): j for j, x in enumerate(f["new_ids"])}
                u["obs_slot"] = np.array([slot_of[i][int(x)] if int(x) not in new_index else -1 - new_index[int(x)] for x in f["obs_lm"]], dtype=np.int32)
            ups.append(u)
        slots = ba.BaWindow.update_batch(got, ups)
        for i in range(n):
            if i % 2 == 1:
                slot_of[i].update({int(x): int(s_) for x, s_ in zip(feeds[i % 3][k]["new_ids"], slots[i])})
    for i in range(n):
        ea, eb = ref[i].export(), got[i].export()
        for key in ea:
            assert np.array_equal(ea[key], eb[key]), (i, key)
    ra = ba.BaWindow.solve_batch(ref); rb = ba.BaWindow.solve_batch(got)
    for i in range(n):
        _assert_same(ra[i], rb[i], ("window", i))
    # a failing window (its keyframe is already there) is reported, the other one is updated; a window listed twice is refused
    f = feeds[0][13]
    bad = dict(pop=None, push=100 + 13, pose=f["pose"], new_ids=[], new_xyz=np.zeros((0, 3)), obs_lm=[], obs_uv=np.zeros((0, 2)))
    ok = dict(pop=100 + 4, push=None)
    with pytest.raises(SsxError):
        ba.BaWindow.update_batch([got[0], got[1]], [bad, ok])
    assert got[1].size()[0] == 9 and got[0].size()[0] == 10
    with pytest.raises(SsxError):
        ba.BaWindow.update_batch([got[2], got[2]], [dict(pop=100 + 4, push=None), dict(pop=100 + 5, push=None)])
    assert got[2].size()[0] == 10
    for w in ref + got:
        w.close()


def test_resident_window_equals_fresh_solves(ctx):
    """ssx_ba_window: keyframes are pushed (pose + the landmarks they introduce + their observations), popped (their
    observations and the landmarks nobody sees any more go with them) and the window is optimised where it lies; after
    every change the solve returns, bit for bit, what ssx_ba_solve returns for the problem ssx_ba_window_export lists --
    a sliding 10-keyframe window over a 16-keyframe trajectory (keyframe and landmark slots reused, dead observation
    blocks, the storage rewritten), a fixed landmark set, an overwritten pose, numeric Jacobians, and three windows solved in
    one call."""
    pr = make_ba_problem(P=16, L=2400, obs_per_lm=5, seed=71, pose_t_noise=0.05)
    feed = _window_feed(pr)
    for jac in (ba.JAC_ANALYTIC, ba.JAC_NUMERIC_G2O):
        win = ba.BaWindow(ctx, pr["K"], pr["cam_ext"], jac_mode=jac)
        for k in range(10):
            win.push(100 + k, **feed[k])
        assert win.size()[0] == 10
        steps = 0
        for k in range(10, 17):
            ex = win.export()
            fresh = ba.ba_solve(ctx, ex, jac_mode=jac)
            got = win.solve()
            _assert_same(got, fresh, ("step", k, jac))
            assert got["chi2"][-1] <= got["chi2"][0]
            assert np.array_equal(win.export()["poses"], got["poses"])          # the state is the result
            steps += 1
            if k == 16:
                break
            win.pop(100 + k - 10)                                            # Map::RemoveOldActiveKeyframe
            win.push(100 + k, **feed[k])
            if k == 12:
                win.set_landmark(int(win.export()["lm_ids"][5]), fixed=1)
                win.set_pose(100 + k, feed[k]["pose"] + np.array([0, 0, 0, 0, 0.01, 0, 0.02]))
        assert steps == 7 and win.size()[0] == 10
        win.close()
    # the slot form of push (the caller keeps the window's landmark slots): same window, same bits
    a = ba.BaWindow(ctx, pr["K"], pr["cam_ext"]); b = ba.BaWindow(ctx, pr["K"], pr["cam_ext"])
    slot_of = {}
    for k in range(13):
        if k >= 10:
            a.pop(100 + k - 10); b.pop(100 + k - 10)
        f = feed[k]
        a.push(100 + k, **f)
        new_index = {int(i): j for j, i in enumerate(f["new_ids"])}
        obs_slot = np.array([slot_of[int(i)] if int(i) not in new_index else -1 - new_index[int(i)] for i in f["obs_lm"]], dtype=np.int32)
        slots = b.push_slots(100 + k, f["pose"], new_ids=f["new_ids"], new_xyz=f["new_xyz"], new_fixed=f["new_fixed"], obs_slot=obs_slot,
                             obs_uv=f["obs_uv"], obs_cam=f["obs_cam"])
        slot_of.update({int(i): int(s_) for i, s_ in zip(f["new_ids"], slots)})
    ea, eb = a.export(), b.export()
    for key in ea:
        assert np.array_equal(ea[key], eb[key]), key
    _assert_same(a.solve(), b.solve(), "slots")
    a.close(); b.close()
    # several windows of one context in one call
    prs = [make_ba_problem(P=12, L=900, obs_per_lm=4, seed=81), make_ba_problem(P=10, L=2000, seed=82), make_ba_problem(P=6, L=300, obs_per_lm=3, seed=83)]
    wins = []
    for q in prs:
        w = ba.BaWindow(ctx, q["K"], q["cam_ext"])
        for k, f in enumerate(_window_feed(q)):
            w.push(k, **f)
        w.pop(0)
        wins.append(w)
    fresh = [ba.ba_solve(ctx, w.export()) for w in wins]
    got = ba.BaWindow.solve_batch(wins)
    for a, b in zip(got, fresh):
        _assert_same(a, b, "batch")
    again = [ba.ba_solve(ctx, w.export()) for w in wins]                     # and from the state the batch left
    got2 = ba.BaWindow.solve_batch(wins)
    for a, b in zip(got2, again):
        _assert_same(a, b, "batch, second solve")
    for w in wins:
        w.close()


def test_window_batches_do_not_depend_on_stream_groups_or_device_turns(ctx):
    """ssx_ba_set_batch_groups (streams a context's batched window solve is spread over) and ssx_ba_device_turns (the device phases of
    different contexts alternate) are orchestration: twelve windows solved in one call return, per window, the bits of a fresh
    ssx_ba_solve of the exported problem, whatever the grouping, with turns on and off, and from two contexts driven by two threads."""
    import ctypes as C
    import threading
    import ssvio_amd
    lib = ctx.lib
    lib.ssx_ba_device_turns.restype = None; lib.ssx_ba_device_turns.argtypes = [C.c_int32]
    lib.ssx_ba_set_batch_groups.restype = C.c_int32; lib.ssx_ba_set_batch_groups.argtypes = [C.c_void_p, C.c_int32]
    prs = [make_ba_problem(P=6 + (k % 6), L=200 + 150 * k, obs_per_lm=3 + (k % 3), seed=500 + k) for k in range(12)]

    def build(c):
        wins = []
        for q in prs:
            w = ba.BaWindow(c, q["K"], q["cam_ext"])
            for k, f in enumerate(_window_feed(q)):
                w.push(k, **f)
            wins.append(w)
        return wins
    wins = build(ctx)
    fresh = [ba.ba_solve(ctx, w.export()) for w in wins]
    for w in wins:
        w.close()
    try:
        for groups, turns in ((0, 0), (1, 0), (2, 1), (3, 0), (4, 1)):
            ctx.check(lib.ssx_ba_set_batch_groups(ctx.handle, groups))
            lib.ssx_ba_device_turns(turns)
            wins = build(ctx)
            for a, b in zip(ba.BaWindow.solve_batch(wins), fresh):
                _assert_same(a, b, f"groups {groups} turns {turns}")
            for w in wins:
                w.close()
        # two contexts, two host threads, turns on: the same bits from both
        lib.ssx_ba_device_turns(1)
        other = ssvio_amd.Context(0)
        out = {}

        def run(name, c):
            ws = build(c)
            out[name] = [ba.BaWindow.solve_batch(ws) for _ in range(1)][0]
            for w in ws:
                w.close()
        th = [threading.Thread(target=run, args=("a", ctx)), threading.Thread(target=run, args=("b", other))]
        for t_ in th:
            t_.start()
        for t_ in th:
            t_.join()
        for name in ("a", "b"):
            for a, b in zip(out[name], fresh):
                _assert_same(a, b, f"context {name}, turns on")
        other.close()
    finally:
        lib.ssx_ba_device_turns(0)
        lib.ssx_ba_set_batch_groups(ctx.handle, 0)


def test_resident_window_matches_reference_golden(ctx):
    """The reference's own window (12 keyframes, config/kitti_00.yaml:30) built by PUSHING its keyframes into an ssx_ba_window
    and solved in place, against the vectors of the compiled reference (tests/golden/ref_golden.npz): rounds, LM iterations,
    trial counts, chi2 trajectory, final poses, per-edge residuals -- the incremental path against g2o, no oracle in between."""
    import os
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_golden.npz"))
    for name in ("win12", "C3"):
        pr = _golden_problem(G, name)
        for jac in (ba.JAC_NUMERIC_G2O, ba.JAC_ANALYTIC):
            win = ba.BaWindow(ctx, pr["K"], pr["cam_ext"], jac_mode=jac)
            for k, f in enumerate(_window_feed(pr)):
                win.push(k, **f)
            ex = win.export()
            g = win.solve()
            win.close()
            assert g["rounds"] == int(G[f"ba_{name}_rounds"]) and g["n_iters"] == len(G[f"ba_{name}_chi2"])
            np.testing.assert_array_equal(g["trials"], G[f"ba_{name}_trials"])
            np.testing.assert_allclose(g["chi2"], G[f"ba_{name}_chi2"], rtol=2e-5)
            assert np.abs(g["poses"] - G[f"ba_{name}_poses"]).max() < 5e-6          # keyframes were pushed in pose order: slot = pose index
            # the window lists observations keyframe by keyframe, the golden vectors landmark by landmark: match them up
            key_w = ex["edge_pose"].astype(np.int64) * 10 ** 7 + (ex["lm_ids"][ex["edge_point"]] - 1000)
            key_g = pr["edge_pose"].astype(np.int64) * 10 ** 7 + pr["edge_point"]
            order = np.argsort(key_w)[np.argsort(np.argsort(key_g))]
            assert np.array_equal(key_w[order], key_g)
            ec = g["edge_chi2"][order][G[f"ba_{name}_edge_sel"]]
            d = np.abs(np.sqrt(ec) - np.sqrt(G[f"ba_{name}_edge_chi2"]))
            frac, p99 = float((d <= RESID_TOL).mean()), float(np.percentile(d, 99))
            print(f"[window {name} jac={jac}] |r_gpu - r_ref| px: median {np.median(d):.2e} p99 {p99:.2e} max {d.max():.2e} <=1e-4: {100 * frac:.2f} %")
            assert frac >= 0.99 and p99 <= 1.5e-4
            if jac == ba.JAC_ANALYTIC:
                assert d.max() < RESID_TOL


def test_window_storage_is_rewritten_when_mostly_dead(ctx):
    """A window that slides for a long time: the observation blocks of popped keyframes stay as dead entries until more than
    half of the storage (and more than 1024 entries) is dead, then the storage is rewritten and re-sent.  Before and after the
    rewrite the solve equals a fresh ssx_ba_solve of the exported problem; slots of keyframes and landmarks are reused
    throughout."""
    pr = make_ba_problem(P=30, L=9000, obs_per_lm=5, seed=91, pose_t_noise=0.03)
    feed = _window_feed(pr)
    win = ba.BaWindow(ctx, pr["K"], pr["cam_ext"], iters=4)
    for k in range(8):
        win.push(k, **feed[k])
    sizes = []
    for k in range(8, 30):
        win.pop(k - 8)
        win.push(k, **feed[k])
        ex = win.export()
        assert ex["P"] == 8 and ex["E"] == len(ex["edge_pose"]) and ex["edge_point"].max() < ex["L"]
        if k % 3 == 0 or k >= 27:
            _assert_same(win.solve(), ba.ba_solve(ctx, ex, iters=4), ("slide", k))
        sizes.append(win.size())
    # every keyframe carries ~1700 observations: 22 pops leave > 30 000 dead entries behind unless the storage was rewritten
    assert sizes[-1][0] == 8 and sizes[-1][2] < 16000
    win.close()


def test_window_misuse(ctx):
    pr = make_ba_problem(P=6, L=90, obs_per_lm=3, seed=2)
    feed = _window_feed(pr)
    from ssvio_amd._lib import SsxError
    win = ba.BaWindow(ctx, pr["K"], pr["cam_ext"])
    win.push(1, **feed[0])
    with pytest.raises(SsxError):
        win.push(1, **feed[1])                                               # the keyframe is already there
    with pytest.raises(SsxError):
        win.push(2, feed[1]["pose"], obs_lm=[999999], obs_uv=[[1.0, 2.0]])   # an observation of an unknown landmark
    with pytest.raises(SsxError):
        win.pop(77)
    assert win.size() == (1, len(feed[0]["new_ids"]), len(feed[0]["obs_lm"]))   # nothing was changed by the failures
    # the removal calls: a wrong flag count and an unknown keyframe are refused; unknown landmarks / pairs are skipped
    n_obs = win.size()[2]
    with pytest.raises(SsxError):
        win.remove_flagged(np.zeros(n_obs + 1, np.uint8))
    with pytest.raises(SsxError):
        win.remove_observations(77, [int(feed[0]["new_ids"][0])])
    assert win.remove_landmarks([10 ** 12]) == 0 and win.remove_observations(1, [10 ** 12]) == 0
    assert win.size() == (1, len(feed[0]["new_ids"]), n_obs)
    with pytest.raises(SsxError):
        ctx.check(ctx.lib.ssx_ba_window_set_fix_rule(win.handle, 7))
    # every observation removed: the landmarks go with them, the keyframe stays, and a solve of the empty graph returns the state
    assert win.remove_flagged(np.ones(n_obs, np.uint8)) == n_obs
    assert win.size() == (1, 0, 0)
    r = win.solve()
    assert r["n_iters"] == 0 and np.array_equal(r["poses"][0], np.asarray(feed[0]["pose"], dtype=np.float64))
    own = np.isin(feed[1]["obs_lm"], feed[1]["new_ids"])                      # ... and the window goes on (the landmarks of keyframe 1 went with it)
    win.push(2, feed[1]["pose"], new_ids=feed[1]["new_ids"], new_xyz=feed[1]["new_xyz"], new_fixed=feed[1]["new_fixed"],
             obs_lm=feed[1]["obs_lm"][own], obs_uv=feed[1]["obs_uv"][own], obs_cam=feed[1]["obs_cam"][own])
    assert own.sum() > 5 and win.size()[0] == 2 and win.size()[2] == int(own.sum())
    g = win.solve()
    assert g["n_iters"] > 0 and np.isfinite(g["poses"]).all()
    win.pop(1); win.pop(2)
    assert win.size() == (0, 0, 0)
    win.close()


# ---- the window driven like the reference's backend drives its map (backend.cpp:205-244, map.cpp:89-194) -----------------------

def _drive(ctx, frames, n_active, jac, golden=None, open_loop=False):
    """Replays a drive (tools.mapmodel.make_window_scenario) the way ssvio's backend would: per keyframe the map changes
    (insert, drop a keyframe, drop unobserved map points, delete condemned ones), the window receives those edits, and is
    solved; the SAME graph re-marshalled from the map (keyframes / map points ascending by id, backend.cpp:88-169) goes through
    a fresh ssx_ba_solve.  Window and fresh solve must agree BIT FOR BIT at every keyframe -- contents, fixed flags, result --
    and the result is written back like backend.cpp:205-244 does (outliers unlinked, ...), which produces the next edits.
    golden: the reference's own run of the drive.  Closed loop (default): the map advances with OUR results, the comparison
    with the reference is about decisions (and drifts by what a gauge-free window amplifies).  open_loop: the map advances
    with the REFERENCE's results (and the window's estimate is overwritten with them), so every keyframe's optimisation starts
    from the reference's own state and is compared one to one."""
    from tools.mapmodel import ActiveMap, apply_edits
    m = ActiveMap(n_active)
    win = ba.BaWindow(ctx, m.K, m.cam_ext, jac_mode=jac, fix_rule=1)
    # worst over the keyframes whose window is pinned by at least one fixed map point | over the gauge-free ones (the first
    # n_active - 1 keyframes of a drive: nothing has left the window yet, so backend.cpp:125-130 fixes nothing)
    worst = {k: dict(pose=0.0, resid=0.0, frac=1.0, p99=0.0, chi2_rel=0.0, n=0, trial_mismatch=0) for k in ("pinned", "free")}
    for r, fr in enumerate(frames):
        for l in fr["condemn"]:
            m.condemn(l)
        m.insert_keyframe(fr["kf_id"], fr["pose"], fr["obs"], fr["new_points"], fr["victim"])
        apply_edits(win, m.take_edits())
        pr, kf_ids, lm_ids, e_feat = m.problem()
        ex = win.export()
        # 1. the window holds exactly the graph the reference would build from its map
        assert list(ex["kf_ids"]) == kf_ids and list(ex["lm_ids"]) == lm_ids, r
        np.testing.assert_array_equal(ex["point_fixed"], pr["point_fixed"])          # backend.cpp:125-130, kept by the window itself
        np.testing.assert_array_equal(ex["poses"], pr["poses"]); np.testing.assert_array_equal(ex["points"], pr["points"])
        key_w = ex["edge_pose"].astype(np.int64) * 10 ** 7 + ex["edge_point"]
        key_m = pr["edge_pose"].astype(np.int64) * 10 ** 7 + pr["edge_point"]
        assert len(np.unique(key_w)) == len(key_w) and np.array_equal(np.sort(key_w), np.sort(key_m)), r
        to_w = np.argsort(key_w)[np.argsort(np.argsort(key_m))]                      # map edge -> window edge
        np.testing.assert_array_equal(ex["edge_uv"][to_w], pr["edge_uv"])
        # 2. same bits as the re-marshalled map
        fresh = ba.ba_solve(ctx, pr, jac_mode=jac)
        got = win.solve()
        np.testing.assert_array_equal(got["poses"], fresh["poses"]); np.testing.assert_array_equal(got["points"], fresh["points"])
        np.testing.assert_array_equal(got["edge_chi2"][to_w], fresh["edge_chi2"])
        np.testing.assert_array_equal(got["trials"], fresh["trials"]); np.testing.assert_array_equal(got["chi2"], fresh["chi2"])
        # 3. the reference's own run of the same drive (g2o through oracle/_ref, tests/golden/ref_window.npz)
        if golden is not None:
            G = golden
            assert list(G[f"w{r}_kf_ids"]) == kf_ids and list(G[f"w{r}_lm_ids"]) == lm_ids, f"keyframe {r}: the map took another path than the reference's"
            np.testing.assert_array_equal(np.unpackbits(G[f"w{r}_fixed"])[:pr["L"]], pr["point_fixed"])
            np.testing.assert_array_equal(np.unpackbits(G[f"w{r}_outlier"])[:pr["E"]], fresh["edge_outlier"])   # the same edges are culled
            assert fresh["rounds"] == int(G[f"w{r}_rounds"])
            W = worst["pinned" if pr["point_fixed"].any() else "free"]
            W["n"] += 1
            same_trials = len(fresh["trials"]) == len(G[f"w{r}_trials"]) and np.array_equal(fresh["trials"], G[f"w{r}_trials"])
            W["trial_mismatch"] += 0 if same_trials else 1
            n_c = min(len(fresh["chi2"]), len(G[f"w{r}_chi2"]))
            # (windows of one or two keyframes with nothing fixed: the cost falls to ~1e-24; below 1e-7 of where it started its
            # digits are rounding noise in the reference too)
            rel = float(np.max(np.abs(fresh["chi2"][:n_c] - G[f"w{r}_chi2"][:n_c]) / (np.abs(G[f"w{r}_chi2"][:n_c]) + 1e-7 * float(G[f"w{r}_chi2"][0]))))
            d = np.abs(np.sqrt(fresh["edge_chi2"][::5]) - np.sqrt(G[f"w{r}_edge_chi2"]))
            dp = float(np.abs(fresh["poses"] - G[f"w{r}_poses"]).max())
            W["chi2_rel"] = max(W["chi2_rel"], rel)
            W["resid"] = max(W["resid"], float(d.max())); W["frac"] = min(W["frac"], float((d <= RESID_TOL).mean()))
            W["p99"] = max(W["p99"], float(np.percentile(d, 99))); W["pose"] = max(W["pose"], dp)
            if os.environ.get("SSX_TEST_VERBOSE"):
                print(f"  [kf {r}] fixed {int(pr['point_fixed'].sum())}/{pr['L']} trials {'same' if same_trials else 'DIFFER'} chi2 rel {rel:.2e}  |r - r_ref| max {d.max():.2e} "
                      f"p99 {np.percentile(d, 99):.2e} within {100 * (d <= RESID_TOL).mean():.2f} %  pose {dp:.2e}")
        if open_loop:
            m.apply(kf_ids, lm_ids, e_feat, G[f"w{r}_poses"], G[f"w{r}_points"], fresh["edge_outlier"])
            for k, p in zip(kf_ids, G[f"w{r}_poses"]):
                win.set_pose(k, p)
            for l, x in zip(lm_ids, G[f"w{r}_points"]):
                win.set_landmark(l, x)
        else:
            m.apply(kf_ids, lm_ids, e_feat, fresh["poses"], fresh["points"], fresh["edge_outlier"])
        if r % 2:
            # the same decisions handed back the short way: the flags of the solve as they are (ssx_ba_window_remove_flagged)
            n_before = win.size()[2]
            assert win.remove_flagged(np.zeros(n_before, np.uint8)) == 0 and win.size()[2] == n_before
            assert [k for k, _ in m.take_edits()] in (["remove_obs"], [])
            assert win.remove_flagged(got["edge_outlier"]) == int(got["edge_outlier"].sum())
        else:
            apply_edits(win, m.take_edits())                                          # ... or keyframe by keyframe (_remove_observations)
    win.close()
    return m.stats, worst


def test_window_driven_like_the_backend_equals_the_remarshalled_map(ctx):
    """ssx_ba_window_remove_* / pop with the first-observer rule / landmarks that come back: 14-keyframe drives through windows
    of 4 .. 7 keyframes, non-oldest keyframes dropped now and then, 4 % gross outliers, map points condemned by the frontend"""
    from tools.mapmodel import make_window_scenario
    for seed, n_active, jac in ((1, 5, ba.JAC_ANALYTIC), (2, 4, ba.JAC_NUMERIC_G2O), (5, 7, ba.JAC_ANALYTIC)):
        frames = make_window_scenario(n_kf=14, n_active=n_active, seed=seed)
        stats, _ = _drive(ctx, frames, n_active, jac)
        assert stats["outlier_edges"] > 100 and stats["condemned"] > 5 and stats["fixed_by_rule"] > 500, stats


def _golden_drive():
    from tools.mapmodel import make_window_scenario
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_window.npz"))
    n_kf, n_active, new_per_kf, track_len, seed = (int(x) for x in G["cfg"])
    frames = make_window_scenario(n_kf=n_kf, n_active=n_active, new_per_kf=new_per_kf, track_len=track_len, seed=seed)
    s = np.array([sum(f["pose"].sum() for f in frames), sum(float(np.sum([uv for _, uv in f["obs"]])) for f in frames),
                  sum(float(np.sum(list(f["new_points"].values()))) for f in frames)])
    np.testing.assert_allclose(s, G["input_sum"], rtol=1e-12)                         # the generator has not drifted
    return G, frames, n_active


@pytest.mark.parametrize("jac", [ba.JAC_NUMERIC_G2O, ba.JAC_ANALYTIC])
def test_window_drive_takes_the_reference_backends_decisions(ctx, jac, record_property):
    """CLOSED LOOP against tests/golden/ref_window.npz: there the REAL g2o optimised every window of a 14-keyframe drive and
    backend.cpp:205-244's edits were applied to the map; here the resident window does, through the C ABI, its own results
    feeding the next keyframe.  At every keyframe: the same map (keyframes, map points, fixed flags), the same outlier edges
    culled, the same rounds and LM trial counts.  The states of the two runs drift apart by what the reference's gauge- and
    scale-free window (no fixed keyframe, monocular edges) amplifies from one keyframe to the next -- ~2e-4 m in the poses,
    <= 2e-3 px in single residuals -- never near a decision (the closest edge of the run is 0.04 away from the chi2 threshold)."""
    G, frames, n_active = _golden_drive()
    stats, worst = _drive(ctx, frames, n_active, jac, golden=G)
    for kind, w in worst.items():
        print(f"[closed loop jac={jac}] worst of the {w['n']} {kind} windows: chi2 rel {w['chi2_rel']:.2e}, |pose - ref| {w['pose']:.2e}, |r - r_ref| max {w['resid']:.2e} "
              f"p99 {w['p99']:.2e} px, within 1e-4 px: {100 * w['frac']:.2f} %, LM trial counts differ in {w['trial_mismatch']}")
        for k, v in w.items():
            record_property(f"{kind}_{k}", v)
    assert [stats[k] for k in ("reentered", "fixed_by_rule", "condemned", "outlier_edges")] == list(G["stats"])
    assert stats["reentered"] > 0
    for w in worst.values():
        assert w["chi2_rel"] < 2e-3 and w["pose"] < 1e-3 and w["p99"] < 3e-3


@pytest.mark.parametrize("jac", [ba.JAC_NUMERIC_G2O, ba.JAC_ANALYTIC])
def test_window_drive_matches_the_reference_backend_golden(ctx, jac, record_property):
    """OPEN LOOP against the same vectors: after every keyframe the map and the window take the REFERENCE's poses and positions
    (ssx_ba_window_set_pose / _set_landmark), so each of the 14 optimisations starts from the state g2o started from and is
    compared one to one.  For the 9 windows that hold at least one fixed map point (every window once the first keyframe has
    left): identical LM trial counts, poses to 2e-6, residuals within 1e-4 px of the reference's at p99 (>= 99 % of them; the
    rest are outlier edges of tens of pixels, <= 1e-3 px).  The 5 windows before that are gauge- and scale-free in the reference
    (no fixed keyframe, left-image edges only): looser bars, stated below."""
    G, frames, n_active = _golden_drive()
    stats, worst = _drive(ctx, frames, n_active, jac, golden=G, open_loop=True)
    for kind, w in worst.items():
        print(f"[open loop jac={jac}] worst of the {w['n']} {kind} windows: chi2 rel {w['chi2_rel']:.2e}, |pose - ref| {w['pose']:.2e}, |r - r_ref| max {w['resid']:.2e} "
              f"p99 {w['p99']:.2e} px, within 1e-4 px: {100 * w['frac']:.2f} %, LM trial counts differ in {w['trial_mismatch']}")
        for k, v in w.items():
            record_property(f"{kind}_{k}", v)
    assert [stats[k] for k in ("reentered", "fixed_by_rule", "condemned", "outlier_edges")] == list(G["stats"])
    pinned, free = worst["pinned"], worst["free"]
    assert pinned["n"] == 9 and free["n"] == 5
    assert pinned["trial_mismatch"] == 0
    # The bars: K_body = 2 (K_max = 4 for the single worst residual) times what the ORACLE's two builds (plain | fused multiply-adds) reach against the same vectors
    # (tests/golden/ref_noise_floor.npz, drive_open_*; make_noise_floor.py prints them).  Pinned windows, numeric | analytic: poses
    # 3.0e-7 | 2.2e-7, p99 7.5e-5 | 6.9e-5 px, 98.84 | 99.23 % within 1e-4 px, the largest single difference 7.1e-4 | 5.5e-4 px on a
    # 30 px outlier edge.  [Until round 5 the bar was ">= 99 % within 1e-4 px", set from what the GPU code of the day did (99.17 %);
    # the reference's own arithmetic restated on the CPU reaches 98.84 %, and so did the GPU after its reduced solve changed.]
    global _NOISE
    if _NOISE is None:
        _NOISE = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_noise_floor.npz"))
    K, KM = float(_NOISE["K_body"]), float(_NOISE["K_max"])
    for kind, w in (("pinned", pinned), ("free", free)):
        fl = {st: float(_NOISE[f"drive_open_jac{jac}_{kind}_{st}"]) for st in ("pose", "resid", "p99", "chi2_rel", "frac")}
        for st in ("pose", "resid", "p99", "chi2_rel"):
            record_property(f"{kind}_{st}_floor", fl[st])
            assert w[st] <= (KM if st == "resid" else K) * fl[st], (kind, st, w[st], fl[st])   # resid = the worst single residual
        assert 1.0 - w["frac"] <= K * (1.0 - fl["frac"]) + 1e-3, (kind, "fraction within 1e-4 px", w["frac"], fl["frac"])
    # north_star's statement where the reference's own noise allows it: the windows pinned by a fixed map point, at p99
    assert pinned["p99"] <= RESID_TOL and pinned["pose"] < 2e-6
    # gauge- AND scale-free windows (the reference fixes no keyframe and has left-image edges only; before the first keyframe
    # leaves, no map point is fixed either): 7 directions of the solution are set by rounding, in g2o as here
