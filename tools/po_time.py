"""Time ssx_pose_only_opt (FrontEnd::EstimateCurrentPose) for a few feature counts; run under rocprofv3 for kernel times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssvio_amd
from ssvio_amd import ba
from tools.synth import make_pose_only_problem
ctx = ssvio_amd.Context(0)
for M in (100, 300, 500, 1000, 1500, 3000):
    pp = make_pose_only_problem(M=M, seed=1, frac_gross=0.05)
    for _ in range(3): r = ba.pose_only_opt(ctx, pp["pose"], pp["K"], pp["xyz"], pp["uv"])
    t = time.perf_counter(); N = 30
    for _ in range(N): r = ba.pose_only_opt(ctx, pp["pose"], pp["K"], pp["xyz"], pp["uv"])
    print(f"M {M:5d}  inliers {r['n_inliers']:5d}  {(time.perf_counter() - t) / N * 1e3:.3f} ms/call")
