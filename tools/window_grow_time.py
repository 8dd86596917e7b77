"""The first keyframes of a stream: a resident window (ssx_ba_window) that GROWS by one keyframe per solve -- what a live stream's
backend sees until its window is full -- against the same window solved again without a change, per solve in wall ms.
    python tools/window_grow_time.py [landmarks=1200] [keyframes=14]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ssvio_amd
from ssvio_amd import ba
from tools.synth import make_ba_problem

L = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
P = int(sys.argv[2]) if len(sys.argv) > 2 else 14
ctx = ssvio_amd.Context(0)
pr = make_ba_problem(P=P, L=L, obs_per_lm=5, seed=5, pose_t_noise=0.05, uv_f32=True)
first = np.full(pr["L"], 10 ** 9, dtype=np.int64)
np.minimum.at(first, pr["edge_point"], pr["edge_pose"])
for rep in range(2):
    win = ba.BaWindow(ctx, pr["K"], pr["cam_ext"])
    print(f"--- run {rep} ({'cold context' if rep == 0 else 'same context, new window'})")
    for k in range(P):
        new = np.nonzero(first == k)[0]
        e = np.nonzero(pr["edge_pose"] == k)[0]
        if k >= 10: win.pop(k - 10)
        win.push(k, pose=pr["poses"][k], new_ids=1000 + new, new_xyz=pr["points"][new], new_fixed=pr["point_fixed"][new], obs_lm=1000 + pr["edge_point"][e],
                 obs_uv=pr["edge_uv"][e], obs_cam=pr["edge_cam"][e])
        if k == 0: continue
        t0 = time.perf_counter(); r = win.solve(want_edges=True); t1 = time.perf_counter()
        nk, nl, no = win.size()
        t2 = time.perf_counter(); r2 = win.solve(want_edges=True); t3 = time.perf_counter()
        print(f"  {nk:2d} keyframes {nl:5d} landmarks {no:6d} observations: solve after the push {1e3 * (t1 - t0):7.3f} ms ({r['n_iters']} LM iterations), again unchanged {1e3 * (t3 - t2):7.3f} ms ({r2['n_iters']})")
    win.close()
