import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ssvio_amd
from ssvio_amd import ba, synth
from oracle import pyoracle as po
ctx = ssvio_amd.Context(0)
pr = synth.make_pose_graph_problem(P=60, n_loops=2, seed=11, meas_noise=0.02, drift=0.05)
r = ba.pose_graph_opt(ctx, pr, iters=3)
o = po.pose_graph_opt(pr, "oracle", iters=3)
print("gpu", r["n_iters"], r["chi2_initial"], r["chi2"], r["lambdas"], r["trials"])
print("orc", o["n_iters"], o["chi2"], o["lambdas"], o["trials"])
r1 = ba.pose_graph_opt(ctx, pr, iters=1)
e0 = np.array([po.pg_edge_eval(pr["meas"][k], pr["poses"][pr["ei"][k]], pr["poses"][pr["ej"][k]])[0] for k in range(pr["E"])])
print("chi2 of initial state (oracle edges)", (e0 ** 2).sum())
