"""Kernel timeline of ONE C3 ba_solve (run under rocprofv3 --kernel-trace); prints start offsets / durations / gaps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssvio_amd
from ssvio_amd import ba
from tools.synth import make_ba_problem
ctx = ssvio_amd.Context(0)
pr = make_ba_problem(P=10, L=4000, seed=1)
for _ in range(3): r = ba.ba_solve(ctx, pr, want_edges=False)
print("iters", r["n_iters"], "trials", int(r["trials"].sum()), "ms_total", r["ms_total"])
