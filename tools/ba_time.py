import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssvio_amd
from ssvio_amd import ba
from tools.synth import make_ba_problem
ctx = ssvio_amd.Context(0)
for cfg in [dict(P=10, L=4000, seed=1), dict(P=12, L=1500, obs_per_lm=4, seed=5)]:
    pr = make_ba_problem(**cfg)
    for _ in range(3): r = ba.ba_solve(ctx, pr, want_edges=False)
    t = time.time(); N = 10
    for _ in range(N): r = ba.ba_solve(ctx, pr, want_edges=False)
    dt = (time.time() - t) / N
    print(cfg, 'iters', r['n_iters'], 'trials', r['trials'].sum(), 'wall ms/solve %.3f' % (dt * 1e3), 'gpu ms %.3f' % r['ms_total'], 'iters/s %.0f' % (r['n_iters'] / dt))
