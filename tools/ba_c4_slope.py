import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import ssvio_amd
from ssvio_amd import ba, _lib
from tools.synth import make_ba_problem
ctx = ssvio_amd.Context(0)
pr = make_ba_problem(P=500, L=80000, obs_per_lm=6, seed=4, loop=True, fix_first_pose=True)
for it in (1, 5, 10, 20, 40):
    best = 1e9
    for _ in range(4):
        t = time.time(); r = ba.ba_solve(ctx, pr, outer_rounds=1, iters=it, want_edges=False); best = min(best, time.time() - t)
    print('iters', it, 'n_iters', r['n_iters'], 'trials', int(r['trials'].sum()), 'wall ms %.3f' % (best * 1e3), 'gpu ms %.2f' % r['ms_total'])
