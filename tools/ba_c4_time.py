import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ssvio_amd
from ssvio_amd import ba, _lib
from tools.synth import make_ba_problem
ctx = ssvio_amd.Context(0)
t = time.time(); pr = make_ba_problem(P=500, L=80000, obs_per_lm=6, seed=4, loop=True, fix_first_pose=True); print('gen s', time.time() - t, 'E', pr['E'])
for _ in range(2): r = ba.ba_solve(ctx, pr, outer_rounds=1, iters=10, want_edges=False)
t = time.time(); r = ba.ba_solve(ctx, pr, outer_rounds=1, iters=10, want_edges=False); dt = time.time() - t
print('C4 1 GPU: iters', r['n_iters'], 'trials', int(r['trials'].sum()), 'wall s %.3f' % dt, 'gpu ms %.1f' % r['ms_total'], 'iters/s %.1f' % (r['n_iters'] / dt), 'chi2', r['chi2'][0], '->', r['chi2'][-1])
_lib.profile_begin(ctx); r = ba.ba_solve(ctx, pr, outer_rounds=1, iters=3, want_edges=False); kt = _lib.profile_end(ctx)
for k, (c, ms) in sorted(kt.items(), key=lambda x: -x[1][1]): print('%-22s calls %5d total_ms %8.3f' % (k, c, ms))
