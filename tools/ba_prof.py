import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, numpy as np
import ssvio_amd
from ssvio_amd import ba, _lib
from tools.synth import make_ba_problem
ctx = ssvio_amd.Context(0)
pr = make_ba_problem(P=10, L=4000, seed=1)
for _ in range(3): ba.ba_solve(ctx, pr, want_edges=False)
_lib.profile_begin(ctx)
r = ba.ba_solve(ctx, pr, want_edges=False)
kt = _lib.profile_end(ctx)
tot = 0
for k, (c, ms) in sorted(kt.items(), key=lambda x: -x[1][1]):
    print('%-22s calls %4d avg_us %8.2f total_ms %7.3f' % (k, c, ms / c * 1e3, ms)); tot += ms
print('sum kernels ms', tot, 'iters', r['n_iters'], 'gpu ms_total', r['ms_total'])
t = time.time(); N = 20
for _ in range(N): r = ba.ba_solve(ctx, pr, want_edges=False)
print('wall ms/solve', (time.time() - t) / N * 1e3)
