import sys, time, numpy as np, os
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/ssvio_amd") else os.getcwd())
import ssvio_amd
from ssvio_amd import ba
from tools.synth import make_ba_problem
ctx = ssvio_amd.Context(0)
for f32 in (True, False):
    pr = make_ba_problem(P=10, L=4000, seed=1, uv_f32=f32)
    for _ in range(5): r = ba.ba_solve(ctx, pr, want_edges=False)
    best = 1e9
    for rep in range(5):
        t = time.perf_counter(); N = 20
        for _ in range(N): r = ba.ba_solve(ctx, pr, want_edges=False)
        best = min(best, (time.perf_counter() - t) / N)
    print('uv_f32', f32, 'wall ms/solve %.4f' % (best * 1e3), 'gpu ms %.3f' % r['ms_total'])
