"""Resident batch of C3 windows: wall time per ssx_ba_batch_solve and per-kernel GPU time (HIP events)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ssvio_amd
from ssvio_amd import ba, _lib
from tools.synth import make_ba_problem
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
REP = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctx = ssvio_amd.Context(0)
wins = [make_ba_problem(P=10, L=4000, seed=1 + 17 * k) for k in range(4)]
batch = ba.BaBatch(ctx, [wins[i % 4] for i in range(B)], resident=True, with_edge_errors=False)
for _ in range(2): batch.solve(download=False)
t = time.time(); n = 0
for _ in range(REP): n += batch.solve(download=False)["n_iters_total"]
dt = (time.time() - t) / REP
print(f"B={B}: {dt * 1e3:.3f} ms per batch solve, {B / dt:.0f} windows/s, {n / REP / dt:.0f} LM it/s")
_lib.profile_begin(ctx); batch.solve(download=False); kt = _lib.profile_end(ctx)
for k, (c, ms) in sorted(kt.items(), key=lambda x: -x[1][1]): print('%-22s calls %5d total_ms %8.3f avg_us %8.1f' % (k, c, ms, 1e3 * ms / c))
