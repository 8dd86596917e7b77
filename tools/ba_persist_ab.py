"""Resident batch of C3 windows, persistent groups on / off, 1 and 2 groups of windows: wall time per solve and the per-kernel
GPU time (HIP events).   python tools/ba_persist_ab.py [windows] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssvio_amd
from ssvio_amd import ba, _lib
from tools.synth import make_ba_problem
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
REP = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ctx = ssvio_amd.Context(0)
wins = [make_ba_problem(P=10, L=4000, seed=1 + 17 * k) for k in range(4)]
batch = ba.BaBatch(ctx, [wins[i % 4] for i in range(B)], resident=True, with_edge_errors=False)
for rnd in range(1):
    for persist in (0, 1):
        for groups in (1, 2):
            batch.set_persistent(persist); batch.set_groups(groups)
            for _ in range(2): batch.solve(download=False)
            t = time.perf_counter(); n = 0
            for _ in range(REP): n += batch.solve(download=False)["n_iters_total"]
            dt = (time.perf_counter() - t) / REP
            print(f"B={B} persist={persist} groups={groups}: {dt * 1e3:.3f} ms per batch solve, {B / dt:.0f} windows/s, {n / REP / dt:.0f} LM it/s", flush=True)
for persist in (0, 1):
    batch.set_persistent(persist); batch.set_groups(1)
    batch.solve(download=False)
    _lib.profile_begin(ctx); batch.solve(download=False); kt = _lib.profile_end(ctx)
    print(f"--- per kernel, persist={persist}, one group")
    for k, (c, ms) in sorted(kt.items(), key=lambda x: -x[1][1]): print('%-22s calls %5d total_ms %8.3f avg_us %8.1f' % (k, c, ms, 1e3 * ms / c))
batch.close(); ctx.close()
