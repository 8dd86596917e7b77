"""Resident batch of 128 C3 windows split into 1 .. 8 groups of windows on as many streams: ms per batch solve.
    python tools/ba_groups_sweep.py [windows] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssvio_amd
from ssvio_amd import ba
from tools.synth import make_ba_problem
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
REP = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ctx = ssvio_amd.Context(0)
wins = [make_ba_problem(P=10, L=4000, seed=1 + 17 * k) for k in range(4)]
batch = ba.BaBatch(ctx, [wins[i % 4] for i in range(B)], resident=True, with_edge_errors=False)
for groups in (1, 2, 3, 4, 6, 8):
    try:
        batch.set_groups(groups)
    except Exception as e:                       # noqa: BLE001
        print(groups, "groups:", e); continue
    for _ in range(2): batch.solve(download=False)
    t = time.perf_counter()
    for _ in range(REP): batch.solve(download=False)
    dt = (time.perf_counter() - t) / REP
    print(f"B={B} groups={groups}: {dt * 1e3:.3f} ms per batch solve", flush=True)
batch.close(); ctx.close()
