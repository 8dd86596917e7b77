"""ssx_ba_solve_batch on host arrays: where a call's time goes.  python tools/batch_time.py [windows] [calls]
(SSX_BATCH_TIMING makes the library print its phases: the timed syncs it adds serialise upload and marshalling, so the
wall time of an instrumented call is an upper bound of a normal one -- both are printed)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N = int(sys.argv[2]) if len(sys.argv) > 2 else 5
if os.environ.get("PHASES"):
    os.environ["SSX_BATCH_TIMING"] = "1"
import ssvio_amd
from ssvio_amd import ba
from tools.synth import make_ba_problem
ctx = ssvio_amd.Context(0)
probs = [make_ba_problem(P=10, L=4000, seed=100 + i, uv_f32=not os.environ.get("UV_F64")) for i in range(min(B, 8))]
wins = [probs[i % len(probs)] for i in range(B)]
bh = ba.BaBatch(ctx, wins)
br = ba.BaBatch(ctx, wins, resident=True, with_edge_errors=False)
for want_edges in (False, True):
    bh.solve(want_edges=want_edges)
    t0 = time.perf_counter()
    for _ in range(N):
        r = bh.solve(want_edges=want_edges, summaries=False)
    t = (time.perf_counter() - t0) / N
    print(f"host arrays, B={B}, edge chi2 {'returned' if want_edges else 'not returned'}: {t * 1e3:.3f} ms per call ({r['n_iters_total']} LM iterations)")
br.solve(download=False)
t0 = time.perf_counter()
for _ in range(N):
    br.solve(download=False)
print(f"resident batch: {(time.perf_counter() - t0) / N * 1e3:.3f} ms per call")
