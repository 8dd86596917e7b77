"""Aggregate local-BA throughput with several independent windows in flight: one ssx context (own stream, own
workspace) per host thread; ctypes releases the GIL inside ssx_ba_solve.  The C3 solve is a chain of small
latency-bound kernels, so independent solves overlap on the GPU."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssvio_amd
from ssvio_amd import ba
from tools.synth import make_ba_problem
REP = 20
for nthr in (1, 2, 4, 8, 16):
    ctxs = [ssvio_amd.Context(0) for _ in range(nthr)]
    prs = [make_ba_problem(P=10, L=4000, seed=1 + k) for k in range(nthr)]
    for c, p in zip(ctxs, prs): ba.ba_solve(c, p, want_edges=False)
    its = [0] * nthr
    def work(k):
        n = 0
        for _ in range(REP): n += ba.ba_solve(ctxs[k], prs[k], want_edges=False)["n_iters"]
        its[k] = n
    th = [threading.Thread(target=work, args=(k,)) for k in range(nthr)]
    t = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    dt = time.perf_counter() - t
    print(f"{nthr:2d} concurrent windows: {sum(its) / dt:8.0f} LM iterations/s, {nthr * REP / dt:7.1f} solves/s")
    for c in ctxs: c.close()
