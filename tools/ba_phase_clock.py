"""Per-phase shader-clock counts of ONE workgroup of the fused linearise + Schur kernel (k_lin_schur / k_lin_schur_b).
Needs a library built with the stamps compiled in:
    python tools/build_variant.py clock --patch tools/patches/ba_experiment_switches.diff -DSSX_PHASE_CLOCK   (then SSX_LIB=$PWD/ssvio_amd/libssx.so.clock)
Run on the GPU box: python tools/ba_phase_clock.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ssvio_amd
from ssvio_amd import ba
from tools.synth import make_ba_problem
ctx = ssvio_amd.Context(0)
pr = make_ba_problem(seed=3)
def dump(tag):
    out = (C.c_longlong * 16)()
    ctx.lib.ssx_debug_phase_clock(out)
    v = list(out)
    names = ["lin:load+jac", "lin:lm sums", "lin:pose blocks", "lin:reduce", "(gap)", "schur:load W/pairs/chol", "schur:Y", "schur:blocks", "schur:c"]
    print(tag, "total", v[9] - v[0], "cycles")
    for i, n in enumerate(names):
        print(f"   {n:28s} {v[i+1]-v[i]:8d}")
    print("   k_solve: load S", v[11]-v[10], " factor", v[12]-v[11], " back substitution", v[13]-v[12], " update + scale", v[14]-v[13])
for _ in range(2): ba.ba_solve(ctx, pr, want_edges=False)
dump("single window (79 WGs)")
wins = [make_ba_problem(seed=10 + i) for i in range(64)]
b = ba.BaBatch(ctx, wins, resident=True, with_edge_errors=False)
b.solve(download=False); b.solve(download=False)
dump("batch of 64 (WG 7 of window 0)")
