"""k_solve64 A/B: the same single-window and batched solves on two builds of the library must return the same BYTES, and how long they take.
    python tools/solve64_ab.py out.npz     (run once per library: SSX_LIB=...; then compare the two files with --compare a.npz b.npz)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if sys.argv[1] == "--compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    bad = [k for k in a.files if not np.array_equal(a[k], b[k])]
    print("arrays", len(a.files), "different", bad)
    sys.exit(1 if bad else 0)
import ssvio_amd
from ssvio_amd import ba
from tools.synth import make_ba_problem
ctx = ssvio_amd.Context(0)
out = {}
cases = [dict(P=10, L=4000, seed=1), dict(P=10, L=700, seed=41), dict(P=4, L=60, obs_per_lm=4, seed=44), dict(P=7, L=300, obs_per_lm=2, seed=45),
         dict(P=10, L=500, seed=46, fix_first_pose=True, frac_fixed=0.3), dict(P=2, L=40, obs_per_lm=2, seed=47), dict(P=9, L=900, seed=48, fix_first_pose=True)]
for i, kw in enumerate(cases):
    pr = make_ba_problem(**kw)
    for jac in (ba.JAC_ANALYTIC, ba.JAC_NUMERIC_G2O):
        r = ba.ba_solve(ctx, pr, jac_mode=jac)
        for k in ("poses", "points", "chi2", "lam", "trials"):
            out[f"{i}_{jac}_{k}"] = np.asarray(r[k])
probs = [make_ba_problem(**kw) for kw in cases[:5]] * 4
rb = ba.BaBatch(ctx, probs, resident=True).solve()
for i, r in enumerate(rb["results"]):
    out[f"b{i}_poses"] = r["poses"]; out[f"b{i}_chi2"] = r["chi2"]
np.savez(sys.argv[1], **out)
pr = make_ba_problem(P=10, L=4000, seed=1, uv_f32=True)
for _ in range(5): r = ba.ba_solve(ctx, pr, want_edges=False)
best = 1e9
for rep in range(5):
    t = time.perf_counter()
    for _ in range(20): r = ba.ba_solve(ctx, pr, want_edges=False)
    best = min(best, (time.perf_counter() - t) / 20)
print(f"one window: wall {best * 1e3:.4f} ms, gpu {r['ms_total']:.3f} ms")
