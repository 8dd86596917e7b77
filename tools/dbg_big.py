import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ssvio_amd
from ssvio_amd import ba
from ssvio_amd.synth import make_ba_problem
from oracle import pyoracle as po
ctx = ssvio_amd.Context(0)
pr = make_ba_problem(P=70, L=2500, obs_per_lm=4, seed=33, fix_first_pose=True)
g = ba.ba_linearize(ctx, pr); o = po.ba_linearize(pr, jac_mode=0)
for k in ("err", "Hll", "bl", "Hpl", "Hpp", "bp"):
    d = np.abs(g[k] - o[k]); print(k, d.max(), np.abs(o[k]).max())
d = np.abs(g["err"] - o["err"]).max(1); w = np.argsort(d)[-5:]
print('worst edges', w, d[w], 'pose', pr['edge_pose'][w], 'pt', pr['edge_point'][w], 'err g', g['err'][w], 'o', o['err'][w])
print('n mismatched', (d > 1e-6).sum(), 'of', len(d), 'first idx', np.nonzero(d > 1e-6)[0][:10])
