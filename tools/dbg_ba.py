import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ssvio_amd
from ssvio_amd import ba
from ssvio_amd.synth import make_ba_problem
from oracle import pyoracle as po
ctx = ssvio_amd.Context(0)
pr = make_ba_problem(P=4, L=60, obs_per_lm=4, seed=2)
g = ba.ba_solve(ctx, pr); o = po.ba_solve(pr, 'oracle', jac_mode=0)
np.set_printoptions(linewidth=200, precision=10)
print('g trials', g['trials']); print('o trials', o['trials'])
print('g chi2', g['chi2']); print('o chi2', o['chi2'])
print('g lam', g['lam']); print('o lam', o['lam'])
