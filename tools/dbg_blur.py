import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ssvio_amd
from ssvio_amd import orb as sorb
from ssvio_amd.synth import make_stereo_pair
from oracle import pyoracle as po
ctx = ssvio_amd.Context(0)
L = make_stereo_pair(seed=0)[0]
ex = sorb.ORBextractor(ctx)
ex.DetectAndCompute(L)
rows, cols = po.level_sizes(L.shape[0], L.shape[1])
prev = L
for l in range(8):
    lvl = ex.stage_level(l)
    ref = L if l == 0 else po.resize_linear(prev, rows[l], cols[l])
    bl = ex.stage_level(l, blurred=True); rb = po.gauss7(ref)
    d = bl.astype(int) - rb.astype(int)
    ys, xs = np.nonzero(d)
    print(l, 'pyr equal', np.array_equal(lvl, ref), 'blur mismatches', len(ys), 'of', d.size,
          'x range', (xs.min(), xs.max()) if len(xs) else None, 'y range', (ys.min(), ys.max()) if len(ys) else None,
          'max abs', np.abs(d).max(), 'sample', list(zip(ys[:6].tolist(), xs[:6].tolist(), d[ys[:6], xs[:6]].tolist())))
    prev = ref
