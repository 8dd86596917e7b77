"""Time ssx_lk_track (host-image entry point) on a KITTI-shaped stereo pair, 2000 points; run under rocprofv3 for kernel times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ssvio_amd
from ssvio_amd import lk, orb
from tools.synth import make_stereo_pair
ctx = ssvio_amd.Context(0)
L, R, _ = make_stereo_pair(seed=0)
k, _ = orb.ORBextractor(ctx, nfeatures=2000, nlevels=1).DetectAndCompute(L)
pts = np.stack([k["x"], k["y"]], 1).astype(np.float32)
for _ in range(3): r = lk.calcOpticalFlowPyrLK(ctx, L, R, pts, pts)
t = time.perf_counter(); N = 20
for _ in range(N): r = lk.calcOpticalFlowPyrLK(ctx, L, R, pts, pts)
dt = (time.perf_counter() - t) / N
print(f"points {len(pts)} tracked {int(r[1].sum())} ms/call {dt*1e3:.3f} (host images in, results out)")
t = time.perf_counter()
for i in range(N): r2 = lk.calcOpticalFlowPyrLK(ctx, None, L if i & 1 else R, pts, pts)
dt2 = (time.perf_counter() - t) / N
print(f"chained (ssx_lk_track_next, previous pyramid kept): ms/call {dt2*1e3:.3f}")
