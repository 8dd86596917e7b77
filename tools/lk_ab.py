"""The LK call as the loop makes it, three ways: two fresh images, chained (one new image), 64 chained jobs in one call (pinned images).
Run once plain and once with SSX_LK_UNFUSED=1 (the per-level pyramid kernels) for the A/B; under rocprofv3 --kernel-trace --stats for
the kernels' own times.   python tools/lk_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ssvio_amd
from ssvio_amd import lk, orb
from tools.synth import make_stereo_pair
ctx = ssvio_amd.Context(0)
L, R, _ = make_stereo_pair(seed=0)
k, _ = orb.ORBextractor(ctx, nfeatures=300, nlevels=1).DetectAndCompute(L)
pts = np.stack([k["x"], k["y"]], 1).astype(np.float32)
guess = pts.copy(); guess[:, 0] -= 8.0
def tm(f, n=40, warm=4):
    for _ in range(warm): f()
    t = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t) / n
a = tm(lambda: lk.calcOpticalFlowPyrLK(ctx, L, R, pts, guess))
b = tm(lambda: lk.calcOpticalFlowPyrLK(ctx, None, R, pts, guess))
S = 64
imgs = [np.ascontiguousarray(np.roll(R, s_ % 7, axis=1)) for s_ in range(S)]
lk.track_batch(ctx, [dict(slot=s_, prev=L, next=imgs[s_], prev_pts=pts, next_pts=guess) for s_ in range(S)])
prep = lk.PreparedTrackBatch(ctx, [dict(slot=s_, prev=None, next=imgs[(s_ + 1) % S], prev_pts=pts, next_pts=guess) for s_ in range(S)])
c = tm(prep.run, 20, 2)
prep.close()
print(f"{'per-level kernels' if os.environ.get('SSX_LK_UNFUSED') else 'k_lk_pyramid'}: {len(pts)} points; two images {a * 1e3:.3f} ms, chained {b * 1e3:.3f} ms, 64 chained jobs {c * 1e3:.3f} ms per call")
