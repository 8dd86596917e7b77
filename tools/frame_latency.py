"""Single stereo pair through the HOST-buffer entry point ssx_stereo_frame (upload + extract + match + triangulate +
download): the PCIe-inclusive latency / rate, never reported as bench.py's `value`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssvio_amd
from ssvio_amd import orb
from ssvio_amd.synth import make_stereo_pair
ctx = ssvio_amd.Context(0)
L, R, _ = make_stereo_pair(seed=0)
for _ in range(5): r = orb.stereo_frame(ctx, L, R)
N = 50
t = time.perf_counter()
for _ in range(N): r = orb.stereo_frame(ctx, L, R)
dt = (time.perf_counter() - t) / N
print(f"ssx_stereo_frame: {dt*1e3:.3f} ms per pair = {1/dt:.0f} pairs/s (host images in, keypoints/descriptors/matches/points out); "
      f"{r['n_matched']} matches, {r['n_triangulated']} triangulated")
