"""Single stereo pair through the HOST-buffer entry point ssx_stereo_frame (upload + extract + match + triangulate +
download): the PCIe-inclusive latency / rate, never reported as bench.py's `value`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssvio_amd
from ssvio_amd import orb
from tools.synth import make_stereo_pair
ctx = ssvio_amd.Context(0)
L, R, _ = make_stereo_pair(seed=0)
for _ in range(5): r = orb.stereo_frame(ctx, L, R)
N = 50
t = time.perf_counter()
for _ in range(N): r = orb.stereo_frame(ctx, L, R)
dt = (time.perf_counter() - t) / N
print(f"ssx_stereo_frame: {dt*1e3:.3f} ms per pair = {1/dt:.0f} pairs/s (host images in, keypoints/descriptors/matches/points out); "
      f"{r['n_matched']} matches, {r['n_triangulated']} triangulated")

# the library call alone: arguments marshalled once (what a C++ caller pays)
import ctypes as C
from ssvio_amd._lib import ptr, u8_p, dbl_p
o = orb.OrbParams(2000, 1.2, 8, 20, 7); mp = orb.match_params(scale_factor=o.scale_factor); rig = orb.stereo_rig()
fb = orb._FrameBuffers(o.nfeatures + 260 * o.nlevels + 64)
args = (ctx.handle, ptr(L, u8_p), ptr(R, u8_p), L.strides[0], L.shape[0], L.shape[1], C.byref(o), C.byref(mp), C.byref(rig), ptr(None, dbl_p), C.byref(fb.out))
for _ in range(5): ctx.check(ctx.lib.ssx_stereo_frame(*args))
t = time.perf_counter()
for _ in range(N): ctx.lib.ssx_stereo_frame(*args)
dt2 = (time.perf_counter() - t) / N
print(f"ssx_stereo_frame, C call only: {dt2*1e3:.3f} ms per pair = {1/dt2:.0f} pairs/s")
