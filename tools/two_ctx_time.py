"""Front-end throughput with the batch split over several contexts (streams) whose steps interleave:
   python tools/two_ctx_time.py [pairs_per_ctx] [n_ctx] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ssvio_amd
from ssvio_amd import orb
from tools.synth import KITTI_H, KITTI_W, make_stereo_pair
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
NC = int(sys.argv[2]) if len(sys.argv) > 2 else 2
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = torch.device("cuda:0")
host = np.stack([np.stack(make_stereo_pair(seed=i)[:2]) for i in range(B)])
ctxs, imgs = [], []
for c in range(NC):
    s = torch.cuda.Stream(device=dev)
    ctx = ssvio_amd.Context(0, stream=s.cuda_stream)
    im = torch.from_numpy(host).to(dev)
    torch.cuda.synchronize(dev)
    orb.stereo_batch_dev(ctx, im.data_ptr(), B, KITTI_W, KITTI_H, KITTI_W)
    ctxs.append((ctx, s)); imgs.append(im)
for rep in range(2):
    for ctx, _ in ctxs: ctx.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
        for ctx, _ in ctxs: orb.stereo_batch_enqueue(ctx)
    for ctx, _ in ctxs: ctx.synchronize()
    dt = time.perf_counter() - t
print(f"pairs/ctx {B} contexts {NC}: {B * NC * steps / dt:.0f} stereo frames/s ({dt / steps * 1e3:.3f} ms per round of {NC} steps)")
