"""Where the headline's distance to the all-resident step comes from: the four combinations of {images resident | streamed from the
host} x {nothing downloaded | poses of every window downloaded}.   python tools/headline_gap.py [pairs] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ssvio_amd
from ssvio_amd import ba, orb
from tools.synth import KITTI_H, KITTI_W, make_ba_problem, make_stereo_pair
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
stream = torch.cuda.Stream(device=dev)
ctx = ssvio_amd.Context(0, stream=stream.cuda_stream); ctx_ba = ssvio_amd.Context(0)
host = np.stack([np.stack(make_stereo_pair(seed=i)[:2]) for i in range(B)])
imgs = torch.from_numpy(host).to(dev); torch.cuda.synchronize()
orb.stereo_batch_dev(ctx, imgs.data_ptr(), B, KITTI_W, KITTI_H, KITTI_W)
wins = [make_ba_problem(P=10, L=4000, seed=1 + 17 * k, uv_f32=True) for k in range(16)]
batch = ba.BaBatch(ctx_ba, [wins[i % 16] for i in range(B)], resident=True, with_edge_errors=False)
ring = [torch.from_numpy(host).pin_memory(), torch.from_numpy(np.ascontiguousarray(host[::-1])).pin_memory()]
fe = orb.StereoStream(ctx, B, KITTI_H, KITTI_W)
def sync():
    ctx.synchronize(); ctx_ba.synchronize(); torch.cuda.synchronize()
for streamed in (0, 1):
    for download in (0, 1):
        if streamed:
            fe.upload(ring[0].data_ptr()); fe.run(); fe.upload(ring[1].data_ptr())
        k = [1]
        def step():
            if streamed:
                fe.run(); k[0] += 1; fe.upload(ring[k[0] & 1].data_ptr())
            else:
                orb.stereo_batch_enqueue(ctx)
            if download: batch.solve(want_edges=False, summaries=False, points=False)
            else: batch.solve(download=False)
            if streamed: fe.wait_counts()
        for _ in range(3): step()
        sync(); t = time.perf_counter()
        for _ in range(STEPS): step()
        sync(); dt = (time.perf_counter() - t) / STEPS
        if streamed:
            fe.wait_counts(); fe.run(); fe.wait_counts()
            orb.stereo_batch_dev(ctx, imgs.data_ptr(), B, KITTI_W, KITTI_H, KITTI_W)
        print(f"images {'streamed' if streamed else 'resident'}, poses {'downloaded' if download else 'left on the device'}: {dt * 1e3:.3f} ms per step, {B / dt:.0f} frames/s", flush=True)
batch.close(); ctx.close(); ctx_ba.close()
