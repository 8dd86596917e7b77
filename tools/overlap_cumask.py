"""Do the two legs of the composite step (front-end batch, batched local BA) overlap, and does partitioning the chip help?
Front-end on one ctx, BA on another, each on its own streams; with `fe_cus` > 0 the front-end's streams are restricted to the
first fe_cus compute units and the BA's to the rest (ssx_config.cu_first / cu_count -> hipExtStreamCreateWithCUMask).
    python tools/overlap_cumask.py [pairs] [steps]
Prints ms per step for: front-end alone, BA alone, both (free-for-all), both with several CU splits."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ssvio_amd
from ssvio_amd import ba, orb
from tools.synth import KITTI_H, KITTI_W, make_ba_problem, make_stereo_pair
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = torch.device("cuda:0")
host = np.stack([np.stack(make_stereo_pair(seed=i)[:2]) for i in range(B)])
imgs = torch.from_numpy(host).to(dev); torch.cuda.synchronize(dev)
wins = [make_ba_problem(P=10, L=4000, seed=1 + 17 * k) for k in range(4)]
step_windows = [wins[i % 4] for i in range(B)]


def run(fe_cus, label):
    n_cu = 256
    if fe_cus > 0:
        cf = ssvio_amd.Context(0, cu_first=0, cu_count=fe_cus)
        cb = ssvio_amd.Context(0, cu_first=fe_cus, cu_count=n_cu - fe_cus)
    else:
        cf = ssvio_amd.Context(0); cb = ssvio_amd.Context(0)
    orb.stereo_batch_dev(cf, imgs.data_ptr(), B, KITTI_W, KITTI_H, KITTI_W)
    batch = ba.BaBatch(cb, step_windows, resident=True, with_edge_errors=False)

    def sync():
        cf.synchronize(); cb.synchronize(); torch.cuda.synchronize(dev)

    def timed(fn):
        for _ in range(3): fn()
        sync(); t = time.perf_counter()
        for _ in range(STEPS): fn()
        sync(); return (time.perf_counter() - t) / STEPS * 1e3

    fe = timed(lambda: orb.stereo_batch_enqueue(cf))
    b_ = timed(lambda: batch.solve(download=False))

    def both():
        orb.stereo_batch_enqueue(cf)
        batch.solve(download=False)
    bo = timed(both)
    print(f"{label:28s} front-end {fe:6.3f} ms   BA {b_:6.3f} ms   both {bo:6.3f} ms   sum {fe + b_:6.3f}   max {max(fe, b_):6.3f}   -> {B / bo * 1e3:8.0f} frames/s", flush=True)
    batch.close(); cf.close(); cb.close()


run(0, "free-for-all")
for fe_cus in (64, 80, 96, 112, 128):
    run(fe_cus, f"front-end {fe_cus} CUs / BA {256 - fe_cus}")
run(0, "free-for-all (again)")
