#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X stereo-SLAM compute core.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--pairs B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): stereo frames/s of the front-end hot path on 1241x376 synthetic stereo, 2000 ORB features
per image: pyramid + grid FAST + octree + orientation + blur + BRIEF on both images, row-band Hamming matching,
DLT triangulation -- configs[1] of BASELINE.json ("C2").  One STEP = one batch of B stereo pairs per GPU, already
resident in HBM when the timed region starts; nothing is downloaded inside the timed region.  N > 1 runs one
process per GPU on independent pairs (replicas, no data-path collective): weak scaling.

Also reported in the same JSON line: BA LM-iterations/s of the local bundle adjustment (configs[2], "C3": 10
keyframes, 4000 landmarks, 20000 edges) -- landmark-sharded over the N GPUs with the RCCL all-reduce hook when
N > 1 --, the roofline of the dominant front-end kernel (HIP-event timing, live), and the CPU baseline (the CPU
oracle = a scalar single-thread port of the same algorithms; plus the reference's own g2o BA when
oracle/_ref/libssvio_ref.so is present).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)


def level_pixels(rows, cols, scale=1.2, nlevels=8):
    px = []
    s = np.float32(1.0)
    for l in range(nlevels):
        inv = np.float32(1.0) / s
        px.append(int(np.rint(np.float32(cols) * inv)) * int(np.rint(np.float32(rows) * inv)))
        s = np.float32(s * np.float32(scale))
    return px


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--pairs", type=int, default=64, help="stereo pairs per step per GPU (batch per launch)")
    ap.add_argument("--cpu-sample", type=int, default=40, help="stereo pairs timed on the CPU baseline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # SSX_BENCH_SINGLE_GPU_GLOO=1 is a TEST MODE for boxes with one GPU: all ranks share cuda:0, the process group is
    # gloo and the BA all-reduce hook is staged through host memory; it exercises every multi-rank code path of this
    # script (its numbers mean nothing).  The real thing is one rank per GPU over RCCL.
    test_mode = world > 1 and os.environ.get("SSX_BENCH_SINGLE_GPU_GLOO") == "1"
    dev_index = 0 if test_mode else local_rank
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if test_mode:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{dev_index}"))
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(dev_index)
    dev = torch.device(f"cuda:{dev_index}")

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if test_mode else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    import ssvio_amd
    from ssvio_amd import _lib, ba, orb
    from ssvio_amd.synth import KITTI_H, KITTI_W, make_ba_problem, make_stereo_pair

    stream = torch.cuda.Stream(device=dev)
    ctx = ssvio_amd.Context(dev_index, stream=stream.cuda_stream)

    def barrier():
        ctx.synchronize()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()

    # ---------------- front-end: B synthetic KITTI-shaped stereo pairs per GPU, resident in HBM ----------------
    B = args.pairs
    host = np.stack([np.stack(make_stereo_pair(seed=rank * 1000 + i)[:2]) for i in range(B)])   # [B][2][H][W] u8
    imgs = torch.from_numpy(host).to(dev)
    torch.cuda.synchronize(dev)
    counts = orb.stereo_batch_dev(ctx, imgs.data_ptr(), B, KITTI_W, KITTI_H, KITTI_W)   # plans, runs once, syncs
    for _ in range(args.warmup):
        orb.stereo_batch_enqueue(ctx)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        orb.stereo_batch_enqueue(ctx)
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(elapsed)
    frames = world * B * args.steps
    value = frames / elapsed

    # ---------------- per-kernel time with HIP events (same workload, profiling on) ----------------
    _lib.profile_begin(ctx)
    PROF_STEPS = 3
    for _ in range(PROF_STEPS):
        orb.stereo_batch_enqueue(ctx)
    kt = _lib.profile_end(ctx)
    px = level_pixels(KITTI_H, KITTI_W)
    I = 2 * B
    # algorithmic bytes every kernel must move per STEP (SURVEY.md section 8-D per-image figures x I images); a
    # kernel launched several times per step (k_resize: 7 levels, k_fast_cells: level 0 | levels 1..7) gets the
    # per-launch average
    kp_total = int(counts[:, 0].sum() + counts[:, 1].sum())
    algo_step = {
        "k_resize": I * (sum(px[:-1]) + sum(px[1:])),                 # read level l-1, write level l
        "k_fast_cells": I * sum(px) + 4.0 * 8000 * I,                 # read the pyramid once, write the candidates
        "k_octree": 4.0 * 8000 * I + 4.0 * kp_total,                  # read candidates, write the selection
        "k_gauss7": 2.0 * I * sum(px),                                # read + write the pyramid
        "k_orient_brief": (961.0 + 1369.0 + 60.0) * kp_total,         # 31x31 + 37x37 patches, 28+32 B out
        "k_row_bucket": 28.0 * kp_total / 2 + 4.0 * kp_total / 2,
        "k_match": 60.0 * kp_total + 8.0 * kp_total / 2,
        "k_triangulate_matches": (56.0 + 25.0) * kp_total / 2,
    }
    dom_name, dom_ms = None, 0.0
    kernels = {}
    for name, (calls, total_ms) in kt.items():
        kernels[name] = {"calls_per_step": calls / PROF_STEPS, "ms_per_step": total_ms / PROF_STEPS}
        if total_ms > dom_ms:
            dom_name, dom_ms = name, total_ms
    dom_calls = kt[dom_name][0]
    dom_avg_s = (dom_ms / dom_calls) * 1e-3
    dom_bytes = algo_step.get(dom_name, 0.0) / (dom_calls / PROF_STEPS)
    achieved = dom_bytes / dom_avg_s / 1e9 if dom_avg_s > 0 else 0.0
    # HBM traffic per launch from the committed PMC passes (tools/collect_profiles.sh: separate FETCH_SIZE /
    # WRITE_SIZE runs; no 16 B/lane streams here, so no FETCH doubling, see the header of the .md); only valid for
    # the batch size it was taken at
    traffic, traffic_src = None, None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            pt = json.load(f)
        if pt.get("pairs_per_step") == B and world == 1:
            key = [k for k in pt["bytes_per_launch"] if k.split("<")[0] == dom_name]
            if key:
                traffic, traffic_src = int(pt["bytes_per_launch"][key[0]]), pt.get("source")
    except (OSError, ValueError):
        pass
    roofline = {"bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                "avg_launch_us": round(dom_avg_s * 1e6, 2), "algorithmic_bytes_per_launch": int(dom_bytes),
                "launches_per_step": dom_calls / PROF_STEPS,
                "note": "the dominant kernel (grid FAST) is bound by integer VALU issue, not by HBM: see DESIGN.md section 4; "
                        "traffic = PMC bytes per launch from " + (traffic_src or "profiles/ (not available for this batch size)")}
    # whole-pipeline figure: 12.0 MB algorithmic bytes per stereo pair
    pipeline_gbs = 12.0e6 * value / world / 1e9

    # ---------------- local BA (C3) ----------------
    pr = make_ba_problem(P=10, L=4000, seed=1)
    ba_kwargs = {}
    if world > 1:
        from ssvio_amd import dist_ba
        pr_local = dist_ba.shard_problem(pr, rank, world)
        hook = dist_ba.make_allreduce_hook_host_staged(dev) if test_mode else dist_ba.make_allreduce_hook(dev)
        ba_kwargs = dict(allreduce=hook, rank=rank, world_size=world)
    else:
        pr_local = pr
    with torch.cuda.stream(stream):
        for _ in range(2):
            r = ba.ba_solve(ctx, pr_local, want_edges=False, **ba_kwargs)
        barrier()
        tb = time.perf_counter()
        BA_REP = 5
        n_it = 0
        for _ in range(BA_REP):
            r = ba.ba_solve(ctx, pr_local, want_edges=False, **ba_kwargs)
            n_it += r["n_iters"]
        barrier()
        ba_elapsed = time.perf_counter() - tb
    ba_elapsed = max_over_ranks(ba_elapsed)
    ba_iters_s = n_it / ba_elapsed
    ba_solve_ms = ba_elapsed / BA_REP * 1e3

    # ---------------- global BA (C4 shape), WEAK scaling over the GPUs ----------------
    # 500 keyframes on a loop, 10 000 landmarks PER GPU (6 observations each): at 8 GPUs this is BASELINE
    # configs[3] exactly (80 000 landmarks, 480 000 edges).  Landmarks are sharded, the 3000 x 3000 reduced system
    # is all-reduced (non-zero tiles only) and solved redundantly on every rank.
    C4_LM_PER_GPU = 10000
    pr4 = make_ba_problem(P=500, L=C4_LM_PER_GPU * world, obs_per_lm=6, seed=4, loop=True, fix_first_pose=True)
    pr4_local = dist_ba.shard_problem(pr4, rank, world) if world > 1 else pr4
    with torch.cuda.stream(stream):
        r4 = ba.ba_solve(ctx, pr4_local, outer_rounds=1, iters=10, want_edges=False, **ba_kwargs)
        barrier()
        tb = time.perf_counter()
        C4_REP = 2
        n_it4 = 0
        for _ in range(C4_REP):
            r4 = ba.ba_solve(ctx, pr4_local, outer_rounds=1, iters=10, want_edges=False, **ba_kwargs)
            n_it4 += r4["n_iters"]
        barrier()
        c4_elapsed = time.perf_counter() - tb
    c4_elapsed = max_over_ranks(c4_elapsed)
    c4 = {"workload": f"C4 shape: 500 KF on a loop x {C4_LM_PER_GPU * world} landmarks x {int(pr4['E'])} edges "
                      f"({C4_LM_PER_GPU} landmarks per GPU, weak scaling), analytic Jacobians, f64",
          "iters_per_s": round(n_it4 / c4_elapsed, 2),
          "edge_iters_per_s": round(float(pr4["E"]) * n_it4 / c4_elapsed, 1),
          "ms_per_lm_iteration": round(c4_elapsed / max(n_it4, 1) * 1e3, 3),
          "chi2_first_last": [float(r4["chi2"][0]), float(r4["chi2"][-1])],
          "sharding": f"landmarks over {world} GPUs + RCCL all-reduce of the non-zero 64x64 tiles" if world > 1 else "none"}

    # ---------------- CPU baseline (rank 0, N == 1 only) ----------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import pyoracle as po
        po.build()
        ns = max(1, min(args.cpu_sample, B))
        tc = time.perf_counter()
        for i in range(ns):
            L, R = host[i, 0], host[i, 1]
            kL, dL = po.orb_extract(L); kR, dR = po.orb_extract(R)
            idx, _ = po.stereo_match(kL, dL, kR, dR)
            m = idx >= 0
            uvL = np.stack([kL["x"][m], kL["y"][m]], 1).astype(np.float64)
            uvR = np.stack([kR["x"][idx[m]], kR["y"][idx[m]]], 1).astype(np.float64)
            po.triangulate(uvL, uvR, (718.856, 718.856, 607.1928, 185.2157), 386.1448 / 718.856)
        cpu_t = time.perf_counter() - tc
        cpu = {"value": round(ns / cpu_t, 3), "unit": "stereo frames/s", "cores": 1, "kind": "port",
               "sample": f"{ns} of the benchmark's stereo pairs through the CPU oracle (scalar C++ restatement, "
                         f"single thread; the reference's OpenCV path cannot be built: OpenCV absent)",
               "host_cores_available": os.cpu_count()}
        # the same port on many cores at once (one process per core, its own pairs): the path is embarrassingly
        # parallel over frames, so this is what the host CPUs could do with the scalar restatement
        try:
            from oracle import pool_worker
            nw = max(1, min(os.cpu_count() or 1, 64))
            rate, reported = pool_worker.frontend_all_cores(nw, pairs_per_worker=2)
            if reported >= max(1, nw // 2):
                cpu["all_cores"] = {"value": round(rate, 2), "unit": "stereo frames/s", "cores": reported,
                                    "sample": f"{reported} processes x 2 pairs each, started together (one per core)"}
        except Exception as e:                                       # the baseline is informative, never fatal
            cpu["all_cores_error"] = str(e)[:200]
        # BA: the reference's own g2o path when the compiled reference library travelled with the repo
        tc = time.perf_counter()
        if po.have_ref() and os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libssvio_ref.so")):
            rr = po.ba_solve(pr, "ref", outer_rounds=1)
            kind = "reference"
        else:
            rr = po.ba_solve(pr, "oracle", outer_rounds=1, jac_mode=1)
            kind = "port"
        ba_cpu_t = time.perf_counter() - tc
        cpu["ba"] = {"value": round(len(rr["chi2"]) / ba_cpu_t, 2), "unit": "BA LM iterations/s", "kind": kind, "cores": 1,
                     "sample": "one optimize(10) of the C3 graph (g2o numeric Jacobians, CSparse), single thread"}

    if rank == 0:
        out = {
            "metric": "stereo frames/s (ORB extract + row-band match + triangulate) on 1241x376",
            "value": round(value, 2), "unit": "stereo frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "C2: 1241x376 synthetic stereo, 2000 ORB feats/img, 8 levels, extract+match+triangulate",
                       "pairs_per_step_per_gpu": B, "parallelism": f"replicas x{world}" if world > 1 else "single GPU",
                       "avg_keypoints_per_image": round(kp_total / I, 1),
                       "avg_matches_per_pair": round(float(counts[:, 2].mean()), 1),
                       "avg_triangulated_per_pair": round(float(counts[:, 3].mean()), 1)},
            "roofline": roofline,
            "pipeline_algorithmic_GBps_per_gpu": round(pipeline_gbs, 3),
            "kernels": {k: {kk: round(vv, 4) for kk, vv in v.items()} for k, v in sorted(kernels.items())},
            "ba": {"workload": "C3: local BA, 10 KF x 4000 landmarks x 20000 edges, analytic Jacobians, f64",
                   "iters_per_s": round(ba_iters_s, 1), "ms_per_solve": round(ba_solve_ms, 3),
                   "lm_iterations_per_solve": n_it // BA_REP,
                   "sharding": f"landmarks over {world} GPUs + RCCL all-reduce" if world > 1 else "none",
                   "includes": "host<->device transfer of the problem and the host LM control loop"},
            "ba_c4": c4,
            "e2e_frames_per_s_with_one_local_BA_per_frame": round(1.0 / (1.0 / (value / world) + ba_solve_ms * 1e-3) * world, 2),
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
