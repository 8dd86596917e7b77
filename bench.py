#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X stereo-SLAM compute core.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--pairs B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): "stereo frames/s (ORB+match+local-BA) on 1241x376".  One STEP = one batch of B synthetic
1241x376 stereo pairs per GPU through the WHOLE hot path, in ONE timed region, the way a live system produces it:

  front-end   the batch is handed over as HOST images (a pinned ring; the upload of step k + 1's batch rides the library's
              copy stream beside step k's kernels, test/test_system.cpp:36-47 feeds a fresh pair per step): pyramid + grid
              FAST + octree + orientation + blur + BRIEF on both images (2000 features each), row-band Hamming matching, DLT
              triangulation (configs[1], "C2"); keypoint / match / triangulation counts of every pair downloaded
  backend     one local bundle adjustment per pair on a window that MOVES: B resident sliding windows (ssx_ba_window), each
              of which, per step, drops its oldest keyframe and takes a new one -- pose, ~400 new landmarks, 2000
              observations, the only data that crosses PCIe on the way in (ssx_ba_window_update_batch) -- is optimised
              where it lies (ssx_ba_window_solve_batch: Backend::OptimizeActiveMap with the reference's defaults, <= 5
              outer rounds x optimize(10), Huber 5.891) and returns the poses AND landmarks of the window to the host:
              /root/reference/src/ssvio/frontend.cpp:546-576 -> backend.cpp:57-76, 78-245, map.cpp:27-56, 89-160.
              10 keyframes x 20 000 observations per window (configs[2], "C3", on a window that moves: ~5200 landmarks,
              the partially observed ones at both ends included).  tools/bench_live.py drives it.

`value` = pairs completed per second, whole job, all GPUs.  N > 1 runs one process per GPU on independent pairs and
windows (replicas, no data-path collective): weak scaling.

Also in the same JSON line:
  frozen_batch    rounds 2-4's headline: the same front-end step beside a FROZEN batch of C3 windows (ssx_ba_batch: marshalled
                  and uploaded before the clock starts, re-solved from the same state every step, poses downloaded)
  resident        the frozen batch with the images resident in HBM too and nothing downloaded (rounds 1-3's `value`)
  c1              BASELINE configs[0] at its stated size: 200 KITTI-00-shaped stereo pairs (rendered corridor drive, the
                  reference's kitti_00.yaml settings) through the headless test_system (ssx_run_kitti): frames/s with and
                  without PNG decoding, APE against the generator's ground truth, the CPU oracle runner beside it
  frontend        the front-end alone (the same batch, its own timed region) -- what round 1 reported as `value`
  ba              C3 alone: LM iterations/s of one window at a time (latency) and of the batched entry point
  ba_c4           BA LM iterations/s on the configs[3] shape, landmark-sharded over the N GPUs through RCCL inside
                  libssx.so (ssx_comm_*) when N > 1
  next_rows       SURVEY.md 8-F's rows with their own lines: LK points/s, pose-only solves/s, vocabulary transform
                  descriptors/s, pose-graph iterations/s, each with algorithmic bytes / flops and the CPU time beside it
  roofline        the kernel with the largest share of the composite step: frac = the larger of the ALGORITHMIC hbm and
                  f64-flop fractions (SURVEY.md 8-D's bytes / flops per launch over the live launch duration, against the
                  guide's peaks); hbm_frac, flops_frac, traffic_ratio (PMC bytes / algorithmic bytes) as scalars beside it
  cpu_baseline    the same composite on ONE host core: CPU oracle front-end (scalar C++ restatement; OpenCV cannot be
                  built here) + the reference's own g2o BA (oracle/_ref/libssvio_ref.so) when it travelled
"""
from __future__ import annotations

import argparse
import concurrent.futures as cf
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
# VALU issue peak: 256 CUs x 4 SIMD-32; a wave64 VALU instruction issues over 2 cycles (guide, "Wave scheduling");
# 2.4 GHz -> 1024 x 2.4e9 / 2 wave-instructions per second
VALU_PEAK_GWIPS = 1024 * 2.4 / 2.0


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
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=128, help="stereo pairs (and BA windows) per step per GPU (measured on MI355X: "
                    "18.9 / 19.8 / 20.1 k frames/s at 64 / 128 / 256 -- the latency-bound kernels of the front-end want the larger batch)")
    ap.add_argument("--cpu-sample", type=int, default=16, help="stereo pairs + windows timed on the CPU baseline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--c1-frames", type=int, default=200, help="stereo pairs of the configs[0] leg (0 = skip it)")
    ap.add_argument("--profile-kernels", action="store_true", help="for the rocprofv3 passes (tools/collect_profiles.sh): the live-backend region "
                    "is left out, so that every batched BA kernel is launched in ONE shape (B windows per launch: the frozen batch and the "
                    "per-kernel pass) and rocprofv3's per-kernel averages can be set beside roofline.avg_launch_us; `value` is then the "
                    "frozen-batch composite and the line says so")
    ap.add_argument("--lean", action="store_true", help="only the headline region + the per-kernel pass (what tools/collect_profiles.sh "
                    "profiles: every launch of a kernel then has the same shape, so rocprofv3's per-kernel averages mean something)")
    args = ap.parse_args()

    # `python bench.py --gpus N` started WITHOUT a launcher: one rank per GPU is this script's job then -- it re-executes itself under
    # torch.distributed.run (one process per GPU over RCCL, rendezvous on 127.0.0.1) and fails loudly if that cannot be done; it never
    # measures one GPU and reports N.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        import torch
        have = torch.cuda.device_count()
        single_gpu_test = os.environ.get("SSX_BENCH_SINGLE_GPU_GLOO") == "1"
        if have < args.gpus and not single_gpu_test:
            sys.exit(f"bench.py: --gpus {args.gpus} but this node shows {have} GPU(s): refusing to report a {args.gpus}-GPU line")
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            port = s_.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
        print(f"[bench] no launcher in the environment: starting {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr)
        sys.exit(subprocess.call(cmd))

    rank = int(os.environ.get("RANK", "0"))
    # configs[0] leg: the 200-pair drive is rendered + written as PNGs by a process of its own (numpy ray casting, 0.65 s per pair
    # and core), started before this process touches the GPU and collected when the GPU regions are done
    c1_dir = f"/tmp/ssx_c1_corridor_{args.c1_frames}"
    c1_gen = None
    if rank == 0 and args.gpus == 1 and args.c1_frames > 0 and not args.lean and not os.path.exists(os.path.join(c1_dir, "times.txt")):
        import subprocess
        c1_gen = subprocess.Popen([sys.executable, "-m", "tools.synth", "corridor", c1_dir, str(args.c1_frames)], cwd=ROOT,
                                  stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    # the HARD drive of the same leg (twice the speed, 1 m sway: a keyframe every ~5 frames, windows of >= 4000 edges; tests/test_host_gpu.py
    # ::test_c1_hard_drive_keeps_the_backend_busy renders the same one)
    hard_dir = "/tmp/ssx_c1_hard_240"
    hard_gen = None
    if rank == 0 and args.gpus == 1 and args.c1_frames > 0 and not args.lean and not os.path.exists(os.path.join(hard_dir, "times.txt")):
        import subprocess
        hard_gen = subprocess.Popen([sys.executable, "-m", "tools.synth", "corridor", hard_dir, "240", "32", "1.6", "1.0"], cwd=ROOT,
                                    stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)

    import torch
    import torch.distributed as dist

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # SSX_BENCH_SINGLE_GPU_GLOO=1 is a TEST MODE for boxes with one GPU: all ranks share cuda:0, the process group is
    # gloo and the BA all-reduce goes through the callback hook staged in host memory; it exercises every multi-rank
    # code path of this script (its numbers mean nothing).  The real thing is one rank per GPU over RCCL.
    test_mode = world > 1 and os.environ.get("SSX_BENCH_SINGLE_GPU_GLOO") == "1"
    dev_index = 0 if test_mode else local_rank
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if test_mode:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{dev_index}"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: the line would claim GPUs that did not run"
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
    from tools.synth import KITTI_H, KITTI_W, make_ba_problem, make_stereo_pair

    stream = torch.cuda.Stream(device=dev)
    # SSX_BENCH_FE_PRIO=1 (tools): the front-end's stream at a LOW HIP priority -- its kernels then fill the spans the backend groups
    # leave idle instead of competing with them (profiles/r06/live_fe_priority_sweep.txt)
    if os.environ.get("SSX_BENCH_FE_PRIO", "").strip() not in ("", "0"):
        import ctypes as _C
        _hip = _C.CDLL(next((ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln), "libamdhip64.so"))
        _st = _C.c_void_p()
        _rc = _hip.hipStreamCreateWithPriority(_C.byref(_st), _C.c_uint(1), _C.c_int(int(os.environ["SSX_BENCH_FE_PRIO"])))
        assert _rc == 0 and _st.value, f"hipStreamCreateWithPriority -> {_rc}"
        stream = torch.cuda.ExternalStream(_st.value, device=dev)
    ctx = ssvio_amd.Context(dev_index, stream=stream.cuda_stream)     # front-end
    ctx_ba = ssvio_amd.Context(dev_index)                             # local BA windows, its own stream (the backend thread)

    def barrier():
        ctx.synchronize()
        ctx_ba.synchronize()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()

    # ---------------- inputs: B synthetic KITTI-shaped stereo pairs per GPU, resident in HBM; C3-shaped windows ----------------
    B = args.pairs
    host = np.stack([np.stack(make_stereo_pair(seed=rank * 1000 + i)[:2]) for i in range(B)])   # [B][2][H][W] u8
    imgs = torch.from_numpy(host).to(dev)
    torch.cuda.synchronize(dev)
    counts = orb.stereo_batch_dev(ctx, imgs.data_ptr(), B, KITTI_W, KITTI_H, KITTI_W)   # plans, runs once, syncs
    # the host side of the headline: a pinned ring of two batches (the same pairs in two orders: every step uploads 2 x B images)
    ring = [torch.from_numpy(host).pin_memory(), torch.from_numpy(np.ascontiguousarray(host[::-1])).pin_memory()]
    fe_stream = orb.StereoStream(ctx, B, KITTI_H, KITTI_W)
    h2d_bytes_per_step = int(ring[0].numel())
    N_WIN = 16 if not args.lean else 4                                 # distinct C3 graphs, used round-robin
    # (uv_f32: the measurements are float values, as the reference's keypoints are -- cv::KeyPoint::pt; it only matters to the
    # host-buffer regions, where the library then sends 8 instead of 16 bytes of coordinates per observation)
    wins = [make_ba_problem(P=10, L=4000, seed=1 + 17 * k + 1000 * rank, uv_f32=True) for k in range(N_WIN)]
    step_windows = [wins[i % N_WIN] for i in range(B)]
    # the windows are resident in HBM like the images (ssx_ba_batch_create: marshalled + uploaded before the clock starts);
    # a second, non-resident batch object measures the same work with host marshalling + PCIe inside the region
    batch = ba.BaBatch(ctx_ba, step_windows, resident=True, with_edge_errors=False)
    batch_host = ba.BaBatch(ctx_ba, step_windows)

    step_no = [0]

    # A streaming server's steady state: the batch of step k + 1 is crossing PCIe, the batch of step k is in the front-end, the
    # windows of step k are being edited / optimised, and the host collects the front-end's results ONE STEP BEHIND (batch k - 1:
    # they are there, nothing waits).  Every batch is uploaded, processed and its results downloaded exactly once; a result is
    # available one step (6 ms) after its batch entered the front-end.
    fe_stream.upload(ring[0].data_ptr()); fe_stream.run()              # batch 0 is in the front-end ...
    fe_stream.upload(ring[1].data_ptr())                               # ... batch 1 on its way when the first step starts
    step_no[0] = 1
    fe_pairs = [0]

    def frontend_step():
        fe_stream.run()                                                # batch k (uploaded during the last step) -> front-end, asynchronous
        step_no[0] += 1
        fe_stream.upload(ring[step_no[0] & 1].data_ptr())              # batch k + 1 starts crossing PCIe (the library's copy stream)

    def frontend_collect():
        c = fe_stream.wait_counts()                                    # the front-end's results of batch k - 1
        fe_pairs[0] += int((c[:, 0] > 0).sum())

    # (the configs[0] drive is rendered by worker processes on the host cores: finished before any clock starts -- the backend
    # groups' host phases of the headline region run on those cores)
    c1_gen_err = None
    if c1_gen is not None:
        _, c1_gen_err = c1_gen.communicate(timeout=1800)
    if hard_gen is not None:
        hard_gen.communicate(timeout=1800)

    # ---------------- timed region 1 (the headline): front-end + one local BA per pair on B live sliding windows ----------------
    if not args.profile_kernels:
        from tools import bench_live
        # three backend groups (host thread + context each), one stream per context, a group may be two steps behind the front-end's
        # release: measured best of the sweep in profiles/r05/live_backend_orchestration.md (the GPU has four hardware queues)
        LIVE_THREADS = int(os.environ.get("SSX_BENCH_WINDOW_THREADS", "3"))
        LIVE_LAG = max(1, int(os.environ.get("SSX_BENCH_LAG", "2")))       # steps a backend group may be behind the front-end's release
        live_steps = args.warmup + args.steps
        live = bench_live.LiveBackend(ssvio_amd, dev_index, B, live_steps, threads=LIVE_THREADS, seed=900 + 1000 * rank)

        def live_steps_run(n):
            # per step: the front-end's enqueue, the release of the step's keyframes to the backend groups, the front-end's results of
            # the batch before; the groups' step is awaited one step late (bench_live.LiveBackend.run: their host phases overlap each
            # other's kernels)
            def fe_part():
                frontend_step()
            for i in range(n):
                fe_part()
                live.release()
                frontend_collect()
                if i >= LIVE_LAG:
                    live.wait_done()
            for _ in range(min(LIVE_LAG, n)):
                live.wait_done()

        def sync_all():
            live.synchronize()
            barrier()

        live_steps_run(args.warmup)
        sync_all()
        live.reset_counters(); fe_pairs[0] = 0
        t0 = time.perf_counter()
        live_steps_run(args.steps)
        sync_all()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        frames = world * B * args.steps
        assert fe_pairs[0] == B * args.steps, "a stereo pair came back without keypoints"
        assert min(live.steps_done) == args.steps, "a backend group did not finish its steps"
        lm_iters = sum(live.iters)
        value = frames / elapsed
        nkf_w, nlm_w, nob_w = live.window_size()
        live_info = {"host_threads": live.G, "streams_per_group": live.batch_groups or 2, "steps_a_group_may_lag": LIVE_LAG, "ms_per_step_inside_solve_calls": round(max(live.t_solve) / args.steps * 1e3, 4),
                     "ms_per_step_inside_update_calls": round(max(live.t_edit) / args.steps * 1e3, 4),
                     "window": {"keyframes": nkf_w, "landmarks": nlm_w, "observations": nob_w},
                     "lm_iterations_per_window": round(lm_iters / (args.steps * B), 2),
                     "what": "ms_per_step_inside_solve_calls / _update_calls = the busiest group thread's time inside ssx_ba_window_solve_batch "
                             "(pending uploads, counting tables, device-side marshalling, solve, download of poses + landmarks) / inside "
                             "ssx_ba_window_update_batch, per step of the whole batch; the groups work side by side, so the calls of one group "
                             "wait for the GPU while the others' kernels run"}
        live.close()
        fe_stream.wait_counts()                                            # drain: the last batch that was run ...
        fe_stream.run(); fe_stream.wait_counts()                           # ... and the one uploaded ahead by the last step

    else:
        live_info = {"skipped": "--profile-kernels: value is the frozen-batch composite", "host_threads": 0, "window": None}
        lm_iters = 0
        fe_stream.wait_counts()
        fe_stream.run(); fe_stream.wait_counts()

    # ---------------- timed region 1f: rounds 2-4's headline -- the same front-end step beside a FROZEN batch of C3 windows ----------------
    fe_stream.upload(ring[0].data_ptr()); fe_stream.run()
    fe_stream.upload(ring[1].data_ptr())
    step_no[0] = 1

    def composite_step():
        frontend_step()
        # B windows on the BA stream: returns when they are done, with the optimised keyframe poses of every window ...
        batch.solve(want_edges=False, summaries=False, points=False)
        frontend_collect()

    def composite_step_resident():
        orb.stereo_batch_enqueue(ctx)                                  # asynchronous on the front-end stream, images resident
        return batch.solve(download=False)                             # B resident windows, nothing downloaded

    def composite_step_host():
        orb.stereo_batch_enqueue(ctx)
        return batch_host.solve(want_edges=False, summaries=False)                      # marshal + upload + solve + download poses / points

    FROZEN_STEPS = max(4, args.steps // 2)
    for _ in range(2):
        composite_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(FROZEN_STEPS):
        composite_step()
    barrier()
    frozen_elapsed = max_over_ranks(time.perf_counter() - t0)
    frozen_value = world * B * FROZEN_STEPS / frozen_elapsed
    if args.profile_kernels:
        value, elapsed, frames = frozen_value, frozen_elapsed * args.steps / FROZEN_STEPS, world * B * args.steps
        lm_iters = 10 * args.steps * B
    fe_stream.wait_counts()
    fe_stream.run(); fe_stream.wait_counts()
    orb.stereo_batch_dev(ctx, imgs.data_ptr(), B, KITTI_W, KITTI_H, KITTI_W)     # back to the resident images for the regions below

    # ---------------- timed region 1r: rounds 1-3's headline -- images and windows resident, nothing downloaded ----------------
    for _ in range(2):
        composite_step_resident()
    barrier()
    t0 = time.perf_counter()
    RES_STEPS = max(4, args.steps // 2)
    for _ in range(RES_STEPS):
        composite_step_resident()
    barrier()
    resident_elapsed = max_over_ranks(time.perf_counter() - t0)
    resident_value = world * B * RES_STEPS / resident_elapsed

    # ---------------- timed region 1b: the same with the windows handed over as HOST buffers every step ----------------
    host_value = float("nan")
    if not args.lean:
        composite_step_host()
        barrier()
        t0 = time.perf_counter()
        HOST_STEPS = max(2, args.steps // 4)
        for _ in range(HOST_STEPS):
            composite_step_host()
        barrier()
        host_elapsed = max_over_ranks(time.perf_counter() - t0)
        host_value = world * B * HOST_STEPS / host_elapsed

    # ---------------- timed region 1c: host buffers, TWO batches in flight (what a server fed by several streams does) ----------------
    # two host threads, each with its own context, marshal + upload + solve + download their batch while the other's is
    # on the GPU; the front-end steps of both are enqueued by this thread
    import threading
    pipe_value = None
    try:
        if args.lean:
            raise RuntimeError("--lean")
        ctx_p = [ssvio_amd.Context(dev_index) for _ in range(2)]
        bh = [ba.BaBatch(c, step_windows) for c in ctx_p]
        for b_ in bh:
            b_.solve(want_edges=False, summaries=False)
        PSTEPS = max(2, args.steps // 4)

        def pipe_worker(k):
            for _ in range(PSTEPS):
                bh[k].solve(want_edges=False, summaries=False)

        # (no collective in here: an exception on one rank must not leave the others waiting; rank 0's own clock, times the ranks)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        th = [threading.Thread(target=pipe_worker, args=(k,)) for k in range(2)]
        for t_ in th:
            t_.start()
        for _ in range(2 * PSTEPS):
            orb.stereo_batch_enqueue(ctx)
        for t_ in th:
            t_.join()
        torch.cuda.synchronize(dev)
        pipe_elapsed = time.perf_counter() - t0
        pipe_value = world * B * 2 * PSTEPS / pipe_elapsed
        for c in ctx_p:
            c.close()
    except Exception as exc:                                           # noqa: BLE001 -- an extra figure, never fatal
        if not args.lean:
            print(f"[bench] pipelined host-buffer region skipped: {exc}", file=sys.stderr)

    # ---------------- timed region 2: the front-end alone (round 1's `value`) ----------------
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        orb.stereo_batch_enqueue(ctx)
    barrier()
    fe_elapsed = max_over_ranks(time.perf_counter() - t0)
    fe_value = frames / fe_elapsed

    # ---------------- timed region 3: the batched BA alone, and one window at a time ----------------
    barrier()
    t0 = time.perf_counter()
    BA_REP = 3
    n_it_b = 0
    for _ in range(BA_REP):
        n_it_b += batch.solve(download=False)["n_iters_total"]
    barrier()
    bab_elapsed = max_over_ranks(time.perf_counter() - t0)
    pr = wins[0]
    for _ in range(2):
        ba.ba_solve(ctx_ba, pr, want_edges=False)
    barrier()
    t0 = time.perf_counter()
    n_it_1 = 0
    ONE_REP = 5
    for _ in range(ONE_REP):
        n_it_1 += ba.ba_solve(ctx_ba, pr, want_edges=False)["n_iters"]
    barrier()
    ba1_elapsed = max_over_ranks(time.perf_counter() - t0)

    # ---------------- single-pair latency: ssx_stereo_frame, HOST images in, host results out ----------------
    # (a live front-end is single-stream: the dependent-launch chain of ONE pair, PCIe both ways; never `value`)
    lat = None
    if rank == 0 and not args.lean:
        import ctypes as C
        from ssvio_amd._lib import ptr, u8_p, dbl_p
        ctx_lat = ssvio_amd.Context(dev_index)
        Lh, Rh, _ = make_stereo_pair(seed=0)
        o_ = orb.OrbParams(2000, 1.2, 8, 20, 7); mp_ = orb.match_params(scale_factor=o_.scale_factor); rig_ = orb.stereo_rig()
        fb = orb._FrameBuffers(o_.nfeatures + 260 * o_.nlevels + 64)
        a_ = (ctx_lat.handle, ptr(Lh, u8_p), ptr(Rh, u8_p), Lh.strides[0], Lh.shape[0], Lh.shape[1], C.byref(o_), C.byref(mp_),
              C.byref(rig_), ptr(None, dbl_p), C.byref(fb.out))
        for _ in range(5):
            ctx_lat.check(ctx_lat.lib.ssx_stereo_frame(*a_))
        LAT_REP = 40
        t0 = time.perf_counter()
        for _ in range(LAT_REP):
            ctx_lat.lib.ssx_stereo_frame(*a_)
        lat_s = (time.perf_counter() - t0) / LAT_REP
        lat = {"ms_per_pair": round(lat_s * 1e3, 4), "pairs_per_s": round(1.0 / lat_s, 1),
               "what": "ssx_stereo_frame, one 1241x376 pair at a time: pageable host images in (pinned staging, level 0 built over PCIe), "
                       "18 dependent launches on one stream, results written to pinned memory by the last kernel, one synchronisation, "
                       "host arrays out; arguments marshalled once (a C caller's cost)"}
        ctx_lat.close()

    # ---------------- per-kernel time with HIP events (same workload, profiling on) ----------------
    PROF_STEPS = 3
    _lib.profile_begin(ctx)
    for _ in range(PROF_STEPS):
        orb.stereo_batch_enqueue(ctx)
    kt = _lib.profile_end(ctx)
    ba_groups = batch.groups
    batch.set_groups(1)                  # per-kernel figures: every kernel alone on the chip, one launch for the whole batch
    batch.solve(download=False)
    _lib.profile_begin(ctx_ba)
    for _ in range(PROF_STEPS):
        batch.solve(download=False)
    kt_ba = _lib.profile_end(ctx_ba)
    batch.set_groups(0)
    # ---------------- the front-end on images of KITTI-like corner density (the bench's pairs are ~10 x corner-denser) ----------------
    fe_kitti = None
    if not args.profile_kernels:
        try:
            from tools.synth import fast9_density
            base_k = [np.stack(make_stereo_pair(seed=rank * 1000 + i, n_blobs=400)[:2]) for i in range(min(B, 8))]
            imgs_k = torch.from_numpy(np.stack([base_k[i % len(base_k)] for i in range(B)])).to(dev)
            counts_k = orb.stereo_batch_dev(ctx, imgs_k.data_ptr(), B, KITTI_W, KITTI_H, KITTI_W)
            for _ in range(2):
                orb.stereo_batch_enqueue(ctx)
            barrier()
            t0 = time.perf_counter()
            K_STEPS = max(4, args.steps // 2)
            for _ in range(K_STEPS):
                orb.stereo_batch_enqueue(ctx)
            barrier()
            fe_k_elapsed = max_over_ranks(time.perf_counter() - t0)
            _lib.profile_begin(ctx)
            for _ in range(PROF_STEPS):
                orb.stereo_batch_enqueue(ctx)
            kt_k = _lib.profile_end(ctx)
            fe_kitti = {"images": "make_stereo_pair(n_blobs=400): the same generator with a tenth of the blobs",
                        "fast9_corner_fraction": {"these": round(fast9_density(base_k[0][0]), 4), "bench_default": round(fast9_density(host[0][0]), 4),
                                                  "what": "interior pixels that are FAST-9 corners at iniThFAST = 20 (street photographs: 0.01 - 0.03)"},
                        "ms_per_step": round(fe_k_elapsed / K_STEPS * 1e3, 4), "stereo_frames_per_s": round(world * B * K_STEPS / fe_k_elapsed, 1),
                        "keypoints_per_image": round(float(counts_k[:, 0].sum() + counts_k[:, 1].sum()) / (2 * B), 1),
                        "kernels_ms_per_step": {k: round(v[1] / PROF_STEPS, 4) for k, v in sorted(kt_k.items())}}
            del imgs_k
        except Exception as exc:                                       # noqa: BLE001 -- an extra figure, never fatal
            print(f"[bench] KITTI-like density pass skipped: {exc}", file=sys.stderr)
        orb.stereo_batch_dev(ctx, imgs.data_ptr(), B, KITTI_W, KITTI_H, KITTI_W)     # back to the bench's own pairs
    px = level_pixels(KITTI_H, KITTI_W)
    I = 2 * B
    kp_total = int(counts[:, 0].sum() + counts[:, 1].sum())
    # algorithmic bytes every kernel must move per STEP (SURVEY.md section 8-D per-image / per-iteration figures)
    E3, L3, P3 = int(pr["E"]), int(pr["L"]), int(pr["P"])
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
    # BA kernels: algorithmic bytes per LAUNCH of the batched kernels (B windows), the terms of SURVEY.md 8-D's bytes_iter
    # (24 B per edge, 24 B per landmark state + 72 B of Hll / bl, 56 B per pose, 288 B per non-zero block of S) by kernel.
    # The edge blocks W = Ji^T w Jj are NOT materialised any more (recomputed where needed), so they do not count.
    # "k_lin_schur" is the fused linearise + Schur kernel of every LM slot but the first of an optimize() (9 of 10 per step);
    # "k_schur" the stand-alone Schur kernel of the first slot.
    # (The timed regions run the batch in `ba_groups` groups of windows side by side on as many streams; the per-kernel
    # pass above runs it as ONE group, so that a launch covers the B windows and has the chip to itself.)
    lin_b = 24.0 * E3 + 24.0 * L3 + 56.0 * P3 + 72.0 * L3
    Bl = B
    algo_launch_ba = {
        "k_linearize": Bl * lin_b,
        "k_lin_schur": Bl * (lin_b + 72.0 * L3 + 288.0 * 55),
        "k_schur": Bl * (24.0 * E3 + 24.0 * L3 + 56.0 * P3 + 72.0 * L3 + 288.0 * 55),
        "k_backsub_residual": Bl * (24.0 * E3 + 72.0 * L3 + 24.0 * L3 + 56.0 * P3 + 16.0 * E3),   # edges, Hll / bl, new points, poses, trial errors
        "k_solve64": Bl * (8.0 * 61 * 60 + 56.0 * 2 * P3),
    }
    kernels = {}
    dom = None
    for src, table in (("frontend", kt), ("ba", kt_ba)):
        for name, (calls, total_ms) in table.items():
            kernels[name] = {"calls_per_step": calls / PROF_STEPS, "ms_per_step": total_ms / PROF_STEPS, "part": src}
            if dom is None or total_ms / PROF_STEPS > kernels[dom]["ms_per_step"]:
                dom = name
    dom_calls = kernels[dom]["calls_per_step"]
    dom_avg_s = kernels[dom]["ms_per_step"] / max(dom_calls, 1e-9) * 1e-3
    if dom in algo_step:
        dom_bytes = algo_step[dom] / dom_calls
    else:
        dom_bytes = algo_launch_ba.get(dom.split("<")[0], 0.0)
    # PMC counters per launch from the committed passes (tools/collect_profiles.sh -> profiles/pmc_counters.json:
    # separate FETCH_SIZE / WRITE_SIZE / SQ passes); only valid for the batch size they were taken at
    traffic = valu_insts = rocprof_avg_us = None
    pmc_src = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_counters.json")) as f:
            pc = json.load(f)
        if pc.get("pairs_per_step") == B and pc.get("ba_groups", 1) == 1 and world == 1:
            base = dom.split("<")[0]
            names = [base + "_b", base] if kernels[dom]["part"] == "ba" else [base]     # the BA kernels of the step are the batched ones
            key = [k for nm in names for k in pc["per_launch"] if k.split("<")[0] == nm]
            if key:
                rec = pc["per_launch"][key[0]]
                rocprof_avg_us = rec.get("avg_us_rocprof_stats")
                traffic = rec.get("hbm_bytes")
                valu_insts = rec.get("SQ_INSTS_VALU")
                pmc_src = pc.get("source")
    except (OSError, ValueError):
        pass
    # ---- the roofline of the dominant kernel: SURVEY.md 8-D's fraction, from ALGORITHMIC work and the guide's peaks ----
    #   hbm    8-D's bytes per launch / live duration / 8 TB/s.  For the BA kernels the bytes are 8-D's
    #          bytes_iter = 24 E + 48 L + 56 P + 288 nnzb(S) + t (24 L + 48 P) with t = 1 -- the WHOLE LM iteration charged to the
    #          one kernel, exactly as 8-D defines the figure (0.785 MB per C3 window) -- times the windows a launch covers;
    #          `by_kernel_accounting` is this script's own per-kernel split of the same terms.
    #   flops  8-D's 23 Mflop per C3 iteration x windows per launch / duration / 78.6 TFLOP/s (f64 vector peak).
    # `frac` / `bound` = the larger of the two.  Hardware instruction counts do not enter it: the VALU-issue figures
    # (SQ_INSTS_VALU of the committed PMC pass) are UTILISATION numbers, reported under `utilisation`, never as a roofline.
    F64_PEAK_TFLOPS = 78.6
    is_ba = kernels[dom]["part"] == "ba"
    nnzb = P3 * (P3 + 1) // 2
    survey_bytes_iter = 24.0 * E3 + 48.0 * L3 + 56.0 * P3 + 288.0 * nnzb + 1 * (24.0 * L3 + 48.0 * P3)
    survey_flops_iter = 400.0 * E3 + L3 * (60.0 + 5 * 108.0 + 15 * 216.0)          # k = 5 observations per landmark: 23 Mflop at C3
    if is_ba:
        hbm_bytes = Bl * survey_bytes_iter
        flops = Bl * survey_flops_iter
    else:
        hbm_bytes = dom_bytes
        flops = None
    hbm_achieved = hbm_bytes / dom_avg_s / 1e9 if dom_avg_s > 0 else 0.0
    hbm = {"achieved": round(hbm_achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm_achieved / HBM_PEAK_GBS, 6),
           "algorithmic_bytes_per_launch": int(hbm_bytes)}
    if is_ba:
        own = dom_bytes / dom_avg_s / 1e9 if dom_avg_s > 0 else 0.0
        hbm["by_kernel_accounting"] = {"algorithmic_bytes_per_launch": int(dom_bytes), "achieved": round(own, 3), "frac": round(own / HBM_PEAK_GBS, 6)}
    # the name rocprofv3 prints for it: the BA kernels of the step are the batched instantiations with analytic Jacobians
    rocprof_name = {"k_lin_schur": "k_lin_schur_b<0>", "k_linearize": "k_linearize_b<0>"}.get(dom, dom + "_b" if is_ba else dom)
    roofline = {"kernel": rocprof_name, "profile_id": dom, "avg_launch_us": round(dom_avg_s * 1e6, 2), "launches_per_step": dom_calls,
                "algorithmic_bytes_per_launch": int(hbm_bytes), "traffic": traffic, "hbm": hbm,
                "hbm_frac": hbm["frac"], "flops_frac": None,
                "traffic_ratio": None if not traffic else round(traffic / hbm_bytes, 3)}
    cands = [("hbm", hbm["achieved"], HBM_PEAK_GBS, "GB/s", hbm["frac"])]
    if flops:
        tf = flops / dom_avg_s / 1e12
        roofline["flops"] = {"achieved": round(tf, 3), "peak": F64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / F64_PEAK_TFLOPS, 5),
                             "algorithmic_flops_per_launch": int(flops),
                             "definition": "SURVEY.md 8-D: 400 flop per edge + (60 + 108 k + 216 k (k + 1) / 2) per landmark with k = 5 "
                                           "observations = 23 Mflop per C3 iteration, x windows per launch; peak = MI355X f64 vector"}
        roofline["flops_frac"] = roofline["flops"]["frac"]
        cands.append(("f64 flops", roofline["flops"]["achieved"], F64_PEAK_TFLOPS, "TFLOP/s", roofline["flops"]["frac"]))
    if valu_insts:
        va = valu_insts / dom_avg_s / 1e9
        util = {"valu_issue": {"value": round(va / VALU_PEAK_GWIPS, 5), "achieved_G_wave_instr_per_s": round(va, 2), "peak": round(VALU_PEAK_GWIPS, 1),
                               "wave_instructions_per_launch": int(valu_insts),
                               "what": "SQ_INSTS_VALU per launch (the committed PMC pass named in `note`, NOT this run) / live launch duration / "
                                       "(1024 SIMD-32 x 2.4 GHz / 2 cycles per wave64 VALU instruction): how busy the issue slots are -- it "
                                       "rewards every extra instruction and is not a roofline fraction"}}
        if is_ba:
            # f64 instructions issue over 4 cycles (tools/microbench/f64_rate.hip); 54 of the 65 VALU instructions of the block
            # loop are f64 (profiles/r02/k_lin_schur_b_block_loop.s): 3.66 cycles per instruction on average
            mix_peak = 1024 * 2.4 / 3.66
            util["issue_utilisation_f64_mix"] = {"value": round(va / mix_peak, 5), "peak_G_wave_instr_per_s": round(mix_peak, 1),
                                                 "what": "the same numerator against a peak derated by the kernel's own instruction mix "
                                                         "(3.66 issue cycles per instruction)"}
        roofline["utilisation"] = util
    best = max(cands, key=lambda c: c[4])
    roofline.update(bound=best[0], achieved=best[1], peak=best[2], unit=best[3], frac=best[4])
    # the two durations the fraction can be taken with, named: this run's HIP events around the kernel ALONE on the chip (one group of
    # B windows, nothing beside it: avg_launch_us, what `frac` uses), and the average of the committed rocprofv3 --kernel-trace
    # --stats pass of `bench.py --lean --profile-kernels` (the same launch shape with the front-end's kernels beside it)
    roofline["frac_alone_on_chip"] = best[4]
    roofline["rocprof_avg_launch_us"] = rocprof_avg_us
    roofline["frac_from_rocprof_avg"] = (round(best[4] * dom_avg_s * 1e6 / rocprof_avg_us, 5) if rocprof_avg_us else None)
    roofline["note"] = ("kernel with the largest share of the composite step (named as rocprofv3 prints it); achieved = ALGORITHMIC bytes / flops "
                        "per launch (SURVEY.md 8-D) / average launch duration measured live with HIP events on the launching stream; frac = "
                        "the larger of hbm_frac and flops_frac; `traffic` (FETCH_SIZE + WRITE_SIZE per launch) and traffic_ratio = traffic / "
                        "algorithmic bytes are PMC values of the committed pass " + (pmc_src or "profiles/ (not available for this batch size)") +
                        ", not of this run")
    pipeline_gbs = 12.0e6 * fe_value / world / 1e9                     # 12.0 MB algorithmic bytes per stereo pair

    # ---------------- global BA (C4 shape), landmark-sharded over the GPUs through RCCL inside the library ----------------
    # 500 keyframes on a loop, 10 000 landmarks PER GPU (6 observations each): at 8 GPUs this is BASELINE configs[3]
    # exactly (80 000 landmarks, 480 000 edges) -- weak scaling; at N = 1 the full configs[3] is timed as well (the
    # strong-scaling reference point).
    comm_kwargs = {}
    hook_kwargs = {}
    comm_kind = "none"
    if world > 1:
        from ssvio_amd import dist_ba
        if test_mode:
            hook_kwargs = dict(allreduce=dist_ba.make_allreduce_hook_host_staged(dev), rank=rank, world_size=world)
            comm_kind = "gloo, host staged (single-GPU test mode)"
        else:
            # ncclCommInitRank inside libssx.so, id broadcast by torch.distributed.  If the library cannot bring RCCL up on
            # this node (every rank agrees on that through one all-reduce), the collective goes through torch.distributed's
            # own RCCL communicator instead (ssx_allreduce_fn callback) and the line says so.
            comm = None
            try:
                comm = dist_ba.init_native_comm(ctx, rank, world)
                ok_native = 1.0
            except Exception as exc:                                   # noqa: BLE001 -- any failure means "use the fallback"
                print(f"[bench] rank {rank}: native RCCL unavailable ({exc}); falling back to torch.distributed", file=sys.stderr)
                ok_native = 0.0
            tn = torch.tensor([ok_native], dtype=torch.float64, device=dev)
            dist.all_reduce(tn, op=dist.ReduceOp.MIN)
            if float(tn.item()) > 0.5:
                comm_kwargs = dict(comm=comm, rank=rank, world_size=world)
                comm_kind = "RCCL inside libssx.so (ssx_comm_*)"
            else:
                if comm is not None:
                    dist_ba.destroy_native_comm(ctx, comm)
                hook_kwargs = dict(allreduce=dist_ba.make_allreduce_hook(dev), rank=rank, world_size=world)
                comm_kind = "torch.distributed all_reduce through the ssx_allreduce_fn callback (native RCCL init failed)"
    dkw = dict(comm_kwargs, **hook_kwargs)

    def c4_problem(n_landmarks):
        # the generator is a Python loop over every observation (20 s at 480 k edges): keep the arrays between runs
        path = f"/tmp/ssx_c4_{n_landmarks}.npz"
        keys = ("poses", "points", "pose_fixed", "point_fixed", "edge_pose", "edge_point", "edge_uv", "edge_cam", "K", "cam_ext")
        try:
            z = np.load(path)
            pr4 = {k: z[k] for k in keys}
        except (OSError, KeyError, ValueError):
            pr4 = make_ba_problem(P=500, L=n_landmarks, obs_per_lm=6, seed=4, loop=True, fix_first_pose=True)
            try:
                np.savez(path + f".{os.getpid()}.tmp.npz", **{k: pr4[k] for k in keys})
                os.replace(path + f".{os.getpid()}.tmp.npz", path)
            except OSError:
                pass
        pr4 = dict(pr4, P=500, L=n_landmarks, E=len(pr4["edge_pose"]))
        return pr4

    def time_c4(n_landmarks, reps):
        pr4 = c4_problem(n_landmarks)
        if world > 1:
            from ssvio_amd import dist_ba
            pr4_local = dist_ba.shard_problem(pr4, rank, world)
        else:
            pr4_local = pr4
        with torch.cuda.stream(stream):
            r4 = ba.ba_solve(ctx, pr4_local, outer_rounds=1, iters=10, want_edges=False, **dkw)
            barrier()
            tb = time.perf_counter()
            n_it4 = 0
            for _ in range(reps):
                r4 = ba.ba_solve(ctx, pr4_local, outer_rounds=1, iters=10, want_edges=False, **dkw)
                n_it4 += r4["n_iters"]
            barrier()
            el = max_over_ranks(time.perf_counter() - tb)
            # the slope: the same solve with 20 iterations; (t20 - t10) / (trials20 - trials10) is what one more LM trial costs once
            # the call's fixed part (host marshalling of the observations, upload, pair lists, download) is paid
            el20 = None
            for _ in range(2):                                           # (the first 20-iteration call of a context grows its buffers)
                barrier()
                tb = time.perf_counter()
                r20 = ba.ba_solve(ctx, pr4_local, outer_rounds=1, iters=20, want_edges=False, **dkw)
                barrier()
                el20 = max_over_ranks(time.perf_counter() - tb)
            tr10, tr20 = int(np.sum(r4["trials"])), int(np.sum(r20["trials"]))
            steady = (el20 - el / reps) / max(tr20 - tr10, 1) * 1e3
            # one more solve with every launch and every all-reduce between HIP events (ssx_ba_options.collect_stats): where an
            # LM iteration's GPU time goes -- landmark-sharded kernels, the reduced solve every rank repeats, reductions, collectives
            rs = ba.ba_solve(ctx, pr4_local, outer_rounds=1, iters=10, want_edges=False, collect_stats=True, **dkw)
            ph = rs.get("phase_ms") or {}
            nit = max(int(rs["n_iters"]), 1)
            split = {"sharded_by_landmark": round((ph.get("linearize", 0.0) + ph.get("schur", 0.0) + ph.get("update", 0.0)) / nit, 4),
                     "replicated_reduced_solve": round(ph.get("linear_solution", 0.0) / nit, 4),
                     "reductions": round(ph.get("reduce", 0.0) / nit, 4), "collective": round(ph.get("comm", 0.0) / nit, 4),
                     "what": "GPU ms per LM iteration of this rank, every launch / all-reduce between HIP events (a profiled solve: "
                             "the events add launch gaps, the sum exceeds the unprofiled ms_per_lm_iteration)"}
        return {"landmarks": n_landmarks, "edges": int(pr4["E"]), "iters_per_s": round(n_it4 / el, 2),
                "edge_iters_per_s": round(float(pr4["E"]) * n_it4 / el, 1), "ms_per_lm_iteration": round(el / max(n_it4, 1) * 1e3, 3),
                "lm_trials": int(np.sum(r4["trials"])), "chi2_first_last": [float(r4["chi2"][0]), float(r4["chi2"][-1])],
                "ms_per_lm_trial_steady_state": round(steady, 3), "fixed_ms_per_call": round(el / reps * 1e3 - steady * tr10, 3),
                "steady_state_what": "(wall of a 20-iteration call - wall of a 10-iteration call) / (their LM trials' difference); "
                                     "fixed = the 10-iteration wall minus its trials at that rate (marshalling, upload, pair lists, download)",
                "phase_ms_per_iteration": split}

    C4_LM_PER_GPU = 10000
    if args.lean:
        def time_c4(n_landmarks, reps):                               # noqa: F811 -- not part of the profiled run
            return {"skipped": "--lean", "ms_per_lm_iteration": None}
    c4 = {"workload": "C4 shape: 500 KF on a loop, 6 observations per landmark, pose 0 fixed, analytic Jacobians, f64",
          "weak": time_c4(C4_LM_PER_GPU * world, 2),
          "sharding": (f"landmarks l mod {world} + RCCL all-reduce of the banded reduced system" if world > 1 else "none"),
          "collective": comm_kind}
    c4["full_configs3"] = time_c4(80000, 2)                           # strong-scaling point: the same 480 k edges at every N
    # Strong-scaling projection from THIS run's phases (GPU ms per LM iteration of the full configs[3] solve, profiled): the
    # landmark-sharded kernels divide by the rank count, the reduced solve every rank repeats and the reductions do not, the
    # collectives are this run's when it had any, else two latency-bound all-reduces of < 1 MB over xGMI (2 x 25 us).
    ph4 = c4["full_configs3"].get("phase_ms_per_iteration") or {}
    if ph4:
        t_sh, t_rep, t_red, t_col = (float(ph4.get(k, 0.0)) for k in ("sharded_by_landmark", "replicated_reduced_solve", "reductions", "collective"))
        t_sh1 = t_sh * world                                           # what one GPU would spend on all landmarks
        col8 = t_col if world > 1 else 0.05
        t1 = t_sh1 + t_rep + t_red
        t8 = t_sh1 / 8.0 + t_rep + t_red + col8
        c4["strong_scaling"] = {"target_at_8_gpus": 3.5,
                                "this_run": {"n_gpus": world, "ms_per_lm_iteration": c4["full_configs3"]["ms_per_lm_iteration"]},
                                "phases_one_gpu_ms": {"sharded_by_landmark": round(t_sh1, 4), "replicated_reduced_solve": round(t_rep, 4),
                                                      "reductions": round(t_red, 4)},
                                "assumed_collective_ms_at_8": round(col8, 4),
                                "projection_at_8_gpus": round(t1 / t8, 2) if t8 > 0 else None,
                                "amdahl_limit": round(t1 / (t_rep + t_red), 2) if t_rep + t_red > 0 else None,
                                "formula": "(sharded + replicated + reductions) / (sharded / 8 + replicated + reductions + collective), all from "
                                           "phase_ms_per_iteration of full_configs3 in this run"}
    else:
        c4["strong_scaling"] = {"target_at_8_gpus": 3.5, "projection_at_8_gpus": None}
    if comm_kwargs:
        try:
            from ssvio_amd import dist_ba
            c4["rccl_rank_world"] = list(dist_ba.native_comm_info(ctx, comm_kwargs["comm"]))
        except Exception as exc:                                       # noqa: BLE001
            c4["rccl_rank_world"] = f"unavailable: {exc}"
    c4["allreduces_per_lm_trial"] = "2 ([band | rhs | pose blocks | chi2] and [chi2', scale, outliers]); 3 on the first trial of an optimize()"

    # ---------------- configs[0] at its stated size: 200 KITTI-00-shaped pairs through the headless test_system ----------------
    # /root/reference/test/test_system.cpp:28-50 + config/kitti_00.yaml: load a KITTI-layout sequence, System::RunStep per pair
    # (LK tracking, pose-only LM, keyframes: detection + stereo LK + triangulation + the local BA on the resident window), save
    # the keyframe trajectory.  Real KITTI is not available offline: the drive is the rendered corridor of tools.synth
    # (forward motion 0.8 m per frame, exact stereo geometry, ground-truth poses).  A single live stream is latency-bound --
    # one pair at a time, a handful of small dependent launches per frame -- so this is a LATENCY figure next to the
    # throughput headline; `eight_streams` is the same loop eight times side by side on this one GPU (configs[4]'s shape).
    c1 = None
    if rank == 0 and world == 1 and args.c1_frames > 0 and not args.lean:
        try:
            import re
            import subprocess
            from ssvio_amd import build as sb
            from tools.synth import write_settings
            if c1_gen is not None and c1_gen.returncode != 0:
                raise RuntimeError("sequence generator failed: " + (c1_gen_err or b"").decode()[-300:])
            _, host_exe = sb.build_host()
            centres = np.load(os.path.join(c1_dir, "centres.npy"))

            def run_kitti(tag, overrides, extra=()):
                cfg = write_settings(os.path.join(c1_dir, f"cfg_{tag}.yaml"), overrides)
                traj = os.path.join(c1_dir, f"traj_{tag}.txt")
                r = subprocess.run([host_exe, f"--config_yaml_path={cfg}", f"--kitti_dataset_path={c1_dir}", f"--trajectory={traj}", f"--device={dev_index}", *extra],
                                   capture_output=True, text=True, timeout=1200)
                if r.returncode != 0:
                    raise RuntimeError(f"ssx_run_kitti [{tag}] failed: " + (r.stdout + r.stderr)[-400:])
                return r.stdout, traj

            def ape(traj):
                tum = np.loadtxt(traj, ndmin=2)
                idx = np.rint(tum[:, 0] / 0.1).astype(int)
                d = (tum[:, 1:4] - tum[0, 1:4]) - (centres[idx] - centres[idx[0]])
                e = np.linalg.norm(d, axis=1)
                return {"keyframes": int(len(tum)), "rmse_m": round(float(np.sqrt((e ** 2).mean())), 4), "max_m": round(float(e.max()), 4),
                        "path_length_m": round(float(np.linalg.norm(np.diff(centres, axis=0), axis=1).sum()), 1)}

            def parse(out):
                m = re.search(r"RunStep ([0-9.]+) ms/frame \(([0-9.]+) frames/s\); waiting for decoded images ([0-9.]+) ms/frame; whole loop ([0-9.]+) frames/s", out)
                st = re.search(r"frames (\d+)  keyframes (\d+)  map points (\d+)  final status (\w+)", out)
                bw = re.search(r"local BA: (\d+) windows, (\d+) LM iterations, (\d+) edges, (\d+) outlier edges", out)
                kf = re.search(r"keyframe insert \+ BA\s+(\d+) calls\s+([0-9.]+) ms/call", out)
                wu = re.search(r"warm-up \(.*?\): ([0-9.]+) ms", out)
                return {"warmup_ms_outside_the_frame_loop": float(wu.group(1)) if wu else None, "frames_per_s_runstep_only": float(m.group(2)), "ms_per_frame_runstep": float(m.group(1)),
                        "frames_per_s_incl_png_decode": float(m.group(4)), "ms_per_frame_waiting_for_decode": float(m.group(3)),
                        "frames": int(st.group(1)), "keyframes": int(st.group(2)), "map_points": int(st.group(3)), "final_status": st.group(4),
                        "ba_windows": int(bw.group(1)), "ba_lm_iterations": int(bw.group(2)), "ba_edges": int(bw.group(3)), "ba_outlier_edges": int(bw.group(4)),
                        "ms_per_keyframe_insert_and_ba": float(kf.group(2)) if kf else None}

            run_kitti("warm", {}, ("--max_frames=20",))                 # (first HIP use of the process image: page-in, code objects)
            out_w, traj_w = run_kitti("window", {"Backend.Window": 1}, ("--decode_threads=24",))
            out_m, traj_m = run_kitti("marshal", {"Backend.Window": 0}, ("--decode_threads=24",))
            c1 = {"workload": f"configs[0] shape: the first {args.c1_frames} pairs of a KITTI-00-shaped drive (synthetic corridor, 1241x376, PNG files in "
                              "KITTI layout) through ssx_run_kitti = the reference's test_system without the viewer, kitti_00.yaml settings, one stream; "
                              "System::Warmup (a synthetic keyframe + window through every compute call: kernels loaded, workspaces sized) runs before "
                              "the first frame and is reported beside the loop, not inside it",
                  "resident_window": dict(parse(out_w), ape_vs_ground_truth=ape(traj_w)),
                  "remarshalled_window": dict(parse(out_m), ape_vs_ground_truth=ape(traj_m)),
                  "trajectories_identical": open(traj_w).read() == open(traj_m).read(),
                  "value": parse(out_w)["frames_per_s_incl_png_decode"], "unit": "stereo frames/s (one live stream, PNG decoding on 24 host threads included)"}
            out_8, _ = run_kitti("streams8", {}, ("--streams=8", "--preload=1"))
            m8 = re.search(r"8 streams: aggregate ([0-9.]+) frames/s", out_8)
            c1["eight_streams_one_gpu"] = {"aggregate_frames_per_s": float(m8.group(1)) if m8 else None,
                                           "what": "--streams=8 --preload=1: eight independent copies of the loop in one process on this GPU (configs[4]'s "
                                                   "shape; one stream per GPU is the driver's multi-GPU run), frames decoded beforehand"}
            # configs[4] on ONE GPU, closed loop: S streams of the same drive, their per-frame LK / pose-only calls and their window
            # optimisations issued as BATCHED library calls (ssvio_amd/host/stream_batcher.hpp); every stream's trajectory is
            # byte-identical to the single-stream run's (checked here for every stream of every S)
            c5 = {"what": "ssx_run_kitti --streams=S --batched=C --preload=1 (C cohorts of streams, each batched on its own): S concurrent streams (full front-end + backend each) on one GPU; "
                          "frames_per_s = all streams' frames / (common start -> last stream's end); frames decoded beforehand",
                  "frames_per_stream": args.c1_frames, "runs": {}}
            ref_traj = open(traj_w).read()
            for S_ in (8, 32, 64, 128):
                cohorts = 2 if S_ >= 64 else 1                           # (two cohorts: one's images cross PCIe while the other's kernels run)
                out_b, traj_b = run_kitti(f"batched{S_}", {}, (f"--streams={S_}", "--preload=1", f"--batched={cohorts}"))
                mb = re.search(r"streams \(batched\): (\d+) frames in ([0-9.]+) s from the common start to the last stream's end = ([0-9.]+) frames/s", out_b)
                mc = re.search(r"batched calls: LK (\d+) \(([\d.]+) jobs each\), pose-only (\d+) \(([\d.]+)\), window solves (\d+) \(([\d.]+)\)", out_b)
                same = all(os.path.exists(f"{traj_b}.{k}") and open(f"{traj_b}.{k}").read() == ref_traj for k in range(S_))
                c5["runs"][str(S_)] = {"frames_per_s": float(mb.group(3)) if mb else None, "seconds": float(mb.group(2)) if mb else None, "cohorts": cohorts,
                                        "jobs_per_call": {"lk": float(mc.group(2)), "pose_only": float(mc.group(4)), "window_solve": float(mc.group(6))} if mc else None,
                                        "every_stream_byte_identical_to_the_single_stream_run": same}
            # the hard drive: the closed loop with a backend that works (>= 40 windows of >= 4000 edges in 240 frames)
            if os.path.exists(os.path.join(hard_dir, "times.txt")):
                cfg_h = write_settings(os.path.join(hard_dir, "cfg_hard.yaml"), {"ORBextractor.nInitFeatures": 500, "ORBextractor.nNewFeatures": 500,
                                                                                  "numFeatures.trackingGood": 450, "Map.ActiveMap.Size": 12})
                rh = subprocess.run([host_exe, f"--config_yaml_path={cfg_h}", f"--kitti_dataset_path={hard_dir}", f"--trajectory={hard_dir}/traj.txt", f"--device={dev_index}",
                                     "--decode_threads=24"], capture_output=True, text=True, timeout=1200)
                if rh.returncode == 0:
                    ph = parse(rh.stdout)
                    c1["hard_drive"] = dict(ph, edges_per_window=round(ph["ba_edges"] / max(ph["ba_windows"], 1), 1),
                                            what="the corridor at 1.6 m per frame with a 1 m sway, 500 features per keyframe, a keyframe below 450 tracked "
                                                 "features (Map.ActiveMap.Size 12): the closed loop of configs[0] with windows of configs[2]'s size")
                else:
                    c1["hard_drive"] = {"skipped": (rh.stdout + rh.stderr)[-300:]}
            best = max((v["frames_per_s"] or 0.0) for v in c5["runs"].values())
            c5["value"] = best
            c5["unit"] = "closed-loop stereo frames/s on one GPU (best S)"
            c5["unbatched_eight_streams"] = c1["eight_streams_one_gpu"]["aggregate_frames_per_s"]
            c1["c5_batched_streams"] = c5
        except Exception as exc:                                       # noqa: BLE001 -- an extra leg, never fatal
            print(f"[bench] configs[0] leg skipped: {exc!r}", file=sys.stderr)
            c1 = {"skipped": repr(exc)[:300]} if c1 is None or "resident_window" not in c1 else dict(c1, c5_error=repr(exc)[:300])

    # ---------------- SURVEY.md 8-F's "next" rows + A11, each with its own line (tools/bench_next.py) ----------------
    next_rows = None
    if rank == 0 and world == 1 and not args.lean:
        try:
            from tools import bench_next
            next_rows = bench_next.next_rows(ssvio_amd, ctx_ba, cpu=not args.no_cpu_baseline)
        except Exception as exc:                                       # noqa: BLE001 -- extra lines, never fatal
            print(f"[bench] next_rows skipped: {exc!r}", file=sys.stderr)
            next_rows = {"skipped": repr(exc)[:300]}

    # ---------------- CPU baseline (rank 0, N == 1 only): the same composite on one host core ----------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.lean:
        from oracle import pyoracle as po
        po.build()
        ns = max(1, min(args.cpu_sample, B))
        use_ref = po.have_ref() and os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libssvio_ref.so"))
        t_fe = t_ba = 0.0
        n_it_cpu = 0
        for i in range(ns):
            tc = time.perf_counter()
            L, R = host[i, 0], host[i, 1]
            kL, dL = po.orb_extract(L); kR, dR = po.orb_extract(R)
            idx, _ = po.stereo_match(kL, dL, kR, dR)
            m = idx >= 0
            uvL = np.stack([kL["x"][m], kL["y"][m]], 1).astype(np.float64)
            uvR = np.stack([kR["x"][idx[m]], kR["y"][idx[m]]], 1).astype(np.float64)
            po.triangulate(uvL, uvR, (718.856, 718.856, 607.1928, 185.2157), 386.1448 / 718.856)
            t_fe += time.perf_counter() - tc
            tc = time.perf_counter()
            rr = po.ba_solve(step_windows[i], "ref") if use_ref else po.ba_solve(step_windows[i], "oracle", jac_mode=1)
            n_it_cpu += len(rr["chi2"])
            t_ba += time.perf_counter() - tc
        cpu = {"value": round(ns / (t_fe + t_ba), 4), "unit": "stereo frames/s", "cores": 1,
               "kind": "reference BA + restated front-end" if use_ref else "port",
               "sample": f"{ns} of the benchmark's stereo pairs + {ns} of its C3 windows, one after the other on one thread: front-end through the "
                         f"CPU oracle (scalar C++ restatement; the reference's OpenCV front-end cannot be built: OpenCV absent), local BA through "
                         + ("the reference's own g2o + CSparse + numeric Jacobians (oracle/_ref/libssvio_ref.so)" if use_ref else "the oracle port (numeric Jacobians)"),
               "frontend_frames_per_s": round(ns / t_fe, 3), "ba_windows_per_s": round(ns / t_ba, 3),
               "ba_lm_iterations_per_s": round(n_it_cpu / t_ba, 2),
               "host_cores_available": os.cpu_count()}
        # the front-end port on many cores at once (one process per core, its own pairs): the path is embarrassingly
        # parallel over frames, so this is what the host CPUs could do with the scalar restatement
        try:
            from oracle import pool_worker
            nw = max(1, min(os.cpu_count() or 1, 64))
            rate, reported = pool_worker.frontend_all_cores(nw, pairs_per_worker=2)
            if reported >= max(1, nw // 2):
                cpu["frontend_all_cores"] = {"value": round(rate, 2), "unit": "stereo frames/s", "cores": reported,
                                             "sample": f"{reported} processes x 2 pairs each, started together (one per core)"}
        except Exception as e:                                       # the baseline is informative, never fatal
            cpu["all_cores_error"] = str(e)[:200]
        # configs[0] on the CPU: the same host loop (tests/host/oracle_runner.cpp: the product's host layer with the CPU oracle behind
        # its Compute interface) over the same 200 pairs, one core; images decoded outside its clock
        if c1 and "skipped" not in c1:
            try:
                import subprocess
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                import host_util
                from tools.synth import write_settings
                runner = host_util.build_test_binaries()["oracle_runner"]
                cfg_c = write_settings(os.path.join(c1_dir, "cfg_cpu.yaml"), {})
                tcpu = time.perf_counter()
                rc = subprocess.run([runner, cfg_c, c1_dir, os.path.join(c1_dir, "traj_cpu.txt")], capture_output=True, text=True, timeout=1800)
                wall = time.perf_counter() - tcpu
                if rc.returncode != 0:
                    raise RuntimeError(rc.stderr[-300:])
                import re
                mm = re.search(r"runstep_seconds first ([0-9.]+) rest ([0-9.]+)", rc.stdout)
                t_all = float(mm.group(1)) + float(mm.group(2))
                c1["cpu_oracle_runner"] = {"frames_per_s_runstep_only": round(args.c1_frames / t_all, 2), "cores": 1, "seconds_wall_incl_decode": round(wall, 2),
                                           "what": "the same 200 pairs and settings through the same host loop with the CPU oracle as Compute (scalar C++ "
                                                   "restatement of detection / LK / pose-only / BA; not the reference's OpenCV + g2o build, which cannot be built here)"}
                c1["speedup_vs_cpu_oracle_runstep"] = round(c1["resident_window"]["frames_per_s_runstep_only"] / c1["cpu_oracle_runner"]["frames_per_s_runstep_only"], 2)
            except Exception as e:                                   # noqa: BLE001
                c1["cpu_oracle_runner"] = {"skipped": str(e)[:200]}

    if rank == 0:
        out = {
            "metric": "stereo frames/s (ORB extract + row-band match + triangulate + one local BA per frame) on 1241x376",
            **({"profiling_mode": "--profile-kernels: the live-backend region was left out; value = frozen_batch"} if args.profile_kernels else {}),
            "value": round(value, 2), "unit": "stereo frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8 (front-end) + f64 (BA)", "data": "synthetic",
            "config": {"workload": "C2 + C3: 1241x376 synthetic stereo, 2000 ORB feats/img, 8 levels, extract+match+triangulate, then one "
                                   "local BA per pair on a live sliding window (10 KF x 20000 edges, one keyframe replaced per step, <= 5 x "
                                   "optimize(10), analytic Jacobians)",
                       "pairs_per_step_per_gpu": B, "ba_windows_per_step_per_gpu": B, "ba_groups": ba_groups,
                       "parallelism": f"replicas x{world}" if world > 1 else "single GPU",
                       "avg_keypoints_per_image": round(kp_total / I, 1),
                       "avg_matches_per_pair": round(float(counts[:, 2].mean()), 1),
                       "avg_triangulated_per_pair": round(float(counts[:, 3].mean()), 1),
                       "lm_iterations_per_window": round(lm_iters / (args.steps * B), 2),
                       "images": "host -> device every step (pinned ring of 2 batches; ssx_stereo_batch_upload of step k + 1's batch on the library's "
                                 "copy stream beside step k's kernels, ssx_stereo_batch_run on the batch uploaded during the step before)",
                       "h2d_bytes_per_step_per_gpu": h2d_bytes_per_step,
                       "downloaded_every_step": "keypoint / match / triangulation counts of every pair (collected one step behind: the batch of step "
                                                "k - 1 while batch k is in the front-end) + the optimised keyframe poses AND landmarks of every window",
                       "windows": "B live sliding windows (ssx_ba_window) resident in HBM: per step every window pops its oldest keyframe and pushes a "
                                  "new one by landmark slots (pose, ~400 landmarks, 2000 observations -- one ssx_ba_window_update_batch call per group), "
                                  "then ssx_ba_window_solve_batch optimises the group's windows where they lie (backend.cpp:88-169, map.cpp:27-56, 89-160)",
                       "window": live_info["window"], "backend_groups": live_info["host_threads"]},
            "live_backend": live_info,
            "c5": (c1 or {}).get("c5_batched_streams") if isinstance(c1, dict) else None,   # configs[4] on one GPU: S batched streams, closed loop
            "nccl_ranks": (c4.get("rccl_rank_world") if world > 1 else [0, 1]),    # (rank, ranks) of the RCCL communicator inside libssx.so: ssx_comm_info
            "frozen_batch": {"value": round(frozen_value, 2), "unit": "stereo frames/s", "ms_per_step": round(frozen_elapsed / FROZEN_STEPS * 1e3, 4),
                             "what": "rounds 2-4's headline: the same front-end step (host images in, counts out) beside a FROZEN batch of B C3 windows "
                                     "(ssx_ba_batch: marshalled + uploaded before the clock starts, re-solved from the same state every step, poses "
                                     "downloaded) -- no keyframe enters or leaves a window"},
            "resident": {"value": round(resident_value, 2), "unit": "stereo frames/s", "ms_per_step": round(resident_elapsed / RES_STEPS * 1e3, 4),
                         "what": "rounds 1-3's headline: the frozen batch with the images resident in HBM too and nothing downloaded"},
            "roofline_frac": roofline["frac"], "roofline_hbm_frac": roofline["hbm_frac"], "roofline_flops_frac": roofline["flops_frac"],
            "roofline_traffic_ratio": roofline["traffic_ratio"],
            "c1": c1,
            "host_buffers_inclusive": {"value": None if host_value != host_value else round(host_value, 2), "unit": "stereo frames/s",
                                       "what": "the same step with the B windows handed over as host arrays every step (ssx_ba_solve_batch: host "
                                               "marshalling on 16 threads + one PCIe upload + one download of poses / points); images still resident",
                                       "two_batches_in_flight": None if pipe_value is None else round(pipe_value, 2)},
            "roofline": roofline,
            "frontend": {"metric": "stereo frames/s (ORB extract + row-band match + triangulate), no BA", "value": round(fe_value, 2),
                         "ms_per_step": round(fe_elapsed / args.steps * 1e3, 4),
                         "pipeline_algorithmic_GBps_per_gpu": round(pipeline_gbs, 3),
                         "single_pair_latency": lat,
                         "kitti_like_density": fe_kitti},
            "kernels": {k: {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items()} for k, v in sorted(kernels.items())},
            "ba": {"workload": "C3: local BA, 10 KF x 4000 landmarks x 20000 edges, analytic Jacobians, f64",
                   "batched": {"windows_per_call": B, "windows_per_s": round(world * B * BA_REP / bab_elapsed, 1),
                               "iters_per_s": round(world * n_it_b / bab_elapsed, 1), "ms_per_call": round(bab_elapsed / BA_REP * 1e3, 3)},
                   "one_window": {"iters_per_s": round(n_it_1 / ba1_elapsed, 1), "ms_per_solve": round(ba1_elapsed / ONE_REP * 1e3, 3),
                                  "lm_iterations_per_solve": n_it_1 // ONE_REP},
                   "note": "batched = resident windows (ssx_ba_batch_solve, no PCIe traffic but the LM control words); one_window = ssx_ba_solve "
                           "incl. host marshalling and host<->device transfer of the window and its result"},
            "ba_c4": c4,
            "next_rows": next_rows,
            "cpu_baseline": cpu,
        }
        if cpu:
            out["speedup_vs_cpu_1core"] = {"ratio": round(value / cpu["value"], 1),
                                           "caveat": "against ONE core running a SCALAR restatement of the front-end (no SIMD; OpenCV's FAST / ORB "
                                                     "would be several times faster) plus the reference's own g2o for the BA: a reported baseline, "
                                                     "not a measure of kernel quality -- the roofline fractions are"}
        print(json.dumps(out))
    batch.close()                        # resident batches own device memory of their ctx: destroy them first
    batch_host.close()
    ctx.close()
    ctx_ba.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
