"""Two ranks on ONE GPU (gloo process group, host-staged all-reduce hook): the multi-rank control flow of
ssx_ba_solve -- landmark shards, all-reduced pose blocks / reduced system / trial scalars, identical LM decisions on
both ranks, and for large windows the tile-pattern exchange + packed non-zero-tile all-reduce -- must reproduce the
single-rank solve up to the summation order of the two partial sums (1e-9 relative)."""
import os
import pickle
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = {
    "small": dict(cfg=dict(P=10, L=900, seed=4), kw=dict()),
    # large windows: the band solver (one workgroup / dissected into segments: the ranks all-reduce the band + rhs, then
    # every rank factors it redundantly) and the 64x64-tile solver (tile-pattern exchange + packed tile all-reduce)
    "large": dict(cfg=dict(P=40, L=1500, obs_per_lm=5, seed=9, loop=False, fix_first_pose=True), kw=dict(outer_rounds=1, iters=6)),
    "large_dissected": dict(cfg=dict(P=130, L=4000, obs_per_lm=6, seed=10, loop=True, wrap=True, fix_first_pose=True), kw=dict(outer_rounds=1, iters=5)),
    "large_tiles": dict(cfg=dict(P=40, L=1500, obs_per_lm=5, seed=9, loop=False, fix_first_pose=True), kw=dict(outer_rounds=1, iters=6, large_solver=1)),
}


def _worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    import ssvio_amd
    from ssvio_amd import ba, dist_ba
    from tools.synth import make_ba_problem
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    res = {}
    s = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s):
        ctx = ssvio_amd.Context(0, stream=s.cuda_stream)
        hook = dist_ba.make_allreduce_hook_host_staged(dev)
        for name, c in CASES.items():
            pr = make_ba_problem(**c["cfg"])
            loc = dist_ba.shard_problem(pr, rank, world)
            r = ba.ba_solve(ctx, loc, allreduce=hook, rank=rank, world_size=world, **c["kw"])
            res[name] = dict(poses=r["poses"], points=r["points"], chi2=r["chi2"], trials=r["trials"], lm_global=loc["lm_global"])
        ctx.close()
    pickle.dump(res, open(os.path.join(out_dir, f"rank{rank}.pkl"), "wb"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_match_the_single_rank_solve(ctx):
    import torch.multiprocessing as mp
    from ssvio_amd import ba
    from tools.synth import make_ba_problem
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, 29751, d), nprocs=2, join=True)
        r0 = pickle.load(open(os.path.join(d, "rank0.pkl"), "rb")); r1 = pickle.load(open(os.path.join(d, "rank1.pkl"), "rb"))
    for name, c in CASES.items():
        pr = make_ba_problem(**c["cfg"])
        one = ba.ba_solve(ctx, pr, **c["kw"])
        a, b = r0[name], r1[name]
        # both ranks took the same decisions and hold the same (replicated) poses
        assert np.array_equal(a["trials"], b["trials"]) and np.array_equal(a["chi2"], b["chi2"]) and np.array_equal(a["poses"], b["poses"])
        assert np.array_equal(a["trials"], one["trials"]), name
        np.testing.assert_allclose(a["chi2"], one["chi2"], rtol=1e-9)
        np.testing.assert_allclose(a["poses"], one["poses"], rtol=0, atol=1e-9)
        pts = np.zeros_like(one["points"])
        pts[a["lm_global"]] = a["points"]; pts[b["lm_global"]] = b["points"]
        np.testing.assert_allclose(pts, one["points"], rtol=0, atol=1e-8)


def _worker_c4(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    import ssvio_amd
    from ssvio_amd import ba, dist_ba
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    z = np.load(os.path.join(out_dir, "c4.npz"))
    pr = {k: z[k] for k in z.files}
    pr.update(P=int(pr["P"]), L=int(pr["L"]), E=int(pr["E"]))
    s = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s):
        ctx = ssvio_amd.Context(0, stream=s.cuda_stream)
        loc = dist_ba.shard_problem(pr, rank, world)
        r = ba.ba_solve(ctx, loc, allreduce=dist_ba.make_allreduce_hook_host_staged(dev), rank=rank, world_size=world, outer_rounds=1, iters=4, want_edges=False)
        ctx.close()
    pickle.dump(dict(poses=r["poses"], points=r["points"], chi2=r["chi2"], trials=r["trials"], lm_global=loc["lm_global"], n_local=int(loc["L"])),
                open(os.path.join(out_dir, f"rank{rank}.pkl"), "wb"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_full_size_c4(ctx):
    """BASELINE configs[3] AT ITS FULL SIZE (500 keyframes on a loop, 80 000 landmarks, 480 000 observations) sharded over two ranks
    that share this GPU: landmarks l mod 2, the banded reduced system + right-hand side + pose blocks all-reduced per LM trial, the
    block cyclic reduction solved redundantly by both ranks.  Both take the single-rank solve's LM decisions and end on its poses
    and landmarks (1e-9 / 1e-8: two partial sums added in another order).  The multi-GPU form of this configuration has never run on
    more than one GPU (no such node was available to any round); this is the closest a one-GPU box gets."""
    import torch.multiprocessing as mp
    from ssvio_amd import ba
    from tools.synth import make_ba_problem
    pr = make_ba_problem(P=500, L=80000, obs_per_lm=6, seed=4, loop=True, fix_first_pose=True)
    keys = ("poses", "points", "pose_fixed", "point_fixed", "edge_pose", "edge_point", "edge_uv", "edge_cam", "K", "cam_ext")
    with tempfile.TemporaryDirectory() as d:
        np.savez(os.path.join(d, "c4.npz"), P=pr["P"], L=pr["L"], E=pr["E"], **{k: pr[k] for k in keys})
        mp.spawn(_worker_c4, args=(2, 29757, d), nprocs=2, join=True)
        a = pickle.load(open(os.path.join(d, "rank0.pkl"), "rb")); b = pickle.load(open(os.path.join(d, "rank1.pkl"), "rb"))
    one = ba.ba_solve(ctx, pr, outer_rounds=1, iters=4, want_edges=False)
    assert a["n_local"] == b["n_local"] == 40000
    assert np.array_equal(a["trials"], b["trials"]) and np.array_equal(a["chi2"], b["chi2"]) and np.array_equal(a["poses"], b["poses"])
    assert np.array_equal(a["trials"], one["trials"])
    np.testing.assert_allclose(a["chi2"], one["chi2"], rtol=1e-9)
    np.testing.assert_allclose(a["poses"], one["poses"], rtol=0, atol=1e-9)
    pts = np.zeros_like(one["points"])
    pts[a["lm_global"]] = a["points"]; pts[b["lm_global"]] = b["points"]
    np.testing.assert_allclose(pts, one["points"], rtol=0, atol=1e-8)


def _worker_native(rank, world, port, out_dir):
    """one rank per GPU, RCCL inside libssx.so (ncclCommInitRank with world_size > 1 through ssx_comm_init)"""
    import torch
    import torch.distributed as dist
    import ssvio_amd
    from ssvio_amd import ba, dist_ba
    from tools.synth import make_ba_problem
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{rank}"))
    res = {}
    ctx = ssvio_amd.Context(rank)
    try:
        comm = dist_ba.init_native_comm(ctx, rank, world)
    except Exception as e:                                   # noqa: BLE001 -- librccl missing / the node refuses the communicator: reported, not a numerics failure
        pickle.dump({"skip": repr(e)}, open(os.path.join(out_dir, f"rank{rank}.pkl"), "wb"))
        dist.barrier(); dist.destroy_process_group()
        return
    res["info"] = dist_ba.native_comm_info(ctx, comm)
    for name, c in CASES.items():
        pr = make_ba_problem(**c["cfg"])
        loc = dist_ba.shard_problem(pr, rank, world)
        r = ba.ba_solve(ctx, loc, comm=comm, rank=rank, world_size=world, **c["kw"])
        res[name] = dict(poses=r["poses"], points=r["points"], chi2=r["chi2"], trials=r["trials"], lm_global=loc["lm_global"])
    dist_ba.destroy_native_comm(ctx, comm)
    ctx.close()
    pickle.dump(res, open(os.path.join(out_dir, f"rank{rank}.pkl"), "wb"))
    dist.barrier()
    dist.destroy_process_group()


def test_native_rccl_two_gpus_match_the_single_gpu_solve(ctx):
    """The first execution of ncclCommInitRank(world_size > 1) inside libssx.so that hardware allows: needs TWO GPUs in the box
    (skipped otherwise -- the builder's boxes have one).  One rank per GPU, landmark shards l mod 2, ncclAllReduce(f64, sum) on
    the ctx stream for the pose blocks / the reduced system (small window, band, tiles) / the trial scalars: both ranks take the
    single-GPU solve's LM decisions and end on its poses and points (1e-9: the two partial sums are added in another order)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs: RCCL refuses two ranks on one device")
    import torch.multiprocessing as mp
    from ssvio_amd import ba
    from tools.synth import make_ba_problem
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_native, args=(2, 29763, d), nprocs=2, join=True)
        r0 = pickle.load(open(os.path.join(d, "rank0.pkl"), "rb")); r1 = pickle.load(open(os.path.join(d, "rank1.pkl"), "rb"))
    if "skip" in r0 or "skip" in r1:
        pytest.skip("RCCL communicator could not be created on this node: " + str(r0.get("skip") or r1.get("skip")))
    assert tuple(r0["info"]) == (0, 2) and tuple(r1["info"]) == (1, 2)
    for name, c in CASES.items():
        pr = make_ba_problem(**c["cfg"])
        one = ba.ba_solve(ctx, pr, **c["kw"])
        a, b = r0[name], r1[name]
        assert np.array_equal(a["trials"], b["trials"]) and np.array_equal(a["chi2"], b["chi2"]) and np.array_equal(a["poses"], b["poses"])
        assert np.array_equal(a["trials"], one["trials"]), name
        np.testing.assert_allclose(a["chi2"], one["chi2"], rtol=1e-9)
        np.testing.assert_allclose(a["poses"], one["poses"], rtol=0, atol=1e-9)
        pts = np.zeros_like(one["points"])
        pts[a["lm_global"]] = a["points"]; pts[b["lm_global"]] = b["points"]
        np.testing.assert_allclose(pts, one["points"], rtol=0, atol=1e-8)
