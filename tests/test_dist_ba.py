"""CPU, world_size 2 over gloo: the landmark sharding that the multi-GPU BA relies on.

What ssx_ba_solve needs from the shards: (1) every landmark and every edge lives on exactly
one rank, poses everywhere; (2) the pose blocks Hpp/bp, the robust chi2 and the Schur complement S = Hpp - sum_l
W_l D_l^-1 W_l^T, b_s = bp - sum_l W_l D_l^-1 bl are ADDITIVE over landmarks, so an all-reduce(sum) of the per-rank
partial systems reproduces the single-GPU system.  Checked with the CPU oracle as the per-shard evaluator and
torch.distributed (gloo) as the collective -- the GPU path swaps in RCCL through the same hook."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _reduced_system(lin, pr, lam):
    """S (6P x 6P) and b_s from the blocks of one linearisation (oracle layout), landmarks eliminated."""
    P = pr["poses"].shape[0]
    S = np.zeros((6 * P, 6 * P)); bs = lin["bp"].reshape(-1).copy()
    for p in range(P):
        S[6 * p:6 * p + 6, 6 * p:6 * p + 6] = lin["Hpp"][p]
    by_lm = {}
    for e, (p, l) in enumerate(zip(pr["edge_pose"], pr["edge_point"])):
        by_lm.setdefault(int(l), []).append((int(p), e))
    for l, obs in by_lm.items():
        if pr["point_fixed"] is not None and pr["point_fixed"][l]:
            continue
        Dinv = np.linalg.inv(lin["Hll"][l] + lam * np.eye(3))
        for (pa, ea) in obs:
            bs[6 * pa:6 * pa + 6] -= lin["Hpl"][ea] @ Dinv @ lin["bl"][l]
            for (pb, eb) in obs:
                S[6 * pa:6 * pa + 6, 6 * pb:6 * pb + 6] -= lin["Hpl"][ea] @ Dinv @ lin["Hpl"][eb].T
    return S, bs


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import pyoracle as po
    from ssvio_amd.dist_ba import shard_problem
    from tools.synth import make_ba_problem
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    pr = make_ba_problem(P=6, L=240, obs_per_lm=4, seed=21)
    sh = shard_problem(pr, rank, world)
    # (1) partition bookkeeping
    owned = torch.zeros(pr["L"], dtype=torch.int64); owned[torch.from_numpy(sh["lm_global"])] = 1
    edges = torch.zeros(pr["E"], dtype=torch.int64); edges[torch.from_numpy(sh["edge_global"])] = 1
    dist.all_reduce(owned); dist.all_reduce(edges)
    assert bool((owned == 1).all()) and bool((edges == 1).all())
    assert np.array_equal(sh["poses"], pr["poses"])
    # (2) additivity of the partial systems
    lin = po.ba_linearize(sh, jac_mode=0)
    lam = 3.7
    S, bs = _reduced_system(lin, sh, lam)
    packed = torch.from_numpy(np.concatenate([S.ravel(), bs, lin["Hpp"].ravel(), lin["bp"].ravel(), [lin["chi2"]]]))
    dist.all_reduce(packed)
    if rank == 0:
        full = po.ba_linearize(pr, jac_mode=0)
        Sf, bsf = _reduced_system(full, pr, lam)
        ref = np.concatenate([Sf.ravel(), bsf, full["Hpp"].ravel(), full["bp"].ravel(), [full["chi2"]]])
        err = np.abs(packed.numpy() - ref).max() / np.abs(ref).max()
        q.put(float(err))
    dist.barrier()
    dist.destroy_process_group()


def test_landmark_sharding_is_additive_world2(po):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    err = q.get(timeout=10)
    assert err < 1e-12, err


def test_shard_edge_cases():
    from ssvio_amd.dist_ba import shard_problem
    from tools.synth import make_ba_problem
    pr = make_ba_problem(P=4, L=5, obs_per_lm=3, seed=1)
    tot = 0
    for r in range(8):                       # more ranks than landmarks: some shards are empty
        sh = shard_problem(pr, r, 8)
        tot += sh["E"]
        assert sh["L"] == len(sh["lm_global"]) and (sh["edge_point"] < max(sh["L"], 1)).all()
    assert tot == pr["E"]
    one = shard_problem(pr, 0, 1)            # world 1: identity
    assert one["E"] == pr["E"] and np.array_equal(one["points"], pr["points"])
