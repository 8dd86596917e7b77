"""Landmark-sharded multi-GPU bundle adjustment -- host-side plumbing.

One process per GPU under torch.distributed (backend "nccl" = RCCL over xGMI on ROCm).  Rank r owns the
landmarks {l : l mod world == r} and ALL edges of those landmarks; every pose is replicated.  libssx.so sums three
small buffers across ranks through the ssx_allreduce_fn hook (include/ssx.h); `make_allreduce_hook` wires that hook
to torch.distributed.all_reduce on a zero-copy tensor view of the device buffer.  The ctx must run on torch's
current stream (ssvio_amd.Context(device, stream=torch.cuda.current_stream().cuda_stream) inside a
`with torch.cuda.stream(s)` block) so that the collective is ordered with the kernels.
"""
from __future__ import annotations

import numpy as np


def shard_problem(pr: dict, rank: int, world: int) -> dict:
    """The shard of `pr` (tools.synth.make_ba_problem layout) owned by `rank`: landmarks l % world == rank,
    re-indexed compactly, with all their edges; poses unchanged.  `lm_global` maps local -> global landmark ids."""
    lm_global = np.nonzero(np.arange(pr["points"].shape[0]) % world == rank)[0]
    remap = -np.ones(pr["points"].shape[0], dtype=np.int64)
    remap[lm_global] = np.arange(len(lm_global))
    keep = remap[pr["edge_point"]] >= 0
    out = dict(pr)
    out["L"] = int(len(lm_global))
    out["points"] = np.ascontiguousarray(pr["points"][lm_global])
    out["point_fixed"] = None if pr.get("point_fixed") is None else np.ascontiguousarray(pr["point_fixed"][lm_global])
    out["edge_pose"] = np.ascontiguousarray(pr["edge_pose"][keep])
    out["edge_point"] = np.ascontiguousarray(remap[pr["edge_point"][keep]].astype(np.int32))
    out["edge_uv"] = np.ascontiguousarray(pr["edge_uv"][keep])
    out["edge_cam"] = None if pr.get("edge_cam") is None else np.ascontiguousarray(pr["edge_cam"][keep])
    out["E"] = int(keep.sum())
    out["lm_global"] = lm_global
    out["edge_global"] = np.nonzero(keep)[0]
    return out


def make_allreduce_hook(device, group=None):
    """-> python callable (user, buf_dev, count, stream) -> 0 that sum-all-reduces `count` doubles in place."""
    import torch
    import torch.distributed as dist

    class _Dev:   # zero-copy view of a raw device pointer
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (int(ptr), False), "version": 2}

    def hook(user, buf, count, stream):
        try:
            t = torch.as_tensor(_Dev(buf, count), device=device)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            return 0
        except Exception as e:   # never let an exception cross the C boundary
            print("ssvio_amd.dist_ba: all_reduce failed:", e, flush=True)
            return 1

    return hook


def make_allreduce_hook_host_staged(device, group=None):
    """Same contract as make_allreduce_hook for process groups WITHOUT device collectives (gloo): the buffer is
    staged through host memory.  Slow by construction; it exists so that the multi-rank control flow of
    ssx_ba_solve (identical LM decisions on every rank, tile-pattern exchange, packed tile all-reduce) can be tested
    with several processes on ONE GPU."""
    import torch
    import torch.distributed as dist

    class _Dev:
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (int(ptr), False), "version": 2}

    def hook(user, buf, count, stream):
        try:
            t = torch.as_tensor(_Dev(buf, count), device=device)
            h = t.cpu()                                   # ordered on the current stream (= the ctx stream), then waits
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
            t.copy_(h)
            return 0
        except Exception as e:
            print("ssvio_amd.dist_ba: host-staged all_reduce failed:", e, flush=True)
            return 1

    return hook


def init_native_comm(ctx, rank: int, world: int, group=None):
    """RCCL inside libssx.so (ssx_comm_*, include/ssx.h): rank 0 draws the ncclUniqueId, torch.distributed only carries its
    128 bytes to the other ranks, every rank then creates the communicator on its ctx's device.  -> ssx_comm handle for
    ba.ba_solve(..., comm=handle).  Works for world == 1 too (a one-rank communicator: the plumbing test)."""
    import ctypes as C

    import torch
    import torch.distributed as dist
    lib = ctx.lib
    ident = (C.c_char * 128)()
    failed = None
    if rank == 0:
        try:
            ctx.check(lib.ssx_comm_unique_id(ctx.handle, ident))
        except Exception as exc:                                       # noqa: BLE001 -- the other ranks are waiting in the broadcast below:
            failed = exc                                               # they get an all-zero id and fail the same way (no rank is left behind)
            ident = (C.c_char * 128)()
    if world > 1:
        t = torch.frombuffer(bytearray(bytes(ident)), dtype=torch.uint8).clone()
        backend = dist.get_backend(group)
        if backend == "nccl":
            t = t.to(torch.device("cuda", ctx.device))
        dist.broadcast(t, src=0, group=group)
        ident = (C.c_char * 128).from_buffer_copy(bytes(t.cpu().numpy().tobytes()))
    if failed is not None or not any(bytes(ident)):
        raise RuntimeError(f"ssx_comm: rank 0 could not draw an ncclUniqueId ({failed})" if failed is not None else
                           "ssx_comm: rank 0 could not draw an ncclUniqueId (all-zero id received)")
    h = C.c_void_p()
    ctx.check(lib.ssx_comm_init(ctx.handle, ident, int(rank), int(world), C.byref(h)))
    return h


def destroy_native_comm(ctx, comm):
    ctx.lib.ssx_comm_destroy.restype = None
    ctx.lib.ssx_comm_destroy(comm)


def native_comm_info(ctx, comm):
    """-> (rank, world_size) as the communicator inside libssx.so reports them (ssx_comm_info)"""
    import ctypes as C
    r, w = C.c_int32(-1), C.c_int32(-1)
    ctx.lib.ssx_comm_info.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    ctx.check(ctx.lib.ssx_comm_info(comm, C.byref(r), C.byref(w)))
    return r.value, w.value
