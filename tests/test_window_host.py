"""The host side of ssx_ba_window (keyframe / landmark slots, the id map, dead observation blocks, the storage rewrite) without
a GPU: ssx_ba_window_selftest drives random pushes (by id and by slot), pops of arbitrary keyframes and failing pushes on a
window that has no device, and checks the window's export against a plain model after every step; two twin windows receive
every edit through ssx_ba_window_update_batch (two windows per call, on the library's host threads) and must equal the window
itself, field for field; a call that names one window twice must be refused.  (The device side --
solves equal to fresh ssx_ba_solve calls, bit for bit -- is tests/test_ba_gpu.py::test_resident_window_*.)"""
import ctypes as C

import pytest

from ssvio_amd import _lib


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 7, 11, 12345])
def test_window_host_logic_against_a_model(seed):
    lib = _lib.load()
    lib.ssx_ba_window_selftest.restype = C.c_int32
    lib.ssx_ba_window_selftest.argtypes = [C.c_uint32, C.c_int32]
    assert lib.ssx_ba_window_selftest(seed, 500) == 0


def test_large_window_observation_pass_is_the_same_on_any_number_of_threads():
    """prepare() of a large window (> 16 free keyframes, >= 65 536 observations) counts its observations on the worker pool; the
    ranks inside a landmark, the offsets and the chunk cuts must be those of the one-thread pass (shuffled observation order, so
    that the ranges of different threads meet inside the landmarks)."""
    import numpy as np
    from ssvio_amd import ba
    from tools.synth import make_ba_problem
    lib = _lib.load()
    lib.ssx_ba_debug_prepare_digest.restype = C.c_uint64
    pr = make_ba_problem(P=40, L=14000, obs_per_lm=5, seed=3, loop=True, fix_first_pose=True)
    assert pr["E"] >= 1 << 16
    order = np.random.default_rng(0).permutation(pr["E"])
    for k in ("edge_pose", "edge_point", "edge_uv", "edge_cam"):
        if k in pr and pr[k] is not None:
            pr[k] = np.ascontiguousarray(pr[k][order])
    keep = []
    st = ba._problem_struct(pr, keep)
    lib.ssx_ba_debug_prepare_digest.argtypes = [C.POINTER(type(st)), C.c_int32]
    ref = lib.ssx_ba_debug_prepare_digest(C.byref(st), 1)
    assert ref != 0
    for threads in (2, 3, 8, 13):
        assert lib.ssx_ba_debug_prepare_digest(C.byref(st), threads) == ref
