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
