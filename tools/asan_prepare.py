"""AddressSanitizer over the HOST side of libssx.so's bundle-adjustment marshalling (prepare(): edge sort, chunks, packed records,
pair lists, work items), no GPU needed:
    SSX_EXTRA_HIPCC_FLAGS="-fsanitize=address -fno-omit-frame-pointer -g" python -c "from ssvio_amd import build; build.build(force=True)"
    ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so) python tools/asan_prepare.py
(rebuild without the flags afterwards)"""
import sys, ctypes as C
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssvio_amd import _lib, ba
from tools.synth import make_ba_problem
lib=_lib.load()
lib.ssx_ba_debug_prepare_seconds.restype=C.c_double
keep=[]
for kw in (dict(P=10,L=4000,obs_per_lm=5,seed=3), dict(P=12,L=1500,obs_per_lm=4,seed=5), dict(P=16,L=1200,obs_per_lm=5,seed=6,fix_first_pose=True),
           dict(P=4,L=60,obs_per_lm=4,seed=2), dict(P=30,L=1200,obs_per_lm=5,seed=28,fix_first_pose=True), dict(P=10,L=600,seed=24,frac_gross=0.45), dict(P=3, L=1, obs_per_lm=3, seed=1)):
    pr=make_ba_problem(**kw)
    st=ba._problem_struct(pr, keep)
    print(kw, "prepare: %.3f ms" % (1e3*lib.ssx_ba_debug_prepare_seconds(C.byref(st), 3)))
lib.ssx_ba_debug_upload_format.restype = C.c_int32
for kw in (dict(P=10, L=900, seed=4, uv_f32=True), dict(P=10, L=900, seed=4)):
    pr = make_ba_problem(**kw); st = ba._problem_struct(pr, keep)
    print(kw, "upload format", lib.ssx_ba_debug_upload_format(C.byref(st)))
print("done")
# round 3: the host side of ssx_ba_window (slots, id map, dead blocks, storage rewrite) against its model
lib.ssx_ba_window_selftest.restype = C.c_int32; lib.ssx_ba_window_selftest.argtypes = [C.c_uint32, C.c_int32]
for seed in (0, 5, 9):
    print("window selftest seed", seed, "->", lib.ssx_ba_window_selftest(seed, 500))
print("done (window)")
