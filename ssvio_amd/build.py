"""Build libssx.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

    python -m ssvio_amd.build [--force]

hipcc cross-compiles without a GPU, so this runs in the CPU-only build container; the resulting
ssvio_amd/libssx.so is git-ignored but travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libssx.so")
ARCH = "gfx950"

COMMON = ["-std=c++17", "-O3", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
          "-Wno-unused-result", "-fvisibility=hidden", "-DSSX_BUILD"]
# SSX_PRODUCT_BUILD=1: the library without the hooks of this repository's tests and tools (include/ssx_test_hooks.h); the default
# build keeps them, because the GPU tests must load the very library that ships
if os.environ.get("SSX_PRODUCT_BUILD"):
    COMMON.append("-DSSX_NO_TEST_HOOKS")
# files whose results must be bit-identical to the CPU oracle (integer / f32 image arithmetic): forbid
# fused multiply-add contraction so every float operation rounds exactly like the scalar C++ restatement
PER_FILE = {
    "orb.hip": ["-ffp-contract=off"],
    "stereo.hip": ["-ffp-contract=off"],
    "lk.hip": ["-ffp-contract=off"],
}


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libssx.so cannot be built (there is no CPU fallback)")


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _newest_dep() -> float:
    t = 0.0
    for root in (CSRC, os.path.join(HERE, "..", "include")):
        for f in os.listdir(root):
            t = max(t, os.path.getmtime(os.path.join(root, f)))
    return max(t, os.path.getmtime(os.path.abspath(__file__)))


def _compile(cc, src):
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    cmd = [cc, *COMMON, *PER_FILE.get(src, []), *os.environ.get("SSX_EXTRA_HIPCC_FLAGS", "").split(), "-c", os.path.join(CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr[-6000:]}")
    return obj, r.stderr


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= _newest_dep():
        return LIB
    cc = hipcc()
    os.makedirs(OBJ, exist_ok=True)
    srcs = sources()
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        results = list(ex.map(lambda s: _compile(cc, s), srcs))
    if verbose:
        for _, err in results:
            if err.strip():
                print(err, file=sys.stderr)
    objs = [o for o, _ in results]
    cmd = [cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
    return LIB


# ---- host layer (ssvio_amd/host): plain C++17 over the C ABI, built with g++ -------------------------------------
HOST = os.path.join(HERE, "host")
HOST_LIB = os.path.join(HOST, "libssx_host.so")
HOST_EXE = os.path.join(HOST, "ssx_run_kitti")
HOST_LIB_SRCS = ["dataset.cpp", "map.cpp", "frontend.cpp", "backend.cpp", "system.cpp", "ssx_compute.cpp", "stream_batcher.cpp"]
HOST_FLAGS = ["-std=c++17", "-O2", "-fPIC", "-Wall", "-Wno-unknown-pragmas", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include"]


def build_host(force: bool = False):
    """libssx_host.so (data model, map, front-end / backend state machines, KITTI + PNG input, Compute on libssx.so)
    and the ssx_run_kitti executable.  Returns (library, executable)."""
    lib = build(force=False)
    deps = [os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith((".cpp", ".hpp"))]
    deps += [os.path.join(CSRC, "se3.hpp"), os.path.join(HERE, "..", "include", "ssx.h"), os.path.join(HERE, "..", "include", "ssx_shim.hpp"),
             os.path.abspath(__file__), lib]
    newest = max(os.path.getmtime(d) for d in deps)
    if not force and all(os.path.exists(f) and os.path.getmtime(f) >= newest for f in (HOST_LIB, HOST_EXE)):
        return HOST_LIB, HOST_EXE
    cxx = shutil.which("g++")
    if not cxx:
        raise RuntimeError("g++ not found: the host layer cannot be built")
    rpath = ["-Wl,-rpath,$ORIGIN", "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath,/opt/rocm/lib"]
    cmd = [cxx, *HOST_FLAGS, "-shared", *[os.path.join(HOST, f) for f in HOST_LIB_SRCS], lib, "-lz", "-lpthread", *rpath, "-o", HOST_LIB]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"g++ failed on the host library:\n{r.stderr[-6000:]}")
    cmd = [cxx, *HOST_FLAGS, os.path.join(HOST, "run_kitti.cpp"), HOST_LIB, lib, "-lz", "-lpthread", *rpath, "-o", HOST_EXE]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"g++ failed on ssx_run_kitti:\n{r.stderr[-6000:]}")
    return HOST_LIB, HOST_EXE


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(*build_host(force="--force" in sys.argv))
