"""Helpers of the host-layer tests: build the test binaries (C++ unit checks, the oracle-backed runner), write a
settings file with the reference's keys, write a synthetic stereo sequence in KITTI layout."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "host", "build")

import sys  # noqa: E402
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from tools.synth import KITTI00_SETTINGS as DEFAULT_CONFIG  # noqa: E402  (the reference's config/kitti_00.yaml values)
from tools.synth import write_settings  # noqa: E402


def write_config(path, overrides):
    return write_settings(path, overrides)


def write_sequence(root, n_frames=12, step=0.12, seed=0, dt=0.1):
    """make_lateral_sequence as <root>/seq/{times.txt,image_0,image_1}.  Returns dir, dt, camera centres [n,3]."""
    from PIL import Image

    from tools.synth import KITTI_BASELINE, make_lateral_sequence
    frames, gt, _ = make_lateral_sequence(n_frames=n_frames, step=step, seed=seed)
    d = os.path.join(root, "seq")
    for sub in ("image_0", "image_1"):
        os.makedirs(os.path.join(d, sub), exist_ok=True)
    with open(os.path.join(d, "times.txt"), "w") as f:
        for i, (L, R) in enumerate(frames):
            f.write(f"{i * dt:e}\n")
            Image.fromarray(L).save(os.path.join(d, "image_0", f"{i:06d}.png"))
            Image.fromarray(R).save(os.path.join(d, "image_1", f"{i:06d}.png"))
    centres = np.stack([-gt[:, 4], -gt[:, 5], -gt[:, 6]], 1)          # identity rotations: centre = -t
    return dict(dir=d, dt=dt, centres=centres, step_m=step * KITTI_BASELINE, frames=frames)


def write_corridor_sequence(root, n_frames=30, step=0.8, seed=0, dt=0.1):
    """make_corridor_sequence (forward drive, KITTI-00-shaped) as <root>/seq/..."""
    from PIL import Image

    from tools.synth import make_corridor_sequence
    frames, gt, centres = make_corridor_sequence(n_frames=n_frames, step=step, seed=seed, workers=min(12, os.cpu_count() or 1))   # (same images as one worker)
    d = os.path.join(root, "seq")
    for sub in ("image_0", "image_1"):
        os.makedirs(os.path.join(d, sub), exist_ok=True)
    with open(os.path.join(d, "times.txt"), "w") as f:
        for i, (L, R) in enumerate(frames):
            f.write(f"{i * dt:e}\n")
            Image.fromarray(L).save(os.path.join(d, "image_0", f"{i:06d}.png"))
            Image.fromarray(R).save(os.path.join(d, "image_1", f"{i:06d}.png"))
    return dict(dir=d, dt=dt, centres=centres, frames=frames)


def parse_runner_log(text):
    out = []
    for line in text.splitlines():
        w = line.split()
        if len(w) >= 14 and w[0] == "frame":
            out.append(dict(frame=int(w[1]), status=int(w[3]), features=int(w[5]), keyframes=int(w[7]), points=int(w[9]),
                            active_kfs=int(w[11]), active_points=int(w[13]), centre=tuple(float(x) for x in w[15:18])))
    return out


def _build_sanitized(kind):
    """SSX_SANITIZE=asan|tsan (tools/sanitize.sh): test_units and oracle_runner with the WHOLE host layer and the CPU
    oracle compiled in from source under -fsanitize=address,undefined / thread (libssx.so stays a plain shared object:
    these tests run on the oracle, the HIP library is never called).  Separate build directory."""
    from ssvio_amd import build as b
    b.build()
    out = os.path.join(OUT, kind)
    os.makedirs(out, exist_ok=True)
    san = {"asan": ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"], "tsan": ["-fsanitize=thread"]}[kind]
    flags = [f for f in b.HOST_FLAGS if f != "-O2"] + ["-O1", "-g", "-fno-omit-frame-pointer", *san, "-I", ROOT]
    host_src = [os.path.join(b.HOST, f) for f in b.HOST_LIB_SRCS]
    orc_src = sorted(os.path.join(ROOT, "oracle", "src", f) for f in os.listdir(os.path.join(ROOT, "oracle", "src")) if f.endswith(".cpp"))
    rpath = ["-Wl,-rpath," + os.path.dirname(b.LIB), "-Wl,-rpath,/opt/rocm/lib"]
    src = os.path.join(ROOT, "tests", "host")
    units, runner = os.path.join(out, "test_units"), os.path.join(out, "oracle_runner")
    deps = host_src + orc_src + [os.path.join(src, f) for f in ("test_units.cpp", "oracle_runner.cpp", "oracle_compute.hpp")]
    newest = max(os.path.getmtime(p) for p in deps)
    if not (os.path.exists(units) and os.path.getmtime(units) >= newest):
        subprocess.check_call(["g++", *flags, os.path.join(src, "test_units.cpp"), *host_src, b.LIB, "-lz", "-lpthread", *rpath, "-o", units])
    if not (os.path.exists(runner) and os.path.getmtime(runner) >= newest):
        subprocess.check_call(["g++", *flags, "-ffp-contract=off", os.path.join(src, "oracle_runner.cpp"), *host_src, *orc_src, b.LIB, "-lz", "-lpthread",
                               *rpath, "-o", runner])
    return dict(units=units, oracle_runner=runner, run_kitti=None, host_lib=None)


def build_test_binaries():
    """tests/host/build/{test_units, oracle_runner}; the product pieces come from ssvio_amd.build.build_host()"""
    if os.environ.get("SSX_SANITIZE") in ("asan", "tsan"):
        return _build_sanitized(os.environ["SSX_SANITIZE"])
    from oracle import pyoracle
    from ssvio_amd import build as b
    host_lib, host_exe = b.build_host()
    pyoracle.build()
    oracle_lib = os.path.join(ROOT, "oracle", "liboracle.so")
    os.makedirs(OUT, exist_ok=True)
    flags = [*b.HOST_FLAGS, "-I", ROOT]
    rpath = ["-Wl,-rpath," + os.path.dirname(host_lib), "-Wl,-rpath," + os.path.dirname(b.LIB), "-Wl,-rpath," + os.path.dirname(oracle_lib),
             "-Wl,-rpath,/opt/rocm/lib"]
    units = os.path.join(OUT, "test_units")
    runner = os.path.join(OUT, "oracle_runner")
    src = os.path.join(ROOT, "tests", "host")
    newest = max(os.path.getmtime(p) for p in (host_lib, oracle_lib, os.path.join(src, "test_units.cpp"), os.path.join(src, "oracle_runner.cpp"),
                                               os.path.join(src, "oracle_compute.hpp")))
    if not (os.path.exists(units) and os.path.getmtime(units) >= newest):
        subprocess.check_call(["g++", *flags, os.path.join(src, "test_units.cpp"), host_lib, b.LIB, "-lpthread", *rpath, "-o", units])
    if not (os.path.exists(runner) and os.path.getmtime(runner) >= newest):
        subprocess.check_call(["g++", *flags, os.path.join(src, "oracle_runner.cpp"), host_lib, b.LIB, oracle_lib, "-lpthread", *rpath, "-o", runner])
    batcher = os.path.join(OUT, "test_batcher_gpu")
    bsrc = os.path.join(src, "test_batcher_gpu.cpp")
    if not (os.path.exists(batcher) and os.path.getmtime(batcher) >= max(os.path.getmtime(host_lib), os.path.getmtime(bsrc))):
        subprocess.check_call(["g++", *flags, bsrc, host_lib, b.LIB, "-lpthread", *rpath, "-o", batcher])
    return dict(units=units, oracle_runner=runner, run_kitti=host_exe, host_lib=host_lib, batcher_gpu=batcher)
