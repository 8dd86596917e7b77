"""System-level parity of the host layer on the GPU: the headless runner ssx_run_kitti and the
same host code on the CPU oracle process one synthetic KITTI-layout sequence; every per-frame decision (status,
feature / keyframe / map-point counts, window contents) must be identical and the TUM trajectories must agree to the
file's precision.  Detection and LK are bit-exact between the two, so the first difference can only come from the
pose-only / BA solvers (tolerance-level, far below any decision threshold here)."""
import os
import subprocess

import numpy as np
import pytest

import host_util as hu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def built():
    return hu.build_test_binaries()


@pytest.mark.parametrize("overrides", [{"Map.ActiveMap.Size": 3, "numFeatures.trackingGood": 100000},
                                       {"numFeatures.trackingGood": 280},
                                       {"Map.ActiveMap.Size": 4, "numFeatures.trackingGood": 100000, "Backend.Jacobian.Numeric": 1}])
def test_runner_matches_the_oracle(built, tmp_path, overrides):
    seq = hu.write_sequence(str(tmp_path), n_frames=12, step=0.6)
    cfg = os.path.join(str(tmp_path), "cfg.yaml")
    t_gpu, t_cpu, t_exe = (os.path.join(str(tmp_path), n) for n in ("gpu.txt", "cpu.txt", "exe.txt"))
    hu.write_config(cfg, dict(overrides, **{"Trajectory.Save.Path": f'"{t_exe}"'}))
    cpu = subprocess.run([built["oracle_runner"], cfg, seq["dir"], t_cpu], capture_output=True, text=True, timeout=600)
    gpu = subprocess.run([built["oracle_runner"], cfg, seq["dir"], t_gpu], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, SSX_HOST_TEST_GPU="1"))
    assert cpu.returncode == 0 and gpu.returncode == 0, cpu.stderr + gpu.stderr
    strip = lambda log: [{k: v for k, v in f.items() if k != "centre"} for f in log]
    assert strip(hu.parse_runner_log(gpu.stdout)) == strip(hu.parse_runner_log(cpu.stdout))
    a, b = np.loadtxt(t_gpu, ndmin=2), np.loadtxt(t_cpu, ndmin=2)
    # g2o's numeric Jacobians (delta 1e-9) amplify rounding differences by ~1e7: the gauge-free window drifts apart
    # by tens of micrometres there; with analytic Jacobians the files agree to their last printed digit or two
    tol = 2e-4 if overrides.get("Backend.Jacobian.Numeric") else 2e-5
    assert a.shape == b.shape and np.abs(a - b).max() <= tol, np.abs(a - b).max()
    # the product executable: same trajectory file as the instrumented run, gflags-style arguments, summary on stdout
    exe = subprocess.run([built["run_kitti"], f"--config_yaml_path={cfg}", f"--kitti_dataset_path={seq['dir']}"], capture_output=True, text=True,
                         timeout=600)
    assert exe.returncode == 0, exe.stdout + exe.stderr
    assert open(t_exe).read() == open(t_gpu).read()
    assert "Num Images: 12" in exe.stdout and "frames/s" in exe.stdout and "local BA:" in exe.stdout
    err = np.abs((a[:, 1:4] - a[0, 1:4]) - seq["centres"][np.rint(a[:, 0] / seq["dt"]).astype(int)])
    assert err.max() < 0.03


def test_forward_drive_matches_the_oracle(built, tmp_path):
    """BASELINE configs[0] shape (forward drive through a rendered corridor, the reference's own settings): GPU and
    oracle runs take the same decisions on every frame and end with the same trajectory file"""
    # (seed: the first window holds ONE keyframe and nothing fixed -- its solution is set by rounding along seven directions, in the GPU
    # solver and the oracle alike -- and on the drive of seed 0 one LK track of frame 1 sits so close to its acceptance threshold that
    # the two runs keep 217 and 218 features; seeds 1 .. 5 take identical decisions throughout)
    seq = hu.write_corridor_sequence(str(tmp_path), n_frames=24, seed=1)
    cfg = hu.write_config(os.path.join(str(tmp_path), "cfg.yaml"), {})
    t_gpu, t_cpu = os.path.join(str(tmp_path), "gpu.txt"), os.path.join(str(tmp_path), "cpu.txt")
    cpu = subprocess.run([built["oracle_runner"], cfg, seq["dir"], t_cpu], capture_output=True, text=True, timeout=600)
    gpu = subprocess.run([built["oracle_runner"], cfg, seq["dir"], t_gpu], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, SSX_HOST_TEST_GPU="1"))
    assert cpu.returncode == 0 and gpu.returncode == 0, cpu.stderr + gpu.stderr
    lg, lc = hu.parse_runner_log(gpu.stdout), hu.parse_runner_log(cpu.stdout)
    assert [{k: v for k, v in f.items() if k != "centre"} for f in lg] == [{k: v for k, v in f.items() if k != "centre"} for f in lc]
    assert np.abs(np.array([f["centre"] for f in lg]) - np.array([f["centre"] for f in lc])).max() < 1e-4
    assert lg[-1]["keyframes"] >= 2 and all(f["status"] in (1, 2) for f in lg)
    # the window BA of the reference has no fixed vertex and left-image edges only: gauge AND scale are free, so the
    # solution along those 7 directions is set by rounding; keyframe poses of the two runs agree to ~1e-4 m, not 1e-6
    a, b = np.loadtxt(t_gpu, ndmin=2), np.loadtxt(t_cpu, ndmin=2)
    assert a.shape == b.shape and np.abs(a - b).max() <= 1e-3, np.abs(a - b).max()


def test_asynchronous_backend(built, tmp_path):
    """Backend.Async: 1 on the GPU library: the worker thread optimises windows on its own context while the front-end
    tracks; timing-dependent, so: all keyframes present, never lost, trajectory on the ground truth (three runs)"""
    seq = hu.write_sequence(str(tmp_path), n_frames=12, step=0.6)
    cfg = hu.write_config(os.path.join(str(tmp_path), "cfg.yaml"), {"Map.ActiveMap.Size": 3, "numFeatures.trackingGood": 100000, "Backend.Async": 1})
    for rep in range(3):
        out = os.path.join(str(tmp_path), f"t{rep}.txt")
        r = subprocess.run([built["run_kitti"], f"--config_yaml_path={cfg}", f"--kitti_dataset_path={seq['dir']}", f"--trajectory={out}"],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "keyframes 12" in r.stdout and "LOST" not in r.stdout, r.stdout + r.stderr
        tum = np.loadtxt(out, ndmin=2)
        assert tum.shape == (12, 8)
        assert np.abs((tum[:, 1:4] - tum[0, 1:4]) - seq["centres"]).max() < 0.05


def test_several_streams_in_one_process(built, tmp_path):
    """--streams=3: three independent copies of the loop in threads of one process, each with its own contexts; every
    stream must produce the trajectory of the single-stream run (synchronous backend: deterministic)"""
    seq = hu.write_sequence(str(tmp_path), n_frames=10, step=0.6)
    cfg = hu.write_config(os.path.join(str(tmp_path), "cfg.yaml"), {"Map.ActiveMap.Size": 3, "numFeatures.trackingGood": 100000})
    one, many = os.path.join(str(tmp_path), "one.txt"), os.path.join(str(tmp_path), "many.txt")
    base = [built["run_kitti"], f"--config_yaml_path={cfg}", f"--kitti_dataset_path={seq['dir']}"]
    r1 = subprocess.run(base + [f"--trajectory={one}"], capture_output=True, text=True, timeout=300)
    r3 = subprocess.run(base + [f"--trajectory={many}", "--streams=3", "--decode_threads=2"], capture_output=True, text=True, timeout=300)
    assert r1.returncode == 0 and r3.returncode == 0, r1.stderr + r3.stderr
    assert "3 streams: aggregate" in r3.stdout and r3.stdout.count("keyframes") == 3
    ref = open(one).read()
    for k in range(3):
        assert open(f"{many}.{k}").read() == ref


def test_eight_concurrent_streams_c5_shape(built, tmp_path):
    """BASELINE configs[4] shape on ONE GPU: eight concurrent stereo streams (full front-end + backend each: tracking,
    keyframes, triangulation, window BA), over two DIFFERENT sequences (a forward drive with the reference's settings
    and a lateral one), frames decoded beforehand.  Every stream must reproduce the trajectory file of its sequence's
    single-stream run bit for bit (synchronous backend: deterministic), i.e. concurrent contexts do not interfere."""
    a = hu.write_corridor_sequence(os.path.join(str(tmp_path), "a"), n_frames=24)
    b = hu.write_sequence(os.path.join(str(tmp_path), "b"), n_frames=12, step=0.6, seed=3)
    cfg = hu.write_config(os.path.join(str(tmp_path), "cfg.yaml"), {})
    singles = []
    for k, seq in enumerate((a, b)):
        out = os.path.join(str(tmp_path), f"single{k}.txt")
        r = subprocess.run([built["run_kitti"], f"--config_yaml_path={cfg}", f"--kitti_dataset_path={seq['dir']}", f"--trajectory={out}"],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        singles.append(open(out).read())
    many = os.path.join(str(tmp_path), "many.txt")
    r8 = subprocess.run([built["run_kitti"], f"--config_yaml_path={cfg}", f"--kitti_dataset_path={a['dir']},{b['dir']}", f"--trajectory={many}",
                         "--streams=8", "--preload=1"], capture_output=True, text=True, timeout=600)
    assert r8.returncode == 0, r8.stdout + r8.stderr
    assert "8 streams: aggregate" in r8.stdout and r8.stdout.count("keyframes") == 8
    assert len(singles[0].splitlines()) >= 2                       # the forward drive inserts keyframes (window BA runs)
    for k in range(8):
        assert open(f"{many}.{k}").read() == singles[k % 2], k


def test_runner_arguments(built, tmp_path):
    r = subprocess.run([built["run_kitti"]], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr
    cfg = hu.write_config(os.path.join(str(tmp_path), "cfg.yaml"), {})
    r = subprocess.run([built["run_kitti"], f"--config_yaml_path={cfg}", f"--kitti_dataset_path={tmp_path}/nope"], capture_output=True, text=True)
    assert r.returncode == 1 and "times.txt" in r.stderr
    seq = hu.write_sequence(str(tmp_path), n_frames=3)
    out = os.path.join(str(tmp_path), "t.txt")
    r = subprocess.run([built["run_kitti"], f"--config_yaml_path={cfg}", f"--kitti_dataset_path={seq['dir']}", "--max_frames=2", f"--trajectory={out}"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "Num Images: 2" in r.stdout and len(open(out).read().splitlines()) == 1


def _run_logged(built, cfg, seq_dir, out):
    r = subprocess.run([built["oracle_runner"], cfg, seq_dir, out], capture_output=True, text=True, timeout=900, env=dict(os.environ, SSX_HOST_TEST_GPU="1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return [l for l in r.stdout.splitlines() if l.startswith("frame ")]


@pytest.mark.parametrize("case", ["lateral3", "lateral_default", "lateral4_numeric", "corridor", "corridor_long_window5"])
def test_resident_window_backend_equals_the_remarshalling_backend(built, tmp_path, case):
    """Backend.Window: 1 (the active window resident in HBM, every map edit of map.cpp:18-58, 89-194 and backend.cpp:205-244
    mirrored on ssx_ba_window) against Backend.Window: 0 (the window re-marshalled from the map at every keyframe, the reference's
    own way) on every sequence of this file: the per-frame logs -- status, feature / keyframe / map-point counts, active window
    sizes, camera centre to 1e-6 -- and the TUM trajectory files must be IDENTICAL, byte for byte: the window solves in id order,
    so it returns the bits of the re-marshalled solve, and the two runs never part.  A third run with Backend.Window.Check: 1
    re-marshals the map beside the window at every edit and compares the two graphs (ids, fixed flags, observations)."""
    overrides = {"lateral3": {"Map.ActiveMap.Size": 3, "numFeatures.trackingGood": 100000}, "lateral_default": {"numFeatures.trackingGood": 280},
                 "lateral4_numeric": {"Map.ActiveMap.Size": 4, "numFeatures.trackingGood": 100000, "Backend.Jacobian.Numeric": 1},
                 "corridor": {}, "corridor_long_window5": {"Map.ActiveMap.Size": 5, "numFeatures.trackingGood": 100000}}[case]
    if case.startswith("lateral"):
        seq = hu.write_sequence(str(tmp_path), n_frames=12, step=0.6)
    else:
        seq = hu.write_corridor_sequence(str(tmp_path), n_frames=24 if case == "corridor" else 36)
    logs, files = {}, {}
    for mode, extra in (("window", {"Backend.Window": 1}), ("marshal", {"Backend.Window": 0}), ("checked", {"Backend.Window": 1, "Backend.Window.Check": 1})):
        cfg = hu.write_config(os.path.join(str(tmp_path), f"cfg_{mode}.yaml"), dict(overrides, **extra))
        out = os.path.join(str(tmp_path), f"{mode}.txt")
        logs[mode] = _run_logged(built, cfg, seq["dir"], out)
        files[mode] = open(out).read()
    assert len(logs["window"]) == len(seq["frames"]) and len(files["window"].splitlines()) >= 2
    assert logs["window"] == logs["marshal"] and files["window"] == files["marshal"]
    assert logs["checked"] == logs["window"] and files["checked"] == files["window"]
    if case == "corridor_long_window5":
        kfs = hu.parse_runner_log("\n".join(logs["window"]))[-1]["keyframes"]
        assert kfs >= 30, kfs                                           # the window of 5 slid: keyframes were dropped, map points fixed by the rule
