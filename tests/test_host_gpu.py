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


def _strip(log):
    return [{k: v for k, v in f.items() if k != "centre"} for f in log]


def _run_both(built, cfg, seq_dir, t_gpu, t_cpu):
    cpu = subprocess.run([built["oracle_runner"], cfg, seq_dir, t_cpu], capture_output=True, text=True, timeout=1800)
    gpu = subprocess.run([built["oracle_runner"], cfg, seq_dir, t_gpu], capture_output=True, text=True, timeout=1800,
                         env=dict(os.environ, SSX_HOST_TEST_GPU="1"))
    assert cpu.returncode == 0 and gpu.returncode == 0, cpu.stderr[-2000:] + gpu.stderr[-2000:]
    return hu.parse_runner_log(gpu.stdout), hu.parse_runner_log(cpu.stdout)


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_forward_drive_matches_the_oracle(built, tmp_path, seed):
    """BASELINE configs[0] shape (forward drive through a rendered corridor, the reference's own settings), six drives: GPU and
    oracle runs take the same decisions on every frame and end with the same trajectory file -- up to ONE kind of difference, which
    is asserted as such.  Detection, LK and the map bookkeeping are bit-exact between the two; the pose-only LM and the window BA
    are not (different summation orders), and the reference's window BA has no fixed vertex and left-image edges only, so gauge AND
    scale are free inside a window and its solution along those seven directions is set by rounding: keyframe poses of two
    faithful implementations agree to ~1e-4 m after the first optimisation, LK's initial guesses (projected map points) then differ
    by ~1e-3 px, and a track that sits on one of LK's acceptance thresholds is kept by one run and dropped by the other.  On the
    drive of seed 0 that happens to one track of frame 1 (217 / 218 features); seeds 1 .. 5 see no such track in 24 frames.  So:
    every frame identical, or -- a difference of at most two in the feature count of some frames (and, from the next keyframe on,
    in the map-point counts: the extra track becomes an extra map point), nothing else (same status, keyframes, window
    keyframes), camera centres within 1e-3 m."""
    seq = hu.write_corridor_sequence(str(tmp_path), n_frames=24, seed=seed)
    cfg = hu.write_config(os.path.join(str(tmp_path), "cfg.yaml"), {})
    t_gpu, t_cpu = os.path.join(str(tmp_path), "gpu.txt"), os.path.join(str(tmp_path), "cpu.txt")
    lg, lc = _run_both(built, cfg, seq["dir"], t_gpu, t_cpu)
    assert len(lg) == len(lc) == 24
    n_diff = 0
    for fg, fc in zip(_strip(lg), _strip(lc)):
        if fg != fc:
            n_diff += 1
            # the extra track is an extra feature, and at the next keyframe an extra map point; decisions stay the same
            soft = ("features", "points", "active_points")
            assert {k: v for k, v in fg.items() if k not in soft} == {k: v for k, v in fc.items() if k not in soft}, (fg, fc)
            assert all(abs(fg[k] - fc[k]) <= 2 for k in soft), (fg, fc)
    if seed != 0:
        assert n_diff == 0, n_diff                                   # (five of the six drives: nothing differs at all)
    assert np.abs(np.array([f["centre"] for f in lg]) - np.array([f["centre"] for f in lc])).max() < 1e-3
    assert lg[-1]["keyframes"] >= 2 and all(f["status"] in (1, 2) for f in lg)
    a, b = np.loadtxt(t_gpu, ndmin=2), np.loadtxt(t_cpu, ndmin=2)
    assert a.shape == b.shape and np.abs(a - b).max() <= 1e-3, np.abs(a - b).max()


def _umeyama(X, Y):
    """similarity (s, R, t) minimising |s R X + t - Y| (what `evo_ape -as` aligns with)"""
    mx, my = X.mean(0), Y.mean(0)
    Xc, Yc = X - mx, Y - my
    U, D, Vt = np.linalg.svd(Yc.T @ Xc / len(X))
    S = np.eye(3); S[2, 2] = np.sign(np.linalg.det(U) * np.linalg.det(Vt))
    R = U @ S @ Vt
    s = np.trace(np.diag(D) @ S) / (Xc ** 2).sum() * len(X)
    return s, R, my - s * R @ mx


def test_c1_200_pairs_through_test_system(built, tmp_path):
    """BASELINE configs[0] AT ITS STATED SIZE (/root/reference/test/test_system.cpp:36-49: 200 stereo pairs, kitti_00.yaml, no loop
    closure): the rendered 200-pair corridor drive (159 m; real KITTI frames do not exist offline) through the headless test_system
    on the GPU library and through the same host code on the CPU oracle.

    What CAN be asserted over 200 frames, and why not more.  The two runs are identical frame by frame until a single LK track
    lands on different sides of an acceptance threshold (see test_forward_drive_matches_the_oracle: inevitable once the gauge-free
    window BA has run; it happens once per ~60 frames on these drives -- frame 1 here); from the next keyframe decision on they are
    two different, equally valid runs of a chaotic system (keyframes at different frames, trajectories ~1 m apart).  So the
    assertions are (1) identity up to the first difference, which must be a feature count off by at most two and nothing else;
    (2) both runs complete without ever losing track, with keyframe counts within two of each other; (3) ACCURACY against the
    generator's ground truth, for both: the reference's window BA holds no fixed vertex and only left-image observations, i.e. it is
    a monocular BA whose scale drifts (measured here: 2-3 %) -- the keyframe trajectory is within 3 % of the path length of the
    truth when merely anchored at the first keyframe, within 0.7 m RMSE (0.45 % of 159 m) after the similarity alignment `evo_ape
    -as` applies, at a scale within 4 % of one; (4) the product executable writes the instrumented GPU run's trajectory file byte
    for byte, twice (determinism)."""
    from tools.synth import make_corridor_sequence, write_kitti_sequence
    d = "/tmp/ssx_c1_corridor_200"                                  # (bench.py's configs[0] leg renders and keeps the same drive)
    if not os.path.exists(os.path.join(d, "times.txt")) or not os.path.exists(os.path.join(d, "centres.npy")):
        frames, _, centres = make_corridor_sequence(n_frames=200, workers=min(32, os.cpu_count() or 1))
        write_kitti_sequence(d, frames)
        np.save(os.path.join(d, "centres.npy"), centres)
    centres = np.load(os.path.join(d, "centres.npy"))
    path_len = float(np.linalg.norm(np.diff(centres, axis=0), axis=1).sum())
    cfg = hu.write_config(os.path.join(str(tmp_path), "cfg.yaml"), {})
    t_gpu, t_cpu = os.path.join(str(tmp_path), "gpu.txt"), os.path.join(str(tmp_path), "cpu.txt")
    lg, lc = _run_both(built, cfg, d, t_gpu, t_cpu)
    assert len(lg) == len(lc) == 200
    # (1) identical up to the first difference; the first difference is a feature count
    sg, sc = _strip(lg), _strip(lc)
    first = next((i for i in range(200) if sg[i] != sc[i]), None)
    if first is not None:
        fg, fc = sg[first], sc[first]
        assert {k: v for k, v in fg.items() if k != "features"} == {k: v for k, v in fc.items() if k != "features"}, (fg, fc)
        assert abs(fg["features"] - fc["features"]) <= 2, (fg, fc)
        assert np.abs(np.array([f["centre"] for f in lg[:first + 1]]) - np.array([f["centre"] for f in lc[:first + 1]])).max() < 1e-3
    # (2) both runs track throughout
    for log in (lg, lc):
        assert all(f["status"] in (1, 2) for f in log) and log[-1]["keyframes"] >= 8
    assert abs(lg[-1]["keyframes"] - lc[-1]["keyframes"]) <= 2
    # (3) accuracy against the ground truth
    for name, tfile in (("gpu", t_gpu), ("cpu", t_cpu)):
        tum = np.loadtxt(tfile, ndmin=2)
        idx = np.rint(tum[:, 0] / 0.1).astype(int)
        est, gt = tum[:, 1:4], centres[idx]
        anchored = np.linalg.norm((est - est[0]) - (gt - gt[0]), axis=1)
        s, R, t = _umeyama(est, gt)
        aligned = np.linalg.norm((s * (R @ est.T).T + t) - gt, axis=1)
        print(f"[c1 {name}] keyframes {len(tum)}, path {path_len:.1f} m: anchored APE rmse {np.sqrt((anchored ** 2).mean()):.3f} max {anchored.max():.3f} m; "
              f"Sim3-aligned rmse {np.sqrt((aligned ** 2).mean()):.3f} max {aligned.max():.3f} m, scale {s:.4f}")
        assert anchored.max() < 0.03 * path_len, (name, anchored.max())
        assert np.sqrt((aligned ** 2).mean()) < 0.7 and abs(s - 1.0) < 0.04, (name, np.sqrt((aligned ** 2).mean()), s)
    # (4) the product executable: the same file, twice
    outs = []
    for rep in range(2):
        out = os.path.join(str(tmp_path), f"exe{rep}.txt")
        r = subprocess.run([built["run_kitti"], f"--config_yaml_path={cfg}", f"--kitti_dataset_path={d}", f"--trajectory={out}", "--decode_threads=16"],
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and "Num Images: 200" in r.stdout, r.stdout[-1000:] + r.stderr[-1000:]
        outs.append(open(out).read())
    assert outs[0] == outs[1] == open(t_gpu).read()


HARD_DRIVE = dict(n_frames=240, step=1.6, lateral_amp=1.0)
HARD_SETTINGS = {"ORBextractor.nInitFeatures": 500, "ORBextractor.nNewFeatures": 500, "numFeatures.trackingGood": 450, "Map.ActiveMap.Size": 12}


def hard_drive_dir():
    """the rendered HARD drive (cached in /tmp; bench.py's c1.hard_drive leg uses the same one)"""
    from tools.synth import make_corridor_sequence, write_kitti_sequence
    d = "/tmp/ssx_c1_hard_240"
    if not os.path.exists(os.path.join(d, "times.txt")) or not os.path.exists(os.path.join(d, "centres.npy")):
        frames, _, centres = make_corridor_sequence(workers=min(32, os.cpu_count() or 1), **HARD_DRIVE)
        write_kitti_sequence(d, frames)
        np.save(os.path.join(d, "centres.npy"), centres)
    return d


def test_c1_hard_drive_keeps_the_backend_busy(built, tmp_path):
    """A configs[0]-shaped drive that EXERCISES THE BACKEND: the corridor at twice KITTI's speed (1.6 m per frame) with a 1 m sway, 500
    features per keyframe and a keyframe as soon as fewer than 450 of them are tracked -- a keyframe every ~5 frames, >= 40 windows of
    >= 4000 edges each over 240 frames (the 200-pair corridor with the reference's settings inserts 12 keyframes and never shows the
    closed loop a window of BASELINE configs[2]'s size: round 5's review).  GPU library and CPU oracle through the same host code,
    asserted like test_c1_200_pairs_through_test_system: identical until the first LK threshold flip (a feature count off by <= 2),
    both track throughout with keyframe counts within three of each other, accuracy against the generator's ground truth, and the
    product executable reproduces the instrumented run's trajectory byte for byte, its summary showing the windows the backend
    optimised."""
    import re
    d = hard_drive_dir()
    centres = np.load(os.path.join(d, "centres.npy"))
    path_len = float(np.linalg.norm(np.diff(centres, axis=0), axis=1).sum())
    cfg = hu.write_config(os.path.join(str(tmp_path), "cfg.yaml"), HARD_SETTINGS)
    t_gpu, t_cpu = os.path.join(str(tmp_path), "gpu.txt"), os.path.join(str(tmp_path), "cpu.txt")
    lg, lc = _run_both(built, cfg, d, t_gpu, t_cpu)
    n = HARD_DRIVE["n_frames"]
    assert len(lg) == len(lc) == n
    sg, sc = _strip(lg), _strip(lc)
    first = next((i for i in range(n) if sg[i] != sc[i]), None)
    print(f"[hard drive] first frame at which the GPU and the oracle run differ: {first}; keyframes {lg[-1]['keyframes']} (GPU) / {lc[-1]['keyframes']} (oracle)")
    if first is not None:
        fg, fc = sg[first], sc[first]
        soft = ("features", "points", "active_points")
        assert {k: v for k, v in fg.items() if k not in soft} == {k: v for k, v in fc.items() if k not in soft}, (fg, fc)
        assert all(abs(fg[k] - fc[k]) <= 2 for k in soft), (fg, fc)
        assert np.abs(np.array([f["centre"] for f in lg[:first + 1]]) - np.array([f["centre"] for f in lc[:first + 1]])).max() < 1e-3
    for log in (lg, lc):
        assert all(f["status"] in (1, 2) for f in log) and log[-1]["keyframes"] >= 40
    assert abs(lg[-1]["keyframes"] - lc[-1]["keyframes"]) <= 3
    for name, tfile in (("gpu", t_gpu), ("cpu", t_cpu)):
        tum = np.loadtxt(tfile, ndmin=2)
        idx = np.rint(tum[:, 0] / 0.1).astype(int)
        est, gt = tum[:, 1:4], centres[idx]
        anchored = np.linalg.norm((est - est[0]) - (gt - gt[0]), axis=1)
        s, R, t = _umeyama(est, gt)
        aligned = np.linalg.norm((s * (R @ est.T).T + t) - gt, axis=1)
        print(f"[hard drive {name}] keyframes {len(tum)}, path {path_len:.1f} m: anchored APE rmse {np.sqrt((anchored ** 2).mean()):.3f} max {anchored.max():.3f} m; "
              f"Sim3-aligned rmse {np.sqrt((aligned ** 2).mean()):.3f} max {aligned.max():.3f} m, scale {s:.4f}")
        assert anchored.max() < 0.03 * path_len, (name, anchored.max())
        assert np.sqrt((aligned ** 2).mean()) < 0.01 * path_len and abs(s - 1.0) < 0.04, (name, np.sqrt((aligned ** 2).mean()), s)
    out = os.path.join(str(tmp_path), "exe.txt")
    r = subprocess.run([built["run_kitti"], f"--config_yaml_path={cfg}", f"--kitti_dataset_path={d}", f"--trajectory={out}", "--decode_threads=16"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-1000:]
    assert open(out).read() == open(t_gpu).read()
    bw = re.search(r"local BA: (\d+) windows, (\d+) LM iterations, (\d+) edges, (\d+) outlier edges", r.stdout)
    assert bw, r.stdout[-1500:]
    windows, edges = int(bw.group(1)), int(bw.group(3))
    print(f"[hard drive] {windows} windows, {edges / windows:.0f} edges per window, {int(bw.group(2)) / windows:.1f} LM iterations per window; "
          + re.search(r"RunStep.*", r.stdout).group(0))
    assert windows >= 40 and edges / windows >= 4000, r.stdout[-1500:]


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


def test_warmup_changes_no_trajectory(built, tmp_path):
    """System::Warmup (one synthetic keyframe + two tracked frames through every compute call, before the first frame) loads kernels
    and sizes workspaces; it touches neither the map nor the tracker: the run with --warmup=0 writes the same bytes -- alone and as
    batched streams, where the streams' warm-up calls go through the batcher too."""
    seq = hu.write_corridor_sequence(os.path.join(str(tmp_path), "a"), n_frames=16)
    cfg = hu.write_config(os.path.join(str(tmp_path), "cfg.yaml"), {"Map.ActiveMap.Size": 4, "numFeatures.trackingGood": 100000})   # a keyframe per frame
    out = {}
    for w in (1, 0):
        traj = os.path.join(str(tmp_path), f"w{w}.txt")
        r = subprocess.run([built["run_kitti"], f"--config_yaml_path={cfg}", f"--kitti_dataset_path={seq['dir']}", f"--trajectory={traj}", f"--warmup={w}"],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        assert ("warm-up" in r.stdout) == (w == 1)
        out[w] = open(traj).read()
    assert out[1] == out[0] and len(out[1].splitlines()) >= 2
    many = os.path.join(str(tmp_path), "many.txt")
    r = subprocess.run([built["run_kitti"], f"--config_yaml_path={cfg}", f"--kitti_dataset_path={seq['dir']}", f"--trajectory={many}", "--streams=5", "--batched=2",
                        "--preload=1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    for k in range(5):
        assert open(f"{many}.{k}").read() == out[0], k


def test_batched_streams_with_asynchronous_backends_finish(built, tmp_path):
    """Backend.Async: 1 under --batched: the backend's worker thread files its window solves in a request slot of its own while the
    stream's thread goes on tracking, and a front-end that waits for its backend (WaitIdle at the end of the sequence) cannot starve
    the dispatcher (a backend's request is served after 2 ms whatever the front-ends do).  Every stream finishes and writes its
    keyframes -- resident window and re-marshalled window (the worker then solves on a context of its own)."""
    seq = hu.write_corridor_sequence(os.path.join(str(tmp_path), "a"), n_frames=14)
    for window in (1, 0):
        cfg = hu.write_config(os.path.join(str(tmp_path), f"cfg{window}.yaml"), {"Map.ActiveMap.Size": 4, "numFeatures.trackingGood": 100000, "Backend.Async": 1,
                                                                              "Backend.Window": window})
        many = os.path.join(str(tmp_path), f"many{window}.txt")
        r = subprocess.run([built["run_kitti"], f"--config_yaml_path={cfg}", f"--kitti_dataset_path={seq['dir']}", f"--trajectory={many}", "--streams=6", "--batched=2",
                            "--preload=1"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        for k in range(6):
            assert len(open(f"{many}.{k}").read().splitlines()) == 14, (window, k)


@pytest.mark.parametrize("streams,cohorts", [(6, 1), (9, 2)])
def test_stream_batcher_results_and_error_isolation(built, streams, cohorts):
    """tests/host/test_batcher_gpu.cpp: the StreamBatcher driven directly by S threads (masked detection, stereo LK, triangulation, two
    chained temporal LK calls, pose-only -- one frame's calls of FrontEnd); every stream's results are byte for byte those of the same
    calls made one by one through SsxCompute, and a stream that files a request the library rejects gets the exception ALONE: the
    other streams of the same batched call get their results."""
    r = subprocess.run([built["batcher_gpu"], str(streams), str(cohorts)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "test_batcher_gpu: ok" in r.stdout


def test_batched_streams_reproduce_the_single_stream_trajectories(built, tmp_path):
    """--streams=K --batched=1 (ssvio_amd/host/stream_batcher.hpp): the K streams' temporal LK and pose-only LM of every frame and their
    window optimisations reach the GPU as batched library calls (ssx_lk_track_batch / ssx_pose_only_opt_batch /
    ssx_ba_window_solve_batch) instead of K separate launches.  Twelve streams over two DIFFERENT sequences of different lengths
    (the shorter streams finish first: batches shrink, stragglers from keyframes rejoin the cohort): every stream writes, byte for
    byte, the trajectory of its sequence's single-stream run; the jobs really were batched (more than one job per call on average)."""
    import re
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
    assert len(singles[0].splitlines()) >= 2                       # the forward drive inserts keyframes (window BA runs)
    for rep in range(2):                                           # (thread timing differs from run to run: the files must not)
        many = os.path.join(str(tmp_path), f"many{rep}.txt")
        r = subprocess.run([built["run_kitti"], f"--config_yaml_path={cfg}", f"--kitti_dataset_path={a['dir']},{b['dir']}", f"--trajectory={many}",
                            "--streams=12", "--preload=1", f"--batched={1 + rep}"], capture_output=True, text=True, timeout=600)   # one cohort, then two
        assert r.returncode == 0, r.stdout + r.stderr
        for k in range(12):
            assert open(f"{many}.{k}").read() == singles[k % 2], (rep, k)
        m = re.search(r"batched calls: LK (\d+) \(([\d.]+) jobs each\), pose-only (\d+) \(([\d.]+)\), window solves (\d+) \(([\d.]+)\)", r.stdout)
        assert m, r.stdout
        assert float(m.group(2)) > (3.0 if rep == 0 else 1.5) and float(m.group(4)) > (3.0 if rep == 0 else 1.5) and int(m.group(5)) >= 1, r.stdout


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
