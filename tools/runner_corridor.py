"""BASELINE configs[0] shape on the GPU box: N frames of the rendered forward drive through ssx_run_kitti (reference
settings), and the same host code on the CPU oracle for comparison (RunStep time only).
   python tools/runner_corridor.py [frames]"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import host_util as hu

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 100
b = hu.build_test_binaries()
with tempfile.TemporaryDirectory() as d:
    t = time.perf_counter()
    seq = hu.write_corridor_sequence(d, n_frames=frames)
    print(f"rendered {frames} stereo pairs in {time.perf_counter() - t:.1f} s")
    cfg = hu.write_config(os.path.join(d, "cfg.yaml"), {})
    res = {}
    for name, env in (("gpu", dict(os.environ, SSX_HOST_TEST_GPU="1")), ("oracle", dict(os.environ))):
        r = subprocess.run([b["oracle_runner"], cfg, seq["dir"], os.path.join(d, name + ".txt")], capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr
        log = hu.parse_runner_log(r.stdout)
        w = [l for l in r.stdout.splitlines() if l.startswith("runstep_seconds")][0].split()
        res[name] = (float(w[4]), log)
        kfs = [i for i in range(1, frames) if log[i]["keyframes"] > log[i - 1]["keyframes"]]
        err = max(np.abs(np.array(log[i]["centre"]) - seq["centres"][i]).max() for i in range(frames))
        print(f"  {name:7s} frames 1..{frames - 1}: {float(w[4]) / (frames - 1) * 1e3:8.3f} ms/frame = {(frames - 1) / float(w[4]):8.1f} frames/s; "
              f"{log[-1]['keyframes']} keyframes at {kfs}, {log[-1]['points']} map points, statuses {sorted(set(f['status'] for f in log))}, "
              f"max |centre - truth| {err:.3f} m over {seq['centres'][-1][2]:.1f} m")
    same = [{k: v for k, v in f.items() if k != "centre"} for f in res["gpu"][1]] == [{k: v for k, v in f.items() if k != "centre"} for f in res["oracle"][1]]
    print(f"  same decisions on every frame: {same}; ratio oracle / gpu: {res['oracle'][0] / res['gpu'][0]:.1f}x")
    for async_ in (0, 1):
        cfg2 = hu.write_config(os.path.join(d, f"cfg{async_}.yaml"), {"Backend.Async": async_})
        r = subprocess.run([b["run_kitti"], f"--config_yaml_path={cfg2}", f"--kitti_dataset_path={seq['dir']}", f"--trajectory={d}/t{async_}.txt", "--decode_threads=16"],
                           capture_output=True, text=True)
        tum = np.loadtxt(f"{d}/t{async_}.txt", ndmin=2)
        fk = np.rint(tum[:, 0] / seq["dt"]).astype(int)
        err = np.abs(tum[:, 1:4] - seq["centres"][fk]).max()
        print(f"--- ssx_run_kitti, Backend.Async = {async_}: keyframes at {fk.tolist()}, max |keyframe centre - truth| {err:.3f} m")
        print(r.stdout, r.stderr[-400:])
