"""End-to-end (front-end + backend) time of the headless runner on the GPU library vs the same host code on the CPU
oracle (one host core), over a synthetic KITTI-layout sequence.  Uses the test binaries (tests/host/build).
   python tools/runner_compare.py [frames] [trackingGood]"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import host_util as hu

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 60
good = int(sys.argv[2]) if len(sys.argv) > 2 else 250
b = hu.build_test_binaries()
with tempfile.TemporaryDirectory() as d:
    seq = hu.write_sequence(d, n_frames=frames, step=0.2)
    cfg = hu.write_config(os.path.join(d, "cfg.yaml"), {"numFeatures.trackingGood": good})
    res = {}
    for name, env in (("gpu", dict(os.environ, SSX_HOST_TEST_GPU="1")), ("oracle", dict(os.environ))):
        r = subprocess.run([b["oracle_runner"], cfg, seq["dir"], os.path.join(d, name + ".txt")], capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr
        log = hu.parse_runner_log(r.stdout)
        w = [l for l in r.stdout.splitlines() if l.startswith("runstep_seconds")][0].split()
        res[name] = (float(w[2]), float(w[4]), log[-1]["keyframes"], log[-1]["points"])
    print(f"{frames} frames, keyframe below {good} inliers; System::RunStep time only (image decoding excluded)")
    for name, (first, rest, kfs, pts) in res.items():
        print(f"  {name:7s} first frame {first * 1e3:8.2f} ms, frames 1..{frames - 1}: {rest / (frames - 1) * 1e3:8.3f} ms/frame = {(frames - 1) / rest:8.1f} frames/s, "
              f"{kfs} keyframes, {pts} map points")
    print(f"  ratio oracle / gpu: {res['oracle'][1] / res['gpu'][1]:.1f}x (oracle = the CPU restatement on ONE host core)")
