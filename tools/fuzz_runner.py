"""Randomised system-level parity (GPU box): random runner settings (window size, keyframe thresholds, feature budgets,
FAST thresholds, Jacobian mode) over short lateral / forward-drive sequences, the same host code once on libssx.so and
once on the CPU oracle; reports every frame whose decisions (status, counts) differ and the trajectory difference.
   python tools/fuzz_runner.py [seed] [rounds]"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import host_util as hu

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rng = np.random.default_rng(300 + seed)
b = hu.build_test_binaries()
bad = 0
strip = lambda log: [{k: v for k, v in f.items() if k != "centre"} for f in log]
for r in range(rounds):
    with tempfile.TemporaryDirectory() as d:
        kind = "corridor" if rng.random() < 0.4 else "lateral"
        n = int(rng.integers(8, 20))
        seq = hu.write_corridor_sequence(d, n_frames=n, seed=int(rng.integers(100))) if kind == "corridor" else \
            hu.write_sequence(d, n_frames=n, step=float(rng.choice([0.1, 0.3, 0.6, 1.0])), seed=int(rng.integers(100)))
        over = {"Map.ActiveMap.Size": int(rng.choice([1, 2, 3, 5, 12])),
                "numFeatures.trackingGood": int(rng.choice([50, 150, 250, 100000])),
                "numFeatures.trackingBad": int(rng.choice([10, 30])),
                "numFeatures.initGood": int(rng.choice([50, 100])),
                "Min.Init.Landmark.Num": int(rng.choice([50, 200])),
                "ORBextractor.nInitFeatures": int(rng.choice([150, 300, 800])),
                "ORBextractor.nNewFeatures": int(rng.choice([50, 100, 300])),
                "ORBextractor.iniThFAST": int(rng.choice([12, 20, 30])),
                "ORBextractor.minThFAST": int(rng.choice([5, 7])),
                "Backend.Open": int(rng.random() < 0.85),
                "Backend.Jacobian.Numeric": int(rng.random() < 0.25)}
        cfg = hu.write_config(os.path.join(d, "cfg.yaml"), over)
        res = {}
        for name, env in (("gpu", dict(os.environ, SSX_HOST_TEST_GPU="1")), ("cpu", dict(os.environ))):
            p = subprocess.run([b["oracle_runner"], cfg, seq["dir"], os.path.join(d, name + ".txt")], capture_output=True, text=True, env=env, timeout=600)
            res[name] = (p.returncode, hu.parse_runner_log(p.stdout), p.stderr[-300:])
        (rg, lg, eg), (rc, lc, ec) = res["gpu"], res["cpu"]
        info = dict(round=r, kind=kind, frames=n, **over)
        if rg != 0 or rc != 0:
            bad += 1; print("FAIL exit codes", rg, rc, info, eg, ec, flush=True); continue
        diff = [i for i, (a, c) in enumerate(zip(strip(lg), strip(lc))) if a != c]
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            tg, tc = np.loadtxt(os.path.join(d, "gpu.txt"), ndmin=2), np.loadtxt(os.path.join(d, "cpu.txt"), ndmin=2)
        dt = (float(np.abs(tg - tc).max()) if tg.size else 0.0) if tg.shape == tc.shape else float("nan")
        dc = float(np.abs(np.array([f["centre"] for f in lg]) - np.array([f["centre"] for f in lc])).max()) if len(lg) == len(lc) else float("nan")
        ok = not diff and dt < 2e-3
        if not ok:
            bad += 1
            print("MISMATCH", info, flush=True)
            if diff:
                i = diff[0]
                print("   first differing frame", i, "gpu", strip(lg)[i], "cpu", strip(lc)[i], flush=True)
                print("   centre difference on the frames before:", [float(np.abs(np.array(lg[k]["centre"]) - np.array(lc[k]["centre"])).max()) for k in range(max(0, i - 4), i + 1)], flush=True)
            print("   tum diff", dt, "centre diff", dc, flush=True)
        else:
            print(f"round {r}: ok  {kind} {n} frames, {lg[-1]['keyframes']} keyframes, statuses {sorted(set(f['status'] for f in lg))}, tum diff {dt:.1e}, centre diff {dc:.1e}", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
