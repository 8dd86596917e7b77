"""Randomised check of the batched streams (GPU box): 2 - 4 DIFFERENT short sequences (lateral / forward drives of different lengths),
random runner settings, a random number of streams S dealt over the sequences and a random number of cohorts C;
ssx_run_kitti --streams=S --batched=C must write, for every stream, byte for byte the trajectory of its sequence's single-stream run
-- whatever the batch composition, with streams finishing at different times, losing track, or never leaving initialisation
(Backend.Async: 1 rounds only ask that every stream finishes: an asynchronous backend is not deterministic by itself).
   python tools/fuzz_batched.py [seed] [rounds]"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import host_util as hu

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rng = np.random.default_rng(900 + seed)
exe = hu.build_test_binaries()["run_kitti"]
bad = 0
for r in range(rounds):
    with tempfile.TemporaryDirectory() as d:
        seqs = []
        for q in range(int(rng.integers(2, 5))):
            n = int(rng.integers(6, 22))
            sd = os.path.join(d, f"s{q}")
            seqs.append(hu.write_corridor_sequence(sd, n_frames=n, seed=int(rng.integers(100))) if rng.random() < 0.4 else
                        hu.write_sequence(sd, n_frames=n, step=float(rng.choice([0.1, 0.3, 0.6, 1.0])), seed=int(rng.integers(100))))
        over = {"Map.ActiveMap.Size": int(rng.choice([2, 3, 5, 10])),
                "numFeatures.trackingGood": int(rng.choice([50, 150, 250, 100000])),
                "numFeatures.trackingBad": int(rng.choice([10, 30])),
                "numFeatures.initGood": int(rng.choice([50, 100])),
                "Min.Init.Landmark.Num": int(rng.choice([50, 200])),
                "Backend.Window": int(rng.random() < 0.8),
                "ORBextractor.nInitFeatures": int(rng.choice([150, 300, 800])),
                "ORBextractor.nNewFeatures": int(rng.choice([50, 100, 300])),
                "Backend.Open": int(rng.random() < 0.9),
                "Backend.Async": int(rng.random() < 0.2),
                "Backend.Jacobian.Numeric": int(rng.random() < 0.2)}
        cfg = hu.write_config(os.path.join(d, "cfg.yaml"), over)
        singles = []
        for q, s in enumerate(seqs):
            t = os.path.join(d, f"single{q}.txt")
            p = subprocess.run([exe, f"--config_yaml_path={cfg}", f"--kitti_dataset_path={s['dir']}", f"--trajectory={t}"], capture_output=True, text=True, timeout=600)
            assert p.returncode == 0, p.stderr[-1000:]
            singles.append(open(t).read())
        S, C = int(rng.integers(3, 41)), int(rng.integers(1, 4))
        many = os.path.join(d, "many.txt")
        p = subprocess.run([exe, f"--config_yaml_path={cfg}", "--kitti_dataset_path=" + ",".join(s["dir"] for s in seqs), f"--trajectory={many}", f"--streams={S}",
                            f"--batched={C}", f"--preload={int(rng.random() < 0.5)}"], capture_output=True, text=True, timeout=900)
        info = dict(round=r, S=S, C=C, frames=[len(s["frames"]) for s in seqs], **over)
        if p.returncode != 0:
            bad += 1; print("FAIL exit code", p.returncode, info, p.stderr[-400:], flush=True); continue
        if over["Backend.Async"]:
            # the asynchronous backend's windows land between frames as the threads' timing has it: no byte-identity to ask for; every
            # stream must finish and write its keyframes
            wrong = [k for k in range(S) if not os.path.exists(f"{many}.{k}")]
        else:
            wrong = [k for k in range(S) if open(f"{many}.{k}").read() != singles[k % len(seqs)]]
        if wrong:
            bad += 1; print("MISMATCH streams", wrong, info, flush=True)
        else:
            print(f"round {r}: ok  S={S} C={C} sequences of {info['frames']} frames, keyframes per sequence {[len(s.splitlines()) for s in singles]}", flush=True)
print("fuzz_batched:", "ALL OK" if not bad else f"{bad} BAD", flush=True)
sys.exit(1 if bad else 0)
