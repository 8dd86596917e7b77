"""Stress of the asynchronous backend (Backend.Async: 1) on the GPU box: random window sizes / keyframe thresholds over an
80-frame sequence, checks exit status, tracking state and trajectory error of every run."""
import sys, subprocess, os, numpy as np, time
sys.path.insert(0,"/root/repo/tests"); sys.path.insert(0,"/root/repo")
import host_util as hu
b = hu.build_test_binaries()
seq = hu.write_sequence("/tmp/hs", n_frames=80, step=0.3)
bad = 0
for rep in range(8):
    over = {"Map.ActiveMap.Size": int(np.random.choice([2,3,7,12])), "numFeatures.trackingGood": int(np.random.choice([100000, 250, 300])), "Backend.Async": 1}
    cfg = hu.write_config("/tmp/hs/cfg.yaml", over)
    t = time.time()
    r = subprocess.run([b["run_kitti"], "--config_yaml_path="+cfg, "--kitti_dataset_path="+seq["dir"], "--trajectory=/tmp/hs/t.txt", "--decode_threads=16"], capture_output=True, text=True, timeout=300)
    tum = np.loadtxt("/tmp/hs/t.txt", ndmin=2)
    fk = np.rint(tum[:,0]/seq["dt"]).astype(int)
    err = np.abs((tum[:,1:4]-tum[0,1:4]) - (seq["centres"][fk]-seq["centres"][fk[0]])).max()
    ok = r.returncode == 0 and "LOST" not in r.stdout and err < 0.1
    bad += not ok
    line = [l for l in r.stdout.splitlines() if l.startswith("frames")]
    print(rep, over, "rc", r.returncode, line, "err %.3f" % err, "%.1fs" % (time.time()-t), "" if ok else "FAIL " + r.stderr[-300:])
print("failures:", bad)
