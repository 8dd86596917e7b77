"""K independent sequences in K threads of ONE ssx_run_kitti process (--streams=K) on one GPU, with the default and a raised
number of HIP hardware queues (GPU_MAX_HW_QUEUES)."""
import os, re, subprocess, sys, tempfile
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import host_util as hu
b = hu.build_test_binaries()
d = "/tmp/hs2"; os.makedirs(d, exist_ok=True)
seq = hu.write_sequence(d, n_frames=300, step=0.03)
cfg = hu.write_config(os.path.join(d, "cfg.yaml"), {"numFeatures.trackingGood": 200})
for preload in (0, 1):
    for k in (2, 4, 8, 16):
        r = subprocess.run([b["run_kitti"], f"--config_yaml_path={cfg}", f"--kitti_dataset_path={seq['dir']}", f"--streams={k}", "--decode_threads=%d" % max(2, 32 // k),
                            f"--preload={preload}"], capture_output=True, text=True)
        print("preload", preload, "streams", k, [l for l in r.stdout.splitlines() if "aggregate" in l], r.stderr[-300:])
