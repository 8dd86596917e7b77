"""Parameter exploration for a HARD configs[0]-shaped drive (a keyframe every few frames, windows of >= 4000 edges): the corridor renderer
with a faster, swaying drive and denser extraction, through ssx_run_kitti; prints keyframes / windows / edges per window.
    python tools/c1_hard_explore.py"""
import os, re, subprocess, sys, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ssvio_amd import build as sb
from tools.synth import make_corridor_sequence, write_kitti_sequence, write_settings
_, exe = sb.build_host()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 120
for step, amp in ((0.8, 0.3), (1.6, 1.0)):
    d = f"/tmp/ssx_hard_{N}_{step}_{amp}"
    if not os.path.exists(os.path.join(d, "times.txt")):
        frames, _, centres = make_corridor_sequence(n_frames=N, step=step, lateral_amp=amp, workers=min(32, os.cpu_count() or 1))
        write_kitti_sequence(d, frames); np.save(os.path.join(d, "centres.npy"), centres)
    for nfeat, good in ((500, 350), (500, 420), (700, 500), (700, 600)):
        cfg = write_settings(os.path.join(d, f"cfg_{nfeat}_{good}.yaml"), {"ORBextractor.nInitFeatures": nfeat, "ORBextractor.nNewFeatures": nfeat, "numFeatures.trackingGood": good,
                                                                             "numFeatures.initGood": 100, "Min.Init.Landmark.Num": 200, "Map.ActiveMap.Size": 12})
        r = subprocess.run([exe, f"--config_yaml_path={cfg}", f"--kitti_dataset_path={d}", f"--trajectory={d}/t.txt", "--decode_threads=24"], capture_output=True, text=True, timeout=900)
        st = re.search(r"frames (\d+)  keyframes (\d+)  map points (\d+)  final status (\w+)", r.stdout)
        bw = re.search(r"local BA: (\d+) windows, (\d+) LM iterations, (\d+) edges, (\d+) outlier edges", r.stdout)
        rs = re.search(r"RunStep ([0-9.]+) ms/frame", r.stdout)
        print(f"step {step} amp {amp} nfeat {nfeat} trackingGood {good}: rc {r.returncode} {st.group(0) if st else r.stderr[-300:]} | {bw.group(0) if bw else ''} | "
              f"edges/window {int(bw.group(3)) / max(int(bw.group(1)), 1):.0f} | {rs.group(0) if rs else ''}", flush=True)
