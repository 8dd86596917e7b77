"""Several independent sequences at once on ONE GPU (BASELINE configs[4] runs one stream per GPU; a single stream is
latency-bound, so one GPU has room for more): k concurrent ssx_run_kitti processes on the same synthetic sequence.
   python tools/runner_streams.py [frames] [k ...]"""
import os, re, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import host_util as hu

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ks = [int(a) for a in sys.argv[2:]] or [1, 2, 4, 8]
b = hu.build_test_binaries()
with tempfile.TemporaryDirectory() as d:
    seq = hu.write_sequence(d, n_frames=frames, step=0.03)
    cfg = hu.write_config(os.path.join(d, "cfg.yaml"), {"numFeatures.trackingGood": 200})
    for k in ks:
        t = time.perf_counter()
        procs = [subprocess.Popen([b["run_kitti"], f"--config_yaml_path={cfg}", f"--kitti_dataset_path={seq['dir']}", f"--trajectory={d}/t{i}.txt",
                                   "--decode_threads=8"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for i in range(k)]
        outs = [p.communicate()[0] for p in procs]
        wall = time.perf_counter() - t
        loop = [float(re.search(r"whole loop ([0-9.]+) frames/s", o).group(1)) for o in outs]
        step = [float(re.search(r"RunStep ([0-9.]+) ms/frame", o).group(1)) for o in outs]
        kf = re.search(r"keyframes (\d+)", outs[0]).group(1)
        print(f"{k} streams: per-stream loop {min(loop):.0f}..{max(loop):.0f} frames/s, aggregate {sum(loop):.0f} frames/s; RunStep {sum(step) / k:.3f} ms/frame; "
              f"{kf} keyframes per stream; wall {wall:.1f} s incl. process start")
