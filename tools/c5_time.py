"""BASELINE configs[4] on one GPU: S concurrent streams of the configs[0] drive through ssx_run_kitti, unbatched (every stream its own
launches) and batched (ssvio_amd/host/stream_batcher.hpp).   python tools/c5_time.py [frames=200] [S ...]"""
import os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ssvio_amd import build as sb
from tools.synth import write_settings
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 200
Ss = [int(x) for x in sys.argv[2:]] or [1, 8, 32, 64]
d = f"/tmp/ssx_c1_corridor_{frames}"
if not os.path.exists(os.path.join(d, "times.txt")):
    t = time.time()
    subprocess.check_call([sys.executable, "-m", "tools.synth", "corridor", d, str(frames)], cwd=ROOT)
    print(f"rendered {frames} pairs in {time.time() - t:.1f} s", flush=True)
_, exe = sb.build_host()
cfg = write_settings(os.path.join(d, "cfg_c5.yaml"), {})
def run(extra, tag):
    traj = os.path.join(d, f"traj_{tag}.txt")
    r = subprocess.run([exe, f"--config_yaml_path={cfg}", f"--kitti_dataset_path={d}", f"--trajectory={traj}", *extra], capture_output=True, text=True, timeout=1800)
    if r.returncode != 0:
        print(r.stdout[-2000:], r.stderr[-2000:]); raise SystemExit(1)
    return r.stdout, traj
run(("--max_frames=20",), "warm")
out1, t1 = run(("--decode_threads=24",), "one")
ref = open(t1).read()
print(re.search(r"RunStep.*", out1).group(0))
for S in Ss:
    if S == 1: continue
    for batched in (0, 1, 2, 3):
        if not batched and S > 16: continue
        if batched > 1 and S < 16: continue
        out, tr = run((f"--streams={S}", "--preload=1", f"--batched={batched}"), f"s{S}b{batched}")
        same = all(open(f"{tr}.{k}").read() == ref for k in range(S))
        m = re.search(r"from the common start to the last stream's end = ([0-9.]+) frames/s", out)
        c = re.search(r"batched calls:.*", out)
        c2 = re.search(r"batched time:.*", out)
        print(f"S={S} batched={batched}: {m.group(1)} frames/s, trajectories identical to the single stream: {same}  {c.group(0) if c else ''}\n      {c2.group(0) if c2 else ''}", flush=True)
