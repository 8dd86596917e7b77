"""One live stream (BASELINE configs[0]'s loop) stage by stage: ssx_run_kitti on the default corridor (200 pairs) and on the hard
drive (240 pairs, windows of ~8000 edges) with its per-stage wall times, and -- given a rocprofv3 kernel trace of such a run -- the
share of the run's span in which the GPU executes a kernel at all (the rest is the host between dependent calls).
    python tools/c1_stages.py run [default|hard]        -> prints the runner's stage table
    python tools/c1_stages.py trace <kernel_trace.csv>  -> busy fraction + the kernels by total time"""
import csv, os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def dataset(kind):
    from tools.synth import write_settings
    if kind == "hard":
        d, gen, over = "/tmp/ssx_c1_hard_240", ["240", "32", "1.6", "1.0"], {"ORBextractor.nInitFeatures": 500, "ORBextractor.nNewFeatures": 500,
                                                                             "numFeatures.trackingGood": 450, "Map.ActiveMap.Size": 12}
    else:
        d, gen, over = "/tmp/ssx_c1_corridor_200", ["200"], {}
    if not os.path.exists(os.path.join(d, "times.txt")):
        t = time.time()
        subprocess.check_call([sys.executable, "-m", "tools.synth", "corridor", d, *gen], cwd=ROOT)
        print(f"rendered {d} in {time.time() - t:.1f} s", flush=True)
    return d, write_settings(os.path.join(d, "cfg_stages.yaml"), over)


def command(kind):
    from ssvio_amd import build as sb
    _, exe = sb.build_host()
    d, cfg = dataset(kind)
    return [exe, f"--config_yaml_path={cfg}", f"--kitti_dataset_path={d}", f"--trajectory={d}/traj_stages.txt", "--preload=1"]


if sys.argv[1] == "run":
    kind = sys.argv[2] if len(sys.argv) > 2 else "default"
    cmd = command(kind)
    subprocess.run(cmd + ["--max_frames=20"], capture_output=True, text=True)
    for rep in range(2):
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=1200)
    print(kind, "\n".join(l for l in r.stdout.splitlines() if re.search(r"calls|RunStep|local BA|per window|frames \d+", l)))
elif sys.argv[1] == "cmd":
    print(" ".join(command(sys.argv[2])))
else:
    rows = list(csv.DictReader(open(sys.argv[2])))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), (re.search(r"(k_\w+|__amd_rocclr_\w+)", r["Kernel_Name"]) or re.search(r"(\w+)", r["Kernel_Name"])).group(1))
                for r in rows)
    t0, t1 = ev[len(ev) // 10][0], ev[-1][1]                      # (the first tenth: context set-up, first keyframes)
    sel = [e for e in ev if e[0] >= t0]
    busy, ce = 0, t0
    for s, e, _ in sel:
        if e > ce: busy += e - max(s, ce); ce = e
    gaps = sorted((sel[i + 1][0] - max(x[1] for x in sel[max(0, i - 8):i + 1]) for i in range(len(sel) - 1)), reverse=True)
    print(f"span {(t1 - t0) / 1e6:.1f} ms, {len(sel)} launches, GPU busy {busy / (t1 - t0):.3f}; gaps > 20 us: {sum(g > 20000 for g in gaps)}, 5..20 us: {sum(5000 < g <= 20000 for g in gaps)}")
    by = {}
    for s, e, k in sel:
        a = by.setdefault(k, [0, 0]); a[0] += e - s; a[1] += 1
    for k, (v, n) in sorted(by.items(), key=lambda x: -x[1][0])[:22]:
        print(f"  {k:30s} {n:6d} launches {v / 1e6:8.2f} ms  {v / n / 1e3:7.1f} us each  {v / (t1 - t0):.3f} of span")
