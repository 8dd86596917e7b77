"""Exploration for the configs[0] test: (a) the 24-frame corridor over seeds 0-5, GPU vs oracle runner: first divergent frame and how
far the logs part; (b) the 200-pair drive: decisions, trajectory agreement, APE raw and after a similarity alignment."""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import host_util as hu
from tools.synth import make_corridor_sequence, write_kitti_sequence, write_settings

built = hu.build_test_binaries()

def run_pair(cfg, seq_dir, tag):
    tg, tc = f"/tmp/c1x_{tag}_gpu.txt", f"/tmp/c1x_{tag}_cpu.txt"
    cpu = subprocess.run([built["oracle_runner"], cfg, seq_dir, tc], capture_output=True, text=True, timeout=1800)
    gpu = subprocess.run([built["oracle_runner"], cfg, seq_dir, tg], capture_output=True, text=True, timeout=1800, env=dict(os.environ, SSX_HOST_TEST_GPU="1"))
    assert cpu.returncode == 0 and gpu.returncode == 0, cpu.stderr[-500:] + gpu.stderr[-500:]
    return hu.parse_runner_log(gpu.stdout), hu.parse_runner_log(cpu.stdout), np.loadtxt(tg, ndmin=2), np.loadtxt(tc, ndmin=2)

def umeyama(X, Y):
    """similarity (s, R, t) minimising |s R X + t - Y|"""
    mx, my = X.mean(0), Y.mean(0)
    Xc, Yc = X - mx, Y - my
    U, D, Vt = np.linalg.svd(Yc.T @ Xc / len(X))
    S = np.eye(3); S[2, 2] = np.sign(np.linalg.det(U) * np.linalg.det(Vt))
    R = U @ S @ Vt
    s = np.trace(np.diag(D) @ S) / (Xc ** 2).sum() * len(X)
    return s, R, my - s * R @ mx

if "seeds" in sys.argv:
    for seed in range(6):
        d = f"/tmp/c1x_seed{seed}"
        os.makedirs(d, exist_ok=True)
        seq = hu.write_corridor_sequence(d, n_frames=24, seed=seed)
        cfg = hu.write_config(os.path.join(d, "cfg.yaml"), {})
        lg, lc, a, b = run_pair(cfg, seq["dir"], f"s{seed}")
        strip = lambda f: {k: v for k, v in f.items() if k != "centre"}
        diff = [i for i in range(len(lg)) if strip(lg[i]) != strip(lc[i])]
        md = max((abs(lg[i][k] - lc[i][k]) for i in diff for k in ("features", "points", "keyframes", "active_kfs", "active_points", "status")), default=0)
        cen = np.abs(np.array([f["centre"] for f in lg]) - np.array([f["centre"] for f in lc])).max()
        print(f"seed {seed}: frames {len(lg)} kfs {lg[-1]['keyframes']}/{lc[-1]['keyframes']} divergent frames {diff[:6]}{'...' if len(diff) > 6 else ''} max count diff {md} centre diff {cen:.2e} "
              f"traj shape {a.shape}/{b.shape} traj diff {np.abs(a - b).max() if a.shape == b.shape else None}")
        if diff:
            i = diff[0]
            print("   first:", strip(lg[i]), strip(lc[i]))
if "c1" in sys.argv:
    d = "/tmp/ssx_c1_corridor_200"
    if not os.path.exists(os.path.join(d, "times.txt")):
        t = time.time()
        frames, gt, centres = make_corridor_sequence(n_frames=200, workers=min(32, os.cpu_count() or 1))
        write_kitti_sequence(d, frames); np.save(os.path.join(d, "centres.npy"), centres)
        print(f"rendered in {time.time() - t:.1f} s")
    centres = np.load(os.path.join(d, "centres.npy"))
    cfg = write_settings(os.path.join(d, "cfg_x.yaml"), {})
    t = time.time()
    lg, lc, a, b = run_pair(cfg, d, "c1")
    print(f"both runs {time.time() - t:.1f} s")
    strip = lambda f: {k: v for k, v in f.items() if k != "centre"}
    diff = [i for i in range(len(lg)) if strip(lg[i]) != strip(lc[i])]
    print("frames", len(lg), "keyframes", lg[-1]["keyframes"], lc[-1]["keyframes"], "divergent", diff[:8], "centre diff", np.abs(np.array([f["centre"] for f in lg]) - np.array([f["centre"] for f in lc])).max())
    print("traj diff", np.abs(a - b).max() if a.shape == b.shape else (a.shape, b.shape))
    for name, tum in (("gpu", a), ("cpu", b)):
        idx = np.rint(tum[:, 0] / 0.1).astype(int)
        est = tum[:, 1:4]; gtc = centres[idx]
        e0 = np.linalg.norm((est - est[0]) - (gtc - gtc[0]), axis=1)
        s, R, tt = umeyama(est, gtc)
        e1 = np.linalg.norm((s * (R @ est.T).T + tt) - gtc, axis=1)
        # rigid alignment (no scale)
        mx, my = est.mean(0), gtc.mean(0); U, D, Vt = np.linalg.svd((gtc - my).T @ (est - mx)); S = np.eye(3); S[2, 2] = np.sign(np.linalg.det(U @ Vt)); Rr = U @ S @ Vt
        e2 = np.linalg.norm((Rr @ (est - mx).T).T + my - gtc, axis=1)
        print(f"{name}: keyframes {len(tum)} at frames {idx.tolist()} path {np.linalg.norm(np.diff(centres, axis=0), axis=1).sum():.1f} m; APE origin-anchored rmse {np.sqrt((e0**2).mean()):.3f} max {e0.max():.3f}; "
              f"SE3-aligned rmse {np.sqrt((e2**2).mean()):.3f}; Sim3-aligned rmse {np.sqrt((e1**2).mean()):.3f} max {e1.max():.3f} scale {s:.4f}")
        # per-frame tracked centre error (every frame, from the log)
    cg = np.array([f["centre"] for f in lg]); err = np.linalg.norm((cg - cg[0]) - (centres[:len(cg)] - centres[0]), axis=1)
    print("per-frame centre error (gpu run, all 200 frames): rmse %.3f max %.3f; at frames 50/100/150/199: %s" % (np.sqrt((err**2).mean()), err.max(), err[[50, 100, 150, 199]].round(3).tolist()))
    dist = np.linalg.norm(cg - cg[0], axis=1); gd = np.linalg.norm(centres[:len(cg)] - centres[0], axis=1)
    print("travelled / true at frames 50/100/150/199:", (dist[[50, 100, 150, 199]] / gd[[50, 100, 150, 199]]).round(4).tolist())
if "scan" in sys.argv:
    for seed in (1, 2, 3, 4):
        d = f"/tmp/c1x_scan{seed}"
        t = time.time()
        frames, gt, centres = make_corridor_sequence(n_frames=200, seed=seed, workers=min(32, os.cpu_count() or 1))
        write_kitti_sequence(d, frames)
        cfg = write_settings(os.path.join(d, "cfg_x.yaml"), {})
        lg, lc, a, b = run_pair(cfg, d, f"scan{seed}")
        strip = lambda f: {k: v for k, v in f.items() if k != "centre"}
        diff = [i for i in range(len(lg)) if strip(lg[i]) != strip(lc[i])]
        cen = np.abs(np.array([f["centre"] for f in lg]) - np.array([f["centre"] for f in lc])).max()
        print(f"seed {seed}: keyframes {lg[-1]['keyframes']}/{lc[-1]['keyframes']} first divergent {diff[:3]} n {len(diff)} centre diff {cen:.2e} traj {a.shape} {b.shape} "
              f"{np.abs(a - b).max() if a.shape == b.shape else None}  ({time.time() - t:.0f} s)")
        if diff: print("   ", strip(lg[diff[0]]), strip(lc[diff[0]]))
