"""Randomised GPU-vs-oracle parity sweep (run on the GPU box; not part of the test suite: it takes minutes).
   python tools/fuzz_parity.py [seed] [rounds]
Every round draws an image size, extractor settings, LK settings and a BA / pose-only / pose-graph problem shape,
runs the GPU library and the CPU oracle, and reports every mismatch.  Exit code 1 on any mismatch."""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import ssvio_amd
from oracle import pyoracle as po
from ssvio_amd import ba, lk, orb
from tools.synth import make_ba_problem, make_pose_graph_problem, make_pose_only_problem, make_stereo_pair

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rng = np.random.default_rng(seed)
po.build()
ctx = ssvio_amd.Context(0)
bad = []


def same(a, b):
    return a.dtype == b.dtype and a.shape == b.shape and a.tobytes() == b.tobytes()


def check(name, cond, info):
    if not cond:
        bad.append((name, info))
        print("MISMATCH", name, info, flush=True)


for r in range(rounds):
    t0 = time.time()
    h = int(rng.integers(60, 900)); w = int(rng.integers(80, 1500))
    if rng.random() < 0.15: h, w = int(rng.integers(700, 1200)), int(rng.integers(1300, 2100))     # > 0.6 Mpx: the global-key octree path
    if rng.random() < 0.1: h, w = int(rng.integers(40, 70)), int(rng.integers(40, 90))            # barely larger than the borders
    nblobs = int(h * w / 120)
    L, R, _ = make_stereo_pair(seed=1000 * seed + r, h=h, w=w, n_blobs=nblobs)
    if rng.random() < 0.25:
        L = (L.astype(np.float32) * 0.3 + 90).astype(np.uint8)          # low contrast: the min-threshold retry
    nfeat = int(rng.choice([50, 300, 1000, 2000, 3500]))
    nlev = int(rng.integers(1, 9)); sf = float(rng.choice([1.1, 1.2, 1.3, 1.5, 2.0]))
    ini = int(rng.choice([10, 20, 35])); mn = int(rng.choice([3, 7, ini]))
    info = dict(round=r, h=h, w=w, nfeat=nfeat, nlev=nlev, sf=sf, ini=ini, mn=mn)
    try:
        ex = orb.ORBextractor(ctx, nfeat, sf, nlev, ini, mn)
        prm = po.orb_params(nfeatures=nfeat, scale_factor=sf, nlevels=nlev, ini_th=ini, min_th=mn)
        mask = None
        if rng.random() < 0.5:
            mask = np.full((h, w), 255, np.uint8)
            for _ in range(int(rng.integers(1, 40))):
                y, x = int(rng.integers(0, h)), int(rng.integers(0, w))
                mask[max(y - 10, 0):y + 11, max(x - 10, 0):x + 11] = 0
        kg, dg = ex.DetectAndCompute(L, mask)
        ko, do = po.orb_extract(L, mask=mask, prm=prm)
        check("extract", same(kg, ko) and same(dg, do), info)
        kd = ex.Detect(L, mask); kdo = po.orb_detect(L, mask=mask, prm=prm)
        check("detect", same(kd, kdo), info)
        kr, dr = ex.DetectAndCompute(R)
        kro, dro = po.orb_extract(R, prm=prm)
        check("extract_R", same(kr, kro) and same(dr, dro), info)
        mi, md = orb.stereo_match(ctx, kg, dg, kr, dr)
        mio, mdo = po.stereo_match(ko, do, kro, dro)
        check("match", np.array_equal(mi, mio) and np.array_equal(md, mdo), info)
        # LK
        if len(kd) > 0:
            pts = np.stack([kd["x"], kd["y"]], 1).astype(np.float32)[:600]
            win = int(rng.choice([5, 7, 11, 15])); lev = int(rng.integers(0, 5)); mi_ = int(rng.choice([3, 10, 30]))
            init = pts + rng.normal(0, 2, pts.shape).astype(np.float32) if rng.random() < 0.5 else None
            g = lk.calcOpticalFlowPyrLK(ctx, L, R, pts, init, winSize=win, maxLevel=lev, maxCount=mi_)
            o = po.lk_track(L, R, pts, init, prm=po.lk_params(win=win, max_level=lev, max_iters=mi_, use_initial_flow=int(init is not None)))
            check("lk", g[3] == o[3] and same(g[0], o[0]) and same(g[1], o[1]) and same(g[2], o[2]), dict(info, win=win, lev=lev, it=mi_))
        # BA (small and medium windows), pose-only, pose graph
        P = int(rng.integers(1, 30)); Lm = int(rng.integers(20, 1500)); k = int(rng.integers(2, 6))
        pr = make_ba_problem(P=P, L=Lm, obs_per_lm=min(k, P), seed=int(rng.integers(1 << 30)), fix_first_pose=bool(rng.random() < 0.5))
        g = ba.ba_solve(ctx, pr)
        o = po.ba_solve(pr, "oracle", jac_mode=0)
        binfo = dict(round=r, P=P, L=Lm, k=k)
        cb = np.asarray(o["chi2"]); nm = 1
        while nm < min(len(cb), len(g["chi2"])) and abs(cb[nm - 1] - cb[nm]) > 1e-6 * abs(cb[nm - 1]) and cb[nm] > 1e-12 * cb[0]: nm += 1
        check("ba_trials", np.array_equal(g["trials"][:nm], o["trials"][:nm]),
              dict(binfo, tg=g["trials"].tolist(), to=o["trials"].tolist(), chi_g=np.asarray(g["chi2"]).tolist()[:6], chi_o=cb.tolist()[:6]))
        if np.array_equal(g["trials"], o["trials"]):
            check("ba_resid", np.abs(np.sqrt(g["edge_chi2"]) - np.sqrt(o["edge_chi2"])).max() < 1e-4, binfo)
        M = int(rng.integers(1, 2200))
        pp = make_pose_only_problem(M=M, seed=int(rng.integers(1 << 30)), frac_gross=float(rng.choice([0, 0.1, 0.4])))
        g = ba.pose_only_opt(ctx, pp["pose"], pp["K"], pp["xyz"], pp["uv"]); o = po.pose_only(pp)
        check("pose_only", g["n_inliers"] == o["n_inliers"] and np.array_equal(g["inliers"], o["inliers"]) and np.abs(g["pose"] - o["pose"]).max() < 1e-7,
              dict(round=r, M=M))
        Pg = int(rng.integers(4, 120))
        pg = make_pose_graph_problem(P=Pg, n_loops=int(rng.integers(0, 4)), seed=int(rng.integers(1 << 30)), n_active=min(7, Pg - 1))
        g = ba.pose_graph_opt(ctx, pg, iters=10); o = po.pose_graph_opt(pg, iters=10)
        # trial counts are only meaningful while chi2 still moves: once converged, rho is rounding noise
        co = np.asarray(o["chi2"]); nmov = 1
        while nmov < min(len(co), len(g["chi2"])) and abs(co[nmov - 1] - co[nmov]) > 1e-6 * abs(co[nmov - 1]): nmov += 1
        check("pose_graph", np.array_equal(g["trials"][:nmov], o["trials"][:nmov]) and np.abs(g["poses"] - o["poses"]).max() < 2e-4,
              dict(round=r, P=Pg, free=int((np.asarray(pg["fixed"]) == 0).sum()), E=len(pg["ei"]), tg=g["trials"].tolist(), to=o["trials"].tolist(),
                   dpose=float(np.abs(g["poses"] - o["poses"]).max()), chi_g=g["chi2"].tolist()[:3], chi_o=o["chi2"].tolist()[:3]))
    except Exception as e:
        traceback.print_exc()
        bad.append(("exception", dict(info, err=str(e))))
    print(f"round {r}: {info} {time.time() - t0:.1f} s", flush=True)
print("mismatches:", len(bad))
for b in bad: print(b)
sys.exit(1 if bad else 0)
