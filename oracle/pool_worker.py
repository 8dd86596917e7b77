"""TEST / BENCH INFRASTRUCTURE: the CPU oracle's stereo front-end on several host cores at once, one process per
core on its own synthetic pairs (the path is embarrassingly parallel over frames) -- the "all cores" leg of
bench.py's cpu_baseline.  The workers are plain subprocesses of this module (`python -m oracle.pool_worker idx pairs
start_at`): they import neither torch nor the HIP library, prepare their images, wait for the common start time, run,
and print "<pairs> <seconds>"."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _work(idx, pairs, start_at):
    import numpy as np

    from oracle import pyoracle as po
    from tools.synth import make_stereo_pair
    imgs = [make_stereo_pair(seed=50000 + idx * pairs + i)[:2] for i in range(pairs)]
    po.orb_extract(imgs[0][0][:64, :96].copy())                       # load the library before the clock starts
    late = time.time() > start_at
    while time.time() < start_at:
        time.sleep(0.002)
    t = time.perf_counter()
    for L, R in imgs:
        kL, dL = po.orb_extract(L); kR, dR = po.orb_extract(R)
        m_idx, _ = po.stereo_match(kL, dL, kR, dR)
        m = m_idx >= 0
        uvL = np.stack([kL["x"][m], kL["y"][m]], 1).astype(np.float64)
        uvR = np.stack([kR["x"][m_idx[m]], kR["y"][m_idx[m]]], 1).astype(np.float64)
        po.triangulate(uvL, uvR, (718.856, 718.856, 607.1928, 185.2157), 386.1448 / 718.856)
    print(pairs, time.perf_counter() - t, int(late), flush=True)


def frontend_all_cores(workers, pairs_per_worker=2, timeout=180.0):
    """-> (stereo frames/s summed over the workers, workers that reported on time)"""
    start_at = time.time() + 4.0 + 0.03 * workers                     # process start + numpy import + image synthesis
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, "-m", "oracle.pool_worker", str(i), str(pairs_per_worker), repr(start_at)], cwd=ROOT, env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for i in range(workers)]
    rate, ok = 0.0, 0
    deadline = time.time() + timeout
    for p in procs:
        try:
            out, _ = p.communicate(timeout=max(1.0, deadline - time.time()))
            w = out.split()
            if p.returncode == 0 and len(w) == 3 and w[2] == "0":
                rate += int(w[0]) / float(w[1]); ok += 1
        except subprocess.TimeoutExpired:
            p.kill()
    return rate, ok


if __name__ == "__main__":
    _work(int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]))
