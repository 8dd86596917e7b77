"""Front-end kernel times per 128 pairs at several batch sizes (SSX_ORB_NO_FORK=1: every kernel alone on one stream): which kernels
depend on the working set (256 images = 783 MB of pyramid + blur; 16 pairs fit the 256 MB MALL).
    SSX_ORB_NO_FORK=1 python tools/fe_batch_sweep.py 8 16 32 64 128"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for P in [int(a) for a in sys.argv[1:]] or [16, 128]:
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--lean", "--steps", "20", "--pairs", str(P)], capture_output=True, text=True).stdout
    d = json.loads(out.strip().splitlines()[-1])
    print(P, " ".join(f"{n}={v['ms_per_step'] * 128 / P:.3f}" for n, v in d["kernels"].items() if v["part"] == "frontend"), flush=True)
