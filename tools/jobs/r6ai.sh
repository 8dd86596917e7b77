O=gpurun_out/r6ai; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_ba_gpu.py tests/test_host_gpu.py tests/test_dist_ba_gpu.py tests/test_misuse_gpu.py -x -q -m gpu 2>&1 | tail -6 | tee $O/tests.txt
for rep in 1 2; do
  echo "== packed"; python tools/one_window_time.py 2>&1 | grep wall
  echo "== SSX_BA_NO_PACK=1"; SSX_BA_NO_PACK=1 python tools/one_window_time.py 2>&1 | grep wall
done | tee $O/one_window_ab.txt
python - <<'P' 2>&1 | tee $O/one_window_edges.txt
import sys, time, os
sys.path.insert(0, os.getcwd())
import ssvio_amd
from ssvio_amd import ba
from tools.synth import make_ba_problem
ctx = ssvio_amd.Context(0)
pr = make_ba_problem(P=10, L=4000, seed=1, uv_f32=True)
for we in (False, True):
    for _ in range(5): r = ba.ba_solve(ctx, pr, want_edges=we)
    best = 1e9
    for rep in range(5):
        t = time.perf_counter(); N = 20
        for _ in range(N): r = ba.ba_solve(ctx, pr, want_edges=we)
        best = min(best, (time.perf_counter() - t) / N)
    print('want_edges', we, 'wall ms/solve %.4f' % (best * 1e3))
P
SSX_BA_NO_PACK=1 python - <<'P' 2>&1 | tee -a $O/one_window_edges.txt
import sys, time, os
sys.path.insert(0, os.getcwd())
import ssvio_amd
from ssvio_amd import ba
from tools.synth import make_ba_problem
ctx = ssvio_amd.Context(0)
pr = make_ba_problem(P=10, L=4000, seed=1, uv_f32=True)
for we in (False, True):
    for _ in range(5): r = ba.ba_solve(ctx, pr, want_edges=we)
    best = 1e9
    for rep in range(5):
        t = time.perf_counter(); N = 20
        for _ in range(N): r = ba.ba_solve(ctx, pr, want_edges=we)
        best = min(best, (time.perf_counter() - t) / N)
    print('NO_PACK want_edges', we, 'wall ms/solve %.4f' % (best * 1e3))
P
