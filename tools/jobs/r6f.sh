mkdir -p gpurun_out/r6f
timeout 1500 python tools/c5_time.py 200 8 32 64 128 > gpurun_out/r6f/c5.txt 2>&1; cat gpurun_out/r6f/c5.txt
timeout 600 python -m pytest tests/test_pg_gpu.py tests/test_ba_gpu.py -x -q -k "pg or pose_graph or band or cyclic or c4" > gpurun_out/r6f/tests.log 2>&1; tail -5 gpurun_out/r6f/tests.log
timeout 300 python -c "
import sys; sys.path.insert(0,'.')
import ssvio_amd
from tools import bench_next
ctx = ssvio_amd.Context(0)
r = bench_next.next_rows(ssvio_amd, ctx, cpu=True)
import json; print(json.dumps({k: r[k] for k in ('pose_graph', 'pose_only', 'lk') if k in r}))
" > gpurun_out/r6f/next.txt 2>&1; tail -c 3000 gpurun_out/r6f/next.txt
