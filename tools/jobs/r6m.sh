mkdir -p gpurun_out/r6m
SSX_BA_TIMING=1 timeout 600 python tools/ba_c4_slope.py > gpurun_out/r6m/c4.txt 2>&1; tail -40 gpurun_out/r6m/c4.txt | cut -c1-600
timeout 600 python -c "
import sys; sys.path.insert(0,'.')
import ssvio_amd
from tools import bench_next
ctx = ssvio_amd.Context(0)
r = bench_next.next_rows(ssvio_amd, ctx, cpu=False)
import json; print(json.dumps({k: {kk: vv for kk, vv in r[k].items() if kk in ('ms_per_call','ms_per_solve','ms_per_call_chained','hbm_frac','flops_frac','batched')} for k in ('pose_only','lk')}))
" > gpurun_out/r6m/next.txt 2>&1; tail -c 1500 gpurun_out/r6m/next.txt
