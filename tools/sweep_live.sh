mkdir -p gpurun_out/r5d
run() { tag=$1; shift; env "$@" python bench.py --steps 20 --warmup 4 --lean > gpurun_out/r5d/$tag.json 2> gpurun_out/r5d/$tag.err; python -c "
import json,sys
try:
    d=json.load(open('gpurun_out/r5d/$tag.json')); lb=d['live_backend']; print('$tag', d['value'], d['ms_per_step'], 'solve',lb['ms_per_step_inside_solve_calls'],'upd',lb['ms_per_step_inside_update_calls'],'frozen',d['frozen_batch']['value'], 'fe', d['frontend']['value'])
except Exception as e: print('$tag', e, open('gpurun_out/r5d/$tag.err').read()[-300:])
"; }
run q4_t3_g1 SSX_BENCH_LAG=2 SSX_BENCH_WINDOW_THREADS=3 SSX_BA_GROUPS=1
run q2_t3_g1 GPU_MAX_HW_QUEUES=2 SSX_BENCH_LAG=2 SSX_BENCH_WINDOW_THREADS=3 SSX_BA_GROUPS=1
run q3_t3_g1 GPU_MAX_HW_QUEUES=3 SSX_BENCH_LAG=2 SSX_BENCH_WINDOW_THREADS=3 SSX_BA_GROUPS=1
run q1_t3_g1 GPU_MAX_HW_QUEUES=1 SSX_BENCH_LAG=2 SSX_BENCH_WINDOW_THREADS=3 SSX_BA_GROUPS=1
run q2_t2_g1 GPU_MAX_HW_QUEUES=2 SSX_BENCH_LAG=2 SSX_BENCH_WINDOW_THREADS=2 SSX_BA_GROUPS=1
run q2_t4_g1 GPU_MAX_HW_QUEUES=2 SSX_BENCH_LAG=2 SSX_BENCH_WINDOW_THREADS=4 SSX_BA_GROUPS=1
run q4_t3_g1_b SSX_BENCH_LAG=2 SSX_BENCH_WINDOW_THREADS=3 SSX_BA_GROUPS=1
