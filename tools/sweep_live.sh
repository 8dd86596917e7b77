mkdir -p gpurun_out/r5d
run() { tag=$1; shift; env "$@" python bench.py --steps 20 --warmup 3 --lean > gpurun_out/r5d/$tag.json 2> gpurun_out/r5d/$tag.err; python -c "
import json,sys
try:
    d=json.load(open('gpurun_out/r5d/$tag.json')); lb=d['live_backend']; print('$tag', d['value'], d['ms_per_step'], 'solve',lb['ms_per_step_inside_solve_calls'],'upd',lb['ms_per_step_inside_update_calls'],'frozen',d['frozen_batch']['value'])
except Exception as e: print('$tag', e, open('gpurun_out/r5d/$tag.err').read()[-300:])
"; }
run t2 SSX_BENCH_WINDOW_THREADS=2
run t3 SSX_BENCH_WINDOW_THREADS=3
run t4 SSX_BENCH_WINDOW_THREADS=4
run t2_noturn SSX_BENCH_WINDOW_THREADS=2 SSX_BENCH_NO_TURNS=1
run t2_h32 SSX_BENCH_WINDOW_THREADS=2 SSX_HOST_THREADS=32
run t2_h8 SSX_BENCH_WINDOW_THREADS=2 SSX_HOST_THREADS=8
run t3_h8 SSX_BENCH_WINDOW_THREADS=3 SSX_HOST_THREADS=8
run t1_h64 SSX_BENCH_WINDOW_THREADS=1 SSX_HOST_THREADS=64
