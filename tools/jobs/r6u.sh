R=$PWD; O=$R/gpurun_out/r6u; mkdir -p $O
run() { env "$@" timeout 600 python bench.py --lean --steps 16 --warmup 4 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('value', j['value'], 'ms/step', j['ms_per_step'], 'solve', j['live_backend']['ms_per_step_inside_solve_calls'], 'frozen', j['frozen_batch']['value'], 'fe', j['frontend']['value'])"; }
for rep in 1 2; do
echo "== T3 default"; run X=1
echo "== T3 NO_FORK"; run SSX_ORB_NO_FORK=1
echo "== T2 NO_FORK"; run SSX_ORB_NO_FORK=1 SSX_BENCH_WINDOW_THREADS=2
echo "== T2 NO_FORK g2"; run SSX_ORB_NO_FORK=1 SSX_BENCH_WINDOW_THREADS=2 SSX_BENCH_BATCH_GROUPS=2
echo "== T4 NO_FORK"; run SSX_ORB_NO_FORK=1 SSX_BENCH_WINDOW_THREADS=4
echo "== T3 HWQ6"; run GPU_MAX_HW_QUEUES=6
done 2>&1 | tee $O/streams.txt
