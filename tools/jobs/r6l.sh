R=$PWD; O=$R/gpurun_out/r6l; mkdir -p $O
run() { env "$@" timeout 600 python bench.py --lean --steps 16 --warmup 4 $PAIRS 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('value', j['value'], 'ms/step', j['ms_per_step'], 'solve', j['live_backend']['ms_per_step_inside_solve_calls'], 'frozen', j['frozen_batch']['value'], 'fe', j['frontend']['value'])"; }
for rep in 1 2; do
PAIRS="--pairs 128"; echo "== 128 T3 L2"; run X=1
PAIRS="--pairs 256"; echo "== 256 T3 L2"; run X=1
PAIRS="--pairs 256"; echo "== 256 T4 L2"; run SSX_BENCH_WINDOW_THREADS=4
PAIRS="--pairs 256"; echo "== 256 T3 L3"; run SSX_BENCH_LAG=3
PAIRS="--pairs 192"; echo "== 192 T3 L2"; run X=1
done 2>&1 | tee $O/pairs.txt
