O=gpurun_out/r6an; mkdir -p $O
export TMPDIR=/tmp
run() { python bench.py --lean --steps 30 --warmup 5 2>$O/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); l=d['live_backend']
print('$1', 'value', d['value'], 'ms/step', d['ms_per_step'], 'solve', l['ms_per_step_inside_solve_calls'], 'frozen', d['frozen_batch']['value'], 'fe', d['frontend']['value'])" || tail -3 $O/err.txt; }
for rep in 1 2 3; do
  unset SSX_BENCH_GROUP_PRIO SSX_BENCH_FE_PRIO; run "fe default, groups default "
  export SSX_BENCH_FE_PRIO=1; run "fe low,     groups default "
  export SSX_BENCH_GROUP_PRIO="-1,-1,-1"; run "fe low,     groups high    "
  unset SSX_BENCH_FE_PRIO; run "fe default, groups high    "
  export SSX_BENCH_FE_PRIO=-1; export SSX_BENCH_GROUP_PRIO="1,1,1"; run "fe high,    groups low     "
done 2>&1 | tee $O/fe_prio_sweep.txt
