O=gpurun_out/r6al; mkdir -p $O
export TMPDIR=/tmp
run() { python bench.py --lean --steps 30 --warmup 5 2>$O/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); l=d['live_backend']
print('$1', 'value', d['value'], 'ms/step', d['ms_per_step'], 'solve', l['ms_per_step_inside_solve_calls'], 'frozen', d['frozen_batch']['value'])" || tail -3 $O/err.txt; }
for rep in 1 2; do
  unset SSX_BENCH_GROUP_PRIO; run "prio none      "
  export SSX_BENCH_GROUP_PRIO="-1,0,1"; run "prio -1,0,1   "
  export SSX_BENCH_GROUP_PRIO="1,0,-1"; run "prio 1,0,-1   "
  export SSX_BENCH_GROUP_PRIO="-1,-1,0"; run "prio -1,-1,0  "
  export SSX_BENCH_GROUP_PRIO="0,1,1"; run "prio 0,1,1    "
  export SSX_BENCH_GROUP_PRIO="1,1,1"; run "prio 1,1,1    "
  export SSX_BENCH_GROUP_PRIO="-1,-1,-1"; run "prio -1,-1,-1 "
done 2>&1 | tee $O/prio_sweep.txt
