O=gpurun_out/r6ap; mkdir -p $O
export TMPDIR=/tmp
run() { python bench.py --lean --steps 30 --warmup 5 2>$O/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); l=d['live_backend']
print('$1', 'value', d['value'], 'ms/step', d['ms_per_step'], 'solve', l['ms_per_step_inside_solve_calls'], 'update', l['ms_per_step_inside_update_calls'], 'frozen', d['frozen_batch']['value'])" || tail -3 $O/err.txt; }
for rep in 1 2; do
  for T in 2 3 4 6 8; do
    for L in 2 3; do
      export SSX_BENCH_WINDOW_THREADS=$T SSX_BENCH_LAG=$L; run "groups $T lag $L"
    done
  done
done 2>&1 | tee $O/groups_sweep.txt
