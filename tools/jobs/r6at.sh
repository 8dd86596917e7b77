O=gpurun_out/r6at; mkdir -p $O
export TMPDIR=/tmp
run() { python bench.py --lean --steps 30 --warmup 5 2>$O/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); l=d['live_backend']
print('$1', 'value', d['value'], 'ms/step', d['ms_per_step'], 'solve', l['ms_per_step_inside_solve_calls'], 'update', l['ms_per_step_inside_update_calls'], 'frozen', d['frozen_batch']['value'])" || tail -3 $O/err.txt; }
for rep in 1 2; do
  unset SSX_BENCH_PAIR_STREAMS; export SSX_BENCH_WINDOW_THREADS=3 SSX_BENCH_LAG=2; run "3 groups, 3 streams (default)   "
  export SSX_BENCH_PAIR_STREAMS=2 SSX_BENCH_WINDOW_THREADS=6 SSX_BENCH_LAG=2; run "6 groups paired on 3 streams    "
  export SSX_BENCH_PAIR_STREAMS=2 SSX_BENCH_WINDOW_THREADS=6 SSX_BENCH_LAG=3; run "6 groups paired, lag 3          "
  export SSX_BENCH_PAIR_STREAMS=2 SSX_BENCH_WINDOW_THREADS=4 SSX_BENCH_LAG=2; run "4 groups paired on 2 streams    "
  export SSX_BENCH_PAIR_STREAMS=3 SSX_BENCH_WINDOW_THREADS=6 SSX_BENCH_LAG=2; run "6 groups in threes on 2 streams "
  export SSX_BENCH_PAIR_STREAMS=2 SSX_BENCH_WINDOW_THREADS=8 SSX_BENCH_LAG=2; run "8 groups paired on 4 streams    "
  export SSX_BENCH_PAIR_STREAMS=3 SSX_BENCH_WINDOW_THREADS=9 SSX_BENCH_LAG=2; run "9 groups in threes on 3 streams "
done 2>&1 | tee $O/pair_streams.txt
