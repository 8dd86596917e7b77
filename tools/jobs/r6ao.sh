O=gpurun_out/r6ao; mkdir -p $O
export TMPDIR=/tmp
run() { python bench.py --lean --steps 30 --warmup 5 2>$O/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); l=d['live_backend']
print('$1', 'value', d['value'], 'ms/step', d['ms_per_step'], 'solve', l['ms_per_step_inside_solve_calls'], 'update', l['ms_per_step_inside_update_calls'], 'frozen', d['frozen_batch']['value'])" || tail -3 $O/err.txt; }
for rep in 1 2 3; do
  export SSX_LIB=$PWD/ssvio_amd/libssx.so.base; run "base pool        "
  unset SSX_LIB; run "new pool         "
  export SSX_POOL_SPIN=0; run "new pool, no spin"; unset SSX_POOL_SPIN
done 2>&1 | tee $O/pool_ab.txt
SSX_WIN_TIMING=1 SSX_BATCH_TIMING=2 python bench.py --lean --steps 12 --warmup 3 > /dev/null 2> $O/timing.err
grep "win_sync_many n=43" $O/timing.err | tail -3; grep "batch_build n=43" $O/timing.err | tail -3; grep "ssx_ba_window_solve_batch n=43" $O/timing.err | tail -3
bash tools/jobs/r6an.sh
timeout 1500 python -m pytest tests/test_ba_gpu.py tests/test_host_gpu.py tests/test_track_gpu.py -x -q -m gpu 2>&1 | tail -3
