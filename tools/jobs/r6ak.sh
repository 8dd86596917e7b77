O=gpurun_out/r6ak; mkdir -p $O
export TMPDIR=/tmp
SSX_WIN_TIMING=1 SSX_BATCH_TIMING=2 python bench.py --lean --steps 12 --warmup 3 > $O/bench_timing.json 2> $O/timing.err
grep -c . $O/timing.err
grep "win_sync_many n=4" $O/timing.err | tail -6
grep "batch_build n=4" $O/timing.err | tail -6
grep "ssx_ba_window_solve_batch n=4" $O/timing.err | tail -6
grep -v "win_sync_many\|batch_build\|ssx_ba_window_solve_batch" $O/timing.err | sort | uniq -c | sort -rn | head -8
