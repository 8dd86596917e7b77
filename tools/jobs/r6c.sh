R=$PWD; O=$R/gpurun_out/r6c; mkdir -p $O
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O2 -o /tmp/copy_engine tools/microbench/copy_engine.hip
cd /tmp
for v in "X=0" "GPU_FORCE_BLIT_COPY_SIZE=0" "GPU_FORCE_BLIT_COPY_SIZE=100000"; do
  rm -rf /tmp/ce; env $v timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/ce -- /tmp/copy_engine 2> /tmp/ce.cases > /dev/null
  echo "== $v" >> $O/copy_engine.txt; python $R/tools/microbench/copy_engine_summary.py /tmp/ce /tmp/ce.cases >> $O/copy_engine.txt 2>&1
done
cd $R; cat $O/copy_engine.txt
timeout 1200 python -m pytest tests/test_ba_gpu.py tests/test_host_gpu.py -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -4 $O/tests.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -- python $R/bench.py --lean --steps 12 --warmup 3 > $O/trace.log 2>&1
cd $R
python tools/live_busy.py $(ls $O/trace/*/*kernel_trace.csv | tail -1) > $O/live_busy.txt 2>&1
python tools/live_copies.py $O/trace > $O/live_copies.txt 2>&1
cat $O/live_busy.txt; grep -v "columns" $O/live_copies.txt
rm -rf $O/trace
for i in 1 2; do timeout 600 python bench.py --lean --steps 20 --warmup 4 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('lean', j['value'], j['ms_per_step'], j['live_backend']['ms_per_step_inside_solve_calls'], j['frozen_batch']['value'])"; done
