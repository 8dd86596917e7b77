R=$PWD; O=$R/gpurun_out/r6b; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ba_gpu.py -x -q -k "busy_chip or litmus or block_cyclic or c4_shape or band_solver or deterministic or full_size" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -4 $O/tests.log
bash tools/jobs/litmus_O1.sh 16 > $O/litmus_O1.log 2>&1; tail -3 $O/litmus_O1.log
hipcc --offload-arch=gfx950 -O2 -o /tmp/copy_engine tools/microbench/copy_engine.hip
AMD_LOG_LEVEL=4 AMD_LOG_MASK=0x380 timeout 300 /tmp/copy_engine 2> /tmp/ce.log; python tools/microbench/copy_engine_summary.py /tmp/ce.log > $O/copy_engine.txt 2>&1
grep -m 12 -n "Copy\|copyBuffer\|Blit" /tmp/ce.log | cut -c1-300 >> $O/copy_engine.txt
wc -l /tmp/ce.log >> $O/copy_engine.txt
cat $O/copy_engine.txt
for v in "GPU_FORCE_BLIT_COPY_SIZE=0" "GPU_BLIT_ENGINE_TYPE=1" "GPU_BLIT_ENGINE_TYPE=2"; do
  echo "== $v" >> $O/copy_engine_env.txt
  env $v AMD_LOG_LEVEL=4 AMD_LOG_MASK=0x380 timeout 300 /tmp/copy_engine 2> /tmp/ce2.log; python tools/microbench/copy_engine_summary.py /tmp/ce2.log >> $O/copy_engine_env.txt 2>&1
done
cat $O/copy_engine_env.txt | head -150
