R=$PWD; O=$R/gpurun_out/r6i; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_host_gpu.py -x -q -s -k "hard_drive or batched_streams or several_streams" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; grep -E "^\[hard|passed|failed|Error|assert|rc=" $O/tests.log | cut -c1-500 | tail -20
timeout 900 python tools/c5_time.py 200 32 64 128 > $O/c5.txt 2>&1; cat $O/c5.txt
timeout 900 python -m pytest tests/test_dist_ba_gpu.py -x -q > $O/dist.log 2>&1; tail -3 $O/dist.log
