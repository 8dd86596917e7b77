R=$PWD; O=$R/gpurun_out/r6j; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_host_gpu.py -x -q -s -k "hard_drive" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; grep -E "^\[hard|passed|failed|Error|assert|rc=" $O/tests.log | cut -c1-500 | tail -20
