O=gpurun_out/r6s; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_lk_gpu.py tests/test_host_gpu.py -x -q -k "lk or batched_streams or several or runner_matches" > $O/tests.log 2>&1; tail -2 $O/tests.log
timeout 1500 python tools/c5_time.py 200 32 64 128 > $O/c5.txt 2>&1; grep -v "^      " $O/c5.txt | cut -c1-220; grep "batched time" $O/c5.txt | cut -c1-120
