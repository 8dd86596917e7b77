R=$PWD; O=$R/gpurun_out/r6h; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_orb_gpu.py tests/test_lk_gpu.py tests/test_host_gpu.py tests/test_pg_gpu.py tests/test_abi.py -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -8 $O/tests.log
timeout 1500 python tools/c5_time.py 200 8 32 64 128 > $O/c5.txt 2>&1; cat $O/c5.txt
timeout 900 python -m pytest tests/test_ba_gpu.py -x -q -s -k "reference_golden or window_drive_matches" 2>&1 | grep -E "^\[|passed|failed|Error|assert" | cut -c1-400 > $O/noise.txt; tail -40 $O/noise.txt
timeout 1200 python tools/c1_hard_explore.py 120 > $O/hard.txt 2>&1; cat $O/hard.txt
