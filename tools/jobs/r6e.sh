R=$PWD; O=$R/gpurun_out/r6e; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_lk_gpu.py tests/test_ba_gpu.py tests/test_host_gpu.py tests/test_track_gpu.py -x -q -k "lk or pose_only or streams or runner or track or batched" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -15 $O/tests.log
