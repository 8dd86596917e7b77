O=gpurun_out/r6af; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_lk_gpu.py tests/test_track_gpu.py -x -q -m gpu 2>&1 | tail -5
for i in 1 2; do python tools/lk_ab.py 2>&1 | tail -1; SSX_LK_UNFUSED=1 python tools/lk_ab.py 2>&1 | tail -1; done | tee $O/lk_ab.txt
timeout 900 python -m pytest tests/test_host_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
timeout 600 python tools/c5_time.py 200 8 64 2>&1 | tail -8 | tee $O/c5.txt
