O=gpurun_out/r6v; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_misuse_gpu.py tests/test_host_gpu.py -x -q -m gpu 2>&1 | tail -15
timeout 600 python tools/c5_time.py 2>&1 | tail -12
