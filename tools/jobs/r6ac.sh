O=gpurun_out/r6ac; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python tools/fuzz_batched.py 2 10 2>&1 | tail -14 | tee $O/fuzz_batched_2.txt
timeout 1500 python tools/fuzz_batched.py 3 10 2>&1 | tail -14 | tee $O/fuzz_batched_3.txt
timeout 900 python -m pytest tests/test_host_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
