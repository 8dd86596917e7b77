O=gpurun_out/r6ab; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python tools/fuzz_batched.py 1 8 2>&1 | tail -12 | tee $O/fuzz_batched_1.txt
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
