O=gpurun_out/r6z; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_host_gpu.py -x -q -m gpu 2>&1 | tail -8
python tools/c1_stages.py run default 2>&1 | grep -v "^wrote\|^rendered" | tee $O/stages_default.txt
python tools/c1_stages.py run hard 2>&1 | grep -v "^wrote\|^rendered" | tee $O/stages_hard.txt
timeout 600 python tools/c5_time.py 200 8 64 2>&1 | tail -8 | tee $O/c5.txt
