O=gpurun_out/r6ad; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_lk_gpu.py tests/test_track_gpu.py tests/test_misuse_gpu.py -x -q -m gpu 2>&1 | tail -15
timeout 900 python -m pytest tests/test_host_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
python tools/c1_stages.py run default 2>&1 | grep -v "^wrote\|^rendered" | tee $O/stages_default_fused.txt
SSX_LK_UNFUSED=1 python tools/c1_stages.py run default 2>&1 | grep -v "^wrote\|^rendered" | tee $O/stages_default_unfused.txt
python tools/c1_stages.py run hard 2>&1 | grep -v "^wrote\|^rendered" | tee $O/stages_hard_fused.txt
timeout 600 python tools/c5_time.py 200 8 64 2>&1 | tail -8 | tee $O/c5.txt
