O=gpurun_out/r6x; mkdir -p $O
export TMPDIR=/tmp
python tools/c1_stages.py run default 2>&1 | tee $O/stages_default.txt
python tools/c1_stages.py run hard 2>&1 | tee $O/stages_hard.txt
