O=gpurun_out/r6y; mkdir -p $O
export TMPDIR=/tmp
python tools/window_grow_time.py 300 14 2>&1 | tee $O/grow_300.txt
python tools/window_grow_time.py 2400 14 2>&1 | tee $O/grow_2400.txt
