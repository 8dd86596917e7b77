O=gpurun_out/r6ar; mkdir -p $O
export TMPDIR=/tmp
SSX_LIB=$PWD/ssvio_amd/libssx.so.prev python tools/solve64_ab.py /tmp/prev.npz 2>&1 | tail -1 | sed 's/^/[one pivot per barrier] /' | tee $O/ab.txt
python tools/solve64_ab.py /tmp/new.npz 2>&1 | tail -1 | sed 's/^/[two pivots per barrier] /' | tee -a $O/ab.txt
python tools/solve64_ab.py --compare /tmp/prev.npz /tmp/new.npz | tee -a $O/ab.txt
for i in 1 2; do
SSX_LIB=$PWD/ssvio_amd/libssx.so.prev python tools/one_window_time.py 2>&1 | grep "True" | sed 's/^/[one] /'
python tools/one_window_time.py 2>&1 | grep "True" | sed 's/^/[two] /'
done | tee -a $O/ab.txt
for v in prev new; do
  if [ $v = new ]; then unset SSX_LIB; else export SSX_LIB=$PWD/ssvio_amd/libssx.so.prev; fi
  SSX_BA_GROUPS=1 python tools/ba_batch_time.py 128 5 2>/dev/null | grep -E "per batch|k_solve" | sed "s/^/[$v] /"
done | tee -a $O/ab.txt
unset SSX_LIB
timeout 1800 python -m pytest tests/test_ba_gpu.py tests/test_host_gpu.py tests/test_dist_ba_gpu.py tests/test_track_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/tests.txt
