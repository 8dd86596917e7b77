O=gpurun_out/r6ah; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_orb_gpu.py tests/test_track_gpu.py -x -q -m gpu 2>&1 | tail -8 | tee $O/tests.txt
for rep in 1 2; do
  bash tools/fe_kernels.sh base
  bash tools/fe_kernels.sh ""
  SSX_ORB_RESIZE_PAIRS=0 bash tools/fe_kernels.sh ""
done 2>&1 | tee $O/fe_ab.txt
for rep in 1 2; do
  SSX_ORB_NO_FORK=1 bash tools/fe_kernels.sh base
  SSX_ORB_NO_FORK=1 bash tools/fe_kernels.sh ""
  SSX_ORB_NO_FORK=1 SSX_ORB_RESIZE_PAIRS=0 bash tools/fe_kernels.sh ""
done 2>&1 | tee $O/fe_ab_nofork.txt
