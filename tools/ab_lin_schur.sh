# A/B of libssx.so against ssvio_amd/libssx.so.base (kept by hand): the resident 128-window C3 batch, one group, per-kernel HIP events
mkdir -p gpurun_out/ab
for rep in 1 2; do
  for v in base new; do
    if [ $v = base ]; then export SSX_LIB=$PWD/ssvio_amd/libssx.so.base; else unset SSX_LIB; fi
    SSX_BA_GROUPS=1 python tools/ba_batch_time.py 128 5 2>/dev/null | grep -E "per batch|k_lin_schur|k_backsub|k_schur |k_linearize|k_reduce|k_solve" | sed "s/^/[$v] /"
  done
done
unset SSX_LIB
