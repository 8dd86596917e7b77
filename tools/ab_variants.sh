# A/B of library variants inside one gpurun call: the resident 128-window C3 batch, one group, per-kernel HIP events.
#   bash tools/ab_variants.sh base occ4 lds4 ...     (ssvio_amd/libssx.so.<name>; "new" = the regular libssx.so)
for rep in 1 2; do
  for v in "$@"; do
    if [ $v = new ]; then unset SSX_LIB; else export SSX_LIB=$PWD/ssvio_amd/libssx.so.$v; fi
    SSX_BA_GROUPS=1 python tools/ba_batch_time.py 128 5 2>/dev/null | grep -E "per batch|k_lin_schur|k_backsub|k_schur |k_linearize|k_reduce|k_solve" | sed "s/^/[$v] /"
  done
done
unset SSX_LIB
