mkdir -p gpurun_out/r6d
bash tools/ab_variants.sh base occ4 lds4 occ4s > gpurun_out/r6d/ab.txt 2>&1
cat gpurun_out/r6d/ab.txt
