mkdir -p gpurun_out/r6n
for t in 8 16 32; do echo "== SSX_BA_PREP_THREADS=$t"; SSX_BA_PREP_THREADS=$t SSX_BA_TIMING=1 timeout 600 python tools/ba_c4_time.py 2>&1 | grep -E "prepare|iters/s" | sed -n '2,4p' | cut -c1-260; done > gpurun_out/r6n/c4_threads.txt 2>&1
cat gpurun_out/r6n/c4_threads.txt
