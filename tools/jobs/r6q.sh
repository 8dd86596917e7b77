mkdir -p gpurun_out/r6q
timeout 1500 python tools/c5_time.py 200 32 64 128 > gpurun_out/r6q/c5.txt 2>&1; grep -v "batched time" gpurun_out/r6q/c5.txt | cut -c1-200
