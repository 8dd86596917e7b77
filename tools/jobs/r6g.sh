mkdir -p gpurun_out/r6g
timeout 1500 python tools/c5_time.py 200 32 64 > gpurun_out/r6g/c5.txt 2>&1; cat gpurun_out/r6g/c5.txt
