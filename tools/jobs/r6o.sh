mkdir -p gpurun_out/r6o
(timeout 900 python tools/fuzz_parity.py 81 40 > gpurun_out/r6o/fuzz1.log 2>&1; tail -3 gpurun_out/r6o/fuzz1.log) 
(timeout 900 python tools/fuzz_parity2.py 82 40 > gpurun_out/r6o/fuzz2.log 2>&1; tail -3 gpurun_out/r6o/fuzz2.log)
(timeout 1200 python tools/fuzz_runner.py 83 8 > gpurun_out/r6o/fuzz3.log 2>&1; tail -4 gpurun_out/r6o/fuzz3.log)
SSX_BA_TIMING=1 timeout 300 python tools/ba_c4_time.py 2>&1 | grep -E "prepare|iters/s" | sed -n '2,3p' | cut -c1-260
