O=gpurun_out/r6aw; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python tools/fuzz_parity.py 611 40 > $O/fuzz_parity_611.log 2>&1; echo "fuzz_parity rc=$?" | tee -a $O/fuzz_parity_611.log; tail -2 $O/fuzz_parity_611.log
timeout 1500 python tools/fuzz_parity2.py 612 40 > $O/fuzz_parity2_612.log 2>&1; echo "fuzz_parity2 rc=$?" | tee -a $O/fuzz_parity2_612.log; tail -2 $O/fuzz_parity2_612.log
timeout 1500 python tools/fuzz_batched.py 613 20 > $O/fuzz_batched_613.log 2>&1; echo "fuzz_batched rc=$?" | tee -a $O/fuzz_batched_613.log; tail -2 $O/fuzz_batched_613.log
