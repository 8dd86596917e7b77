R=$PWD; O=$R/gpurun_out/r6aq; mkdir -p $O
export TMPDIR=/tmp
(timeout 1800 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log); grep -E "passed|failed|rc=" $O/tests.log | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
for i in 1 2; do python tools/one_window_time.py 2>&1 | grep wall; done | tee $O/one_window.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.json; echo
timeout 900 python tools/fuzz_parity.py 601 12 > $O/fuzz_parity_601.log 2>&1; echo "fuzz_parity rc=$?" | tee -a $O/fuzz_parity_601.log; tail -3 $O/fuzz_parity_601.log
timeout 900 python tools/fuzz_parity2.py 602 12 > $O/fuzz_parity2_602.log 2>&1; echo "fuzz_parity2 rc=$?" | tee -a $O/fuzz_parity2_602.log; tail -3 $O/fuzz_parity2_602.log
timeout 900 python tools/fuzz_batched.py 603 10 > $O/fuzz_batched_603.log 2>&1; echo "fuzz_batched rc=$?" | tee -a $O/fuzz_batched_603.log; tail -3 $O/fuzz_batched_603.log
timeout 900 python tools/fuzz_runner.py 604 8 > $O/fuzz_runner_604.log 2>&1; echo "fuzz_runner rc=$?" | tee -a $O/fuzz_runner_604.log; tail -4 $O/fuzz_runner_604.log
