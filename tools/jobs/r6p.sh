O=gpurun_out/r6p; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2 3 4 5 6; do timeout 900 python -m pytest tests/test_host_gpu.py -x -q -k "batched_streams or several_streams or eight_concurrent" > $O/streams_$i.log 2>&1; tail -1 $O/streams_$i.log; done
for i in 1 2 3; do timeout 600 python -m pytest tests/test_ba_gpu.py -x -q -k "litmus or busy_chip" > $O/thr_$i.log 2>&1; tail -1 $O/thr_$i.log; done
(timeout 1500 python -m pytest tests -m gpu -x -q > $O/full.log 2>&1; echo "full rc=$?" >> $O/full.log); tail -3 $O/full.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
