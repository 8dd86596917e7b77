R=$PWD; O=$R/gpurun_out/r6k; mkdir -p $O
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log); tail -4 $O/tests.log
bash tools/collect_profiles.sh r6v1 bench_v1 r06 > $O/collect.log 2>&1; tail -30 $O/collect.log
cp -r gpurun_out/prof/r6v1/summary $O/summary 2>/dev/null; cp gpurun_out/prof/r6v1/bench.json $O/bench.json; cp gpurun_out/prof/r6v1/bench.err $O/bench.err
