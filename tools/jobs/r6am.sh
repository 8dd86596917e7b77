R=$PWD; O=$R/gpurun_out/r6am; mkdir -p $O
export TMPDIR=/tmp
(timeout 1800 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log); tail -4 $O/tests.log
for i in 1 2; do python tools/lk_ab.py 2>&1 | tail -1; SSX_LIB=$PWD/ssvio_amd/libssx.so.base python tools/lk_ab.py 2>&1 | tail -1 | sed 's/^/[base] /'; done | tee $O/lk_ab.txt
bash tools/collect_profiles.sh r6v2 bench_v2 r06 > $O/collect.log 2>&1; tail -12 $O/collect.log
cp -r gpurun_out/prof/r6v2/summary $O/summary 2>/dev/null; cp gpurun_out/prof/r6v2/bench.json $O/bench.json; cp gpurun_out/prof/r6v2/bench.err $O/bench.err
