O=gpurun_out/r6au; mkdir -p $O
export TMPDIR=/tmp
(timeout 1800 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log); grep -E "passed|failed|rc=" $O/tests.log | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; python -c "
import json
d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print('value', d['value'], 'ms/step', d['ms_per_step'], 'one window', d['ba']['one_window']['ms_per_solve'], 'c4', d['ba_c4']['full_configs3']['iters_per_s'], 'c5', d['c1']['c5_batched_streams']['value'])"
