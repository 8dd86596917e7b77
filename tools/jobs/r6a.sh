R=$PWD; O=$R/gpurun_out/r6a; mkdir -p $O
export TMPDIR=/tmp
(timeout 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log)
tail -3 $O/tests.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -- python $R/bench.py --lean --steps 12 --warmup 3 > $O/trace.log 2>&1
cd $R
python tools/live_busy.py $(ls $O/trace/*/*kernel_trace.csv | tail -1) > $O/live_busy.txt 2>&1
python tools/live_copies.py $O/trace > $O/live_copies.txt 2>&1
cat $O/live_busy.txt $O/live_copies.txt
rm -rf $O/trace
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
python - <<'P'
import json
l=[x for x in open('gpurun_out/r6a/bench.json') if x.startswith('{')][-1]; j=json.loads(l)
print({k:j[k] for k in ('value','ms_per_step')}); print(j['extra_keys'] if 'extra_keys' in j else '')
print(json.dumps(j.get('live_backend'))[:600])
P
