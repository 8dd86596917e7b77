O=gpurun_out/r6aj; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace -- python $R/bench.py --lean --steps 12 --warmup 3 > $R/$O/bench_traced.json 2> $R/$O/bench_traced.err)
python tools/live_busy.py $(ls $O/trace/*/*kernel_trace.csv | tail -1) > $O/live_busy.txt 2>&1
python tools/live_gaps.py $(ls $O/trace/*/*kernel_trace.csv | tail -1) > $O/live_gaps.txt 2>&1
cat $O/live_busy.txt | head -3; cat $O/live_gaps.txt
rm -rf $O/trace
SSX_WIN_TIMING=1 python bench.py --lean --steps 12 --warmup 3 > $O/bench_timing.json 2> $O/win_timing.err
python - <<'P'
import re,collections
rows=[l for l in open('gpurun_out/r6aj/win_timing.err') if 'ssx_ba_window_solve_batch' in l]
acc=collections.defaultdict(list)
for l in rows:
    m=re.search(r"n=(\d+).*sync ([\d.]+) ms, views \+ marshal \+ upload ([\d.]+) ms, solve \+ download ([\d.]+) ms, unpack ([\d.]+)", l)
    if m: acc[int(m.group(1))].append([float(x) for x in m.groups()[1:]])
for n,v in acc.items():
    import numpy as np
    v=np.array(v[len(v)//3:])
    print('n',n,'calls',len(v),'mean ms: sync %.3f views+marshal+upload %.3f solve+download %.3f unpack %.3f'%tuple(v.mean(0)))
P
