O=gpurun_out/r6t; mkdir -p $O
export TMPDIR=/tmp
echo tests-skipped
T0=$(date +%s); timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench wall $(( $(date +%s) - T0 )) s"; tail -3 $O/bench.err | cut -c1-300
python - <<'P'
import json
j=json.loads([l for l in open('gpurun_out/r6t/bench.json') if l.startswith('{')][-1])
print('value', j['value'], j['ms_per_step']); print('c5', {k:(v['frames_per_s'], v['every_stream_byte_identical_to_the_single_stream_run']) for k,v in j['c5']['runs'].items()})
print('hard', {k:j['c1']['hard_drive'].get(k) for k in ('frames_per_s_runstep_only','ba_windows','edges_per_window')})
print('c4', j['ba_c4']['full_configs3']['ms_per_lm_iteration'], j['ba_c4']['full_configs3'].get('iters_per_s'))
print('next', {k:(v.get('ms_per_solve') or v.get('ms_per_call'), v.get('batched',{}).get('flops_frac') or v.get('batched',{}).get('hbm_frac')) for k,v in j['next_rows'].items()})
P
