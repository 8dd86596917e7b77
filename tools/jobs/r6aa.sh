O=gpurun_out/r6aa; mkdir -p $O
export TMPDIR=/tmp
T0=$(date +%s); timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5; echo "gpu tests wall $(( $(date +%s) - T0 )) s"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
T0=$(date +%s); timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench wall $(( $(date +%s) - T0 )) s"; tail -3 $O/bench.err | cut -c1-300
python - <<'P'
import json
j=json.loads([l for l in open('gpurun_out/r6aa/bench.json') if l.startswith('{')][-1])
print('value', j['value'], j['ms_per_step'], 'frozen', j['frozen_batch']['value'])
c1=j['c1']; print('c1', {k:c1['resident_window'].get(k) for k in ('warmup_ms_outside_the_frame_loop','frames_per_s_runstep_only','ms_per_keyframe_insert_and_ba','frames_per_s_incl_png_decode')}, 'ident', c1['trajectories_identical'])
print('c5', {k:(v['frames_per_s'], v['every_stream_byte_identical_to_the_single_stream_run']) for k,v in j['c5']['runs'].items()}, 'unbatched8', j['c5']['unbatched_eight_streams'])
print('hard', {k:c1['hard_drive'].get(k) for k in ('frames_per_s_runstep_only','ba_windows','edges_per_window','ms_per_keyframe_insert_and_ba')})
print('oracle', c1['cpu_oracle_runner']['frames_per_s_runstep_only'], c1['speedup_vs_cpu_oracle_runstep'])
P
