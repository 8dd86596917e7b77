"""One line of the bench's JSON: python bench.py ... | python tools/bench_brief.py [ba|fe|host]"""
import json, sys
what = sys.argv[1] if len(sys.argv) > 1 else "ba"
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d.get("kernels", {})
out = [f"value {d['value']:.0f}", f"{d['ms_per_step']:.3f} ms/step", f"fe {d['frontend']['value']:.0f}", f"ba {d['ba']['batched']['ms_per_call']:.3f} ms/call"]
if what in ("ba", "fe"):
    part = "ba" if what == "ba" else "frontend"
    out += [f"{n}={v['ms_per_step']:.3f}" for n, v in k.items() if v["part"] == part]
if what == "host":
    h = d["host_buffers_inclusive"]; r = h.get("resident_windows_one_keyframe_replaced_per_step") or {}
    out += [f"host {h['value']}", f"two {h['two_batches_in_flight']}", f"windows {r.get('value')} ({r.get('ms_per_step')} ms/step, solve {r.get('ms_per_step_inside_solve_calls')}, update {r.get('ms_per_step_inside_update_calls')})",
            f"one window {d['ba']['one_window']['ms_per_solve']} ms", f"C4 {d['ba_c4']['full_configs3']['iters_per_s']}"]
print(" ".join(out))
