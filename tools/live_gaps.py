"""Where the GPU idles in the headline region: from a rocprofv3 kernel trace of `bench.py --lean`, every gap of the union of kernel
intervals between the first and last k_win_scatter of the timed half, attributed to the kernel that ENDS it (what the chip was waiting for)
and to the kernel that ran last before it.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python bench.py --lean --steps 12 --warmup 3
    python tools/live_gaps.py [kernel_trace.csv] [min_gap_us]"""
import collections, csv, glob, re, sys

path = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob("gpurun_out/trace/*/*kernel_trace.csv"))[-1]
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
rows = list(csv.DictReader(open(path)))
name = lambda r: (re.search(r"(k_\w+|__amd_rocclr_\w+)", r["Kernel_Name"]) or re.search(r"(\w+)", r["Kernel_Name"])).group(1)
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name(r), r.get("Queue_Id", "")) for r in rows), key=lambda e: e[0])
ws = [i for i, e in enumerate(ev) if e[2] == "k_win_scatter"]
a, b = ws[len(ws) // 2], ws[-1]
t0, t1 = ev[a][0], ev[b][0]
sel = [e for e in ev if t0 <= e[0] < t1]
span = (t1 - t0) / 1e3
by_next, by_prev, hist = collections.Counter(), collections.Counter(), collections.Counter()
cur_end, last = sel[0][1], sel[0]
idle = 0.0
for e in sel[1:]:
    if e[0] > cur_end:
        g = (e[0] - cur_end) / 1e3
        idle += g
        if g >= min_gap:
            by_next[e[2]] += g; by_prev[last[2]] += g
            hist[min(int(g // 10) * 10, 100)] += g
    if e[1] > cur_end:
        cur_end, last = e[1], e
steps = len(ws) - 1 - len(ws) // 2
print(f"span {span / 1e3:.3f} ms, idle {idle / 1e3:.3f} ms = {idle / span:.3f} of the span ({steps} window-scatter launches)")
print("idle by the kernel that ends the gap (us, share of idle):")
for k, v in by_next.most_common(12): print(f"  {k:28s} {v:9.1f} {v / idle:6.3f}")
print("idle by the last kernel before the gap:")
for k, v in by_prev.most_common(12): print(f"  {k:28s} {v:9.1f} {v / idle:6.3f}")
print("idle by gap length (us bucket -> us):", dict(sorted(hist.items())))
