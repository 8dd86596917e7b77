"""Print the kernel timeline of the LAST bench step from a rocprofv3 kernel_trace.csv (start offset, duration, gap)."""
import csv, glob, sys

path = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob("gpurun_out/trace/*/*kernel_trace.csv"))[-1]
rows = list(csv.DictReader(open(path)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1], r.get("Stream_Id", r.get("Queue_Id", "")))
             for r in rows), key=lambda e: e[0])
# last occurrence of the first kernel of a step
first = "k_copy_level0"
idx = [i for i, e in enumerate(ev) if e[2] == first]
if len(idx) < 2:
    sys.exit("no step boundary found")
a, b = idx[-2], idx[-1]
t0 = ev[a][0]
prev_end = t0
print(f"step: {(ev[b][0] - t0) / 1e3:.1f} us, {b - a} launches")
for s, e, k, q in ev[a:b]:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:7.1f}  q={q}  {k}")
    prev_end = max(prev_end, e)
