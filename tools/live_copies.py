"""Which copies of the headline region run as blit KERNELS (on the compute units) and which on the SDMA engines.
    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/trace -- python bench.py --lean --steps 12 --warmup 3
    python tools/live_copies.py [trace dir]
Prints, for the timed steps (between the middle and the last k_win_scatter launch, like tools/live_busy.py): the
`__amd_rocclr_*` kernels by grid size (bytes moved ~ grid x 16 or 4 per lane), queue and share of the span, and the rows of the
memory-copy trace by direction / agent pair / size bucket."""
import collections
import csv
import glob
import re
import sys

d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/trace"
kt = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(kt)))
print("kernel trace columns:", list(rows[0].keys()))
name = lambda r: (re.search(r"(k_\w+|__amd_rocclr_\w+)", r["Kernel_Name"]) or re.search(r"(\w+)", r["Kernel_Name"])).group(1)
ev = sorted(rows, key=lambda r: int(r["Start_Timestamp"]))
ws = [i for i, r in enumerate(ev) if name(r) == "k_win_scatter"]
a, b = ws[len(ws) // 2], ws[-1]
t0, t1 = int(ev[a]["Start_Timestamp"]), int(ev[b]["Start_Timestamp"])
span = t1 - t0
steps = len(ws) - 1 - len(ws) // 2
print(f"span {span / 1e6:.3f} ms, {steps} k_win_scatter launches")
agg = collections.OrderedDict()
for r in ev:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < t0 or s >= t1 or not name(r).startswith("__amd_rocclr"):
        continue
    gx = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
    key = (name(r), gx, r.get("Queue_Id", "?"), r.get("Stream_Id", "?"))
    v = agg.setdefault(key, [0, 0])
    v[0] += 1; v[1] += e - s
print("blit kernels inside the span: (kernel, grid_x, queue, stream) -> launches, total us, share of span")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"  {k}: {v[0]} launches, {v[1] / 1e3:.1f} us, {v[1] / span:.4f}")
tot = sum(v[1] for v in agg.values())
print(f"  all blit kernels: {tot / span:.4f} of the span, {sum(v[0] for v in agg.values()) / max(steps, 1):.1f} launches per k_win_scatter launch")
mc = sorted(glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True))
if not mc:
    print("no memory-copy trace")
    sys.exit(0)
mrows = list(csv.DictReader(open(mc[-1])))
if not mrows:
    print("memory-copy trace is empty")
    sys.exit(0)
print("memory-copy trace columns:", list(mrows[0].keys()))
magg = collections.OrderedDict()
for r in mrows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < t0 or s >= t1:
        continue
    nb = None
    for c in ("Bytes", "Size", "bytes"):
        if c in r and r[c] not in ("", None):
            nb = int(r[c]); break
    bucket = "?" if nb is None else ("<4K" if nb < 4096 else "<64K" if nb < 65536 else "<1M" if nb < (1 << 20) else "<16M" if nb < (16 << 20) else ">=16M")
    key = (r.get("Direction", "?"), r.get("Source_Agent_Id", "?"), r.get("Destination_Agent_Id", "?"), bucket)
    v = magg.setdefault(key, [0, 0, 0])
    v[0] += 1; v[1] += e - s; v[2] += nb or 0
print("memory copies inside the span: (direction, src agent, dst agent, size) -> count, total us, MB")
for k, v in sorted(magg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k}: {v[0]} copies, {v[1] / 1e3:.1f} us, {v[2] / 1e6:.2f} MB")
