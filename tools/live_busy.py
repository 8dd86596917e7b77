"""GPU busy fraction of the headline region from a rocprofv3 kernel trace of `bench.py --lean`: the union of all kernel intervals
between the first and the last k_win_scatter launch of the timed steps (that kernel runs only in the live-backend region), the same
split by front-end / backend kernels, and the sum of kernel durations (> the union where kernels overlap).
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python bench.py --lean --steps 12 --warmup 3
    python tools/live_busy.py [kernel_trace.csv]"""
import csv, glob, re, sys

path = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob("gpurun_out/trace/*/*kernel_trace.csv"))[-1]
rows = list(csv.DictReader(open(path)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), (re.search(r"(k_\w+|__amd_rocclr_\w+)", r["Kernel_Name"]) or re.search(r"(\w+)", r["Kernel_Name"])).group(1)) for r in rows), key=lambda e: e[0])
ws = [i for i, e in enumerate(ev) if e[2] == "k_win_scatter"]
n_skip = int(sys.argv[2]) if len(sys.argv) > 2 else len(ws) // 2      # skip the pre-roll / warm-up half
a, b = ws[n_skip], ws[-1]
t0, t1 = ev[a][0], ev[b][0]
sel = [e for e in ev if e[0] >= t0 and e[0] < t1]
FE = ("k_copy_level0", "k_resize", "k_fast_cells", "k_octree", "k_gauss7", "k_orient_brief", "k_row_bucket", "k_match", "k_triangulate_matches",
      "k_counts", "k_pair_counts", "k_pack_frame")
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = None, None
    for s, e in iv:
        if cs is None: cs, ce = s, e
        elif s <= ce: ce = max(ce, e)
        else: tot += ce - cs; cs, ce = s, e
    if cs is not None: tot += ce - cs
    return tot
span = t1 - t0
allu = union([(s, min(e, t1)) for s, e, _ in sel])
feu = union([(s, min(e, t1)) for s, e, k in sel if k in FE])
bau = union([(s, min(e, t1)) for s, e, k in sel if k not in FE])
steps = len(ws) - 1 - n_skip
print(f"span {span / 1e6:.3f} ms over {steps} window-sync launches; busy (union) {allu / span:.3f}; front-end kernels busy {feu / span:.3f}; backend + copies busy {bau / span:.3f}")
by = {}
for s, e, k in sel: by[k] = by.get(k, 0) + (min(e, t1) - s)
tot = sum(by.values())
print(f"sum of kernel durations / span = {tot / span:.3f}")
for k, v in sorted(by.items(), key=lambda x: -x[1])[:16]: print(f"  {k:32s} {v / span:.3f}")
