"""Summarise a rocprofv3 run of tools/microbench/copy_engine.hip: per case (cut at the k_marker launches, grid = case index x 64),
rows of the memory-copy trace (copies the SDMA engines made) and __amd_rocclr_copyBuffer launches (copies made by a blit kernel on
the compute units).   python tools/microbench/copy_engine_summary.py <trace dir> <stderr of the run>"""
import csv, glob, re, sys
d, cases_file = sys.argv[1], sys.argv[2]
names = {int(m.group(1)): m.group(2).strip() for m in (re.match(r"CASE (\d+) (.*)", l) for l in open(cases_file)) if m}
kt = list(csv.DictReader(open(sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1])))
mcf = sorted(glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True))
mc = list(csv.DictReader(open(mcf[-1]))) if mcf else []
marks = sorted((int(r["Start_Timestamp"]), int(r["Grid_Size_X"]) // 64) for r in kt if "k_marker" in r["Kernel_Name"])
marks.append((1 << 62, -1))
for (t0, idx), (t1, _) in zip(marks[:-1], marks[1:]):
    blit = [r for r in kt if "__amd_rocclr" in r["Kernel_Name"] and t0 <= int(r["Start_Timestamp"]) < t1]
    sd = [r for r in mc if t0 <= int(r["Start_Timestamp"]) < t1]
    bl_us = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in blit) / 1e3
    sd_us = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sd) / 1e3
    print(f"{names.get(idx, '?'):58s} sdma copies {len(sd):2d} ({sd_us / max(len(sd), 1):8.1f} us each)   blit kernels {len(blit):2d} ({bl_us / max(len(blit), 1):8.1f} us each)")
