"""Turn gpurun_out/prof/<tag>/ (tools/collect_profiles.sh) into the committed summaries under profiles/<round>/.

    python tools/summarize_profiles.py <tag> <round-dir> <name>

writes <name>_kernel_stats.csv (rocprofv3 --stats table, kernel names shortened), <name>_pmc_hbm.md, <name>.json
(the bench line) and refreshes profiles/pmc_traffic.json, which bench.py reads for roofline.traffic.
"""
import collections
import csv
import glob
import json
import os
import re
import sys

tag, rdir, name = sys.argv[1], sys.argv[2], sys.argv[3]
src = os.path.join("gpurun_out", "prof", tag)
os.makedirs(rdir, exist_ok=True)


def short(n):
    m = re.search(r"(k_\w+(?:<[^>]*>)?|__amd_rocclr_\w+)", n)
    return m.group(1) if m else n[:48]


# 1. kernel stats
f = sorted(glob.glob(os.path.join(src, "stats", "*", "*kernel_stats.csv")))[-1]
rows = list(csv.DictReader(open(f)))
with open(os.path.join(rdir, name + "_kernel_stats.csv"), "w") as o:
    o.write("kernel,calls,total_ns,avg_ns,min_ns,max_ns,percent\n")
    for r in rows:
        o.write(f"{short(r['Name'])},{r['Calls']},{r['TotalDurationNs']},{float(r['AverageNs']):.1f},{r['MinNs']},{r['MaxNs']},{r['Percentage']}\n")
avg_ns = {short(r["Name"]): float(r["AverageNs"]) for r in rows}


# 2. PMC
def pmc(sub, counter):
    f = sorted(glob.glob(os.path.join(src, sub, "*", "*counter_collection.csv")))[-1]
    tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        tot[k] += float(r["Counter_Value"])
        cnt[k] += 1
    return {k: (tot[k] / cnt[k], cnt[k]) for k in tot}


fe, wr = pmc("fetch", "FETCH_SIZE"), pmc("write", "WRITE_SIZE")
traffic = {}
bj = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
pairs = bj["config"]["pairs_per_step_per_gpu"]
with open(os.path.join(rdir, name + "_pmc_hbm.md"), "w") as o:
    o.write(f"# PMC memory-side counters of `python bench.py --steps 5 --warmup 2 --no-cpu-baseline` ({pairs} pairs/step), MI355X\n\n"
            "Two separate `rocprofv3 --pmc <counter> --kernel-trace` passes (FETCH_SIZE, WRITE_SIZE), averaged per launch.\n"
            "Raw counter units are KB.  /opt/skills/guides/MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE counts half of\n"
            "the bytes of a WIDE (16 B/lane) streaming read and must be doubled for those; other widths and WRITE_SIZE are to be\n"
            "calibrated on a known byte count.  No kernel of this path issues 16 B/lane streams (1-8 B per lane), so no doubling\n"
            "is applied: `traffic` = FETCH_SIZE + WRITE_SIZE.  Calibration on `k_copy_level0` (reads 32 x 1241 x 376 B = 14.9 MB,\n"
            "writes 32 x 1280 x 376 B = 15.4 MB at 16 pairs/step): WRITE_SIZE is exact, FETCH_SIZE reads 0.64 of the byte count (the input was\n"
            "just produced and is partly L2-resident); `k_gauss7` reads 1.26 x the pyramid because of its tile halo and FETCH_SIZE\n"
            "shows exactly that.  The counters sit between L2 and the fabric: reads served by L2 never show up.\n\n"
            "| kernel | launches | avg us (stats pass) | FETCH_SIZE KB | WRITE_SIZE KB | traffic MB/launch |\n|---|---|---|---|---|---|\n")
    for k in sorted(fe, key=lambda k: -(fe[k][0] + wr.get(k, (0, 0))[0])):
        w = wr.get(k, (0.0, 0))[0]
        t = (fe[k][0] + w) * 1024.0
        traffic[k] = t
        o.write(f"| {k} | {fe[k][1]} | {avg_ns.get(k, 0) / 1e3:.1f} | {fe[k][0]:.1f} | {w:.1f} | {t / 1e6:.2f} |\n")
json.dump(bj, open(os.path.join(rdir, name + ".json"), "w"), indent=1)
json.dump({"source": f"{rdir}/{name}_pmc_hbm.md", "pairs_per_step": bj["config"]["pairs_per_step_per_gpu"],
           "bytes_per_launch": {k: round(v) for k, v in traffic.items()}},
          open(os.path.join("profiles", "pmc_traffic.json"), "w"), indent=1)
print(open(os.path.join(rdir, name + "_pmc_hbm.md")).read())
