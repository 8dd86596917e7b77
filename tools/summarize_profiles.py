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
    o.write(f"# PMC memory-side counters of `python bench.py --steps 5 --warmup 2 --lean` ({pairs} pairs/step), MI355X\n\n"
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

# 3. SQ passes: VALU instructions / issue, wave cycles (achieved occupancy), LDS stalls; kernel resource table
def pmc_all(sub):
    fs = sorted(glob.glob(os.path.join(src, sub, "*", "*counter_collection.csv")))
    if not fs:
        return {}, {}
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    meta = {}
    for r in csv.DictReader(open(fs[-1])):
        k = short(r["Kernel_Name"])
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
        meta[k] = dict(wg=int(r["Workgroup_Size"]), lds=int(r["LDS_Block_Size"]), scratch=int(r["Scratch_Size"]), vgpr=int(r["VGPR_Count"]),
                       agpr=int(r["Accum_VGPR_Count"]), sgpr=int(r["SGPR_Count"]))
    return {k: {c: tot[k][c] / cnt[k][c] for c in tot[k]} for k in tot}, meta


# registers / scratch / LDS per kernel as the COMPILER reports them (tools/kernel_resources.py; the dispatch rows of rocprofv3
# show half the VGPRs of a wave64 kernel and no dynamic LDS): <round-dir>/kernel_resources.csv, else profiles/kernel_resources.csv
cres = {}
for cand in (os.path.join(rdir, "kernel_resources.csv"), os.path.join("profiles", "kernel_resources.csv")):
    if os.path.exists(cand):
        for r in csv.DictReader(open(cand)):
            dyn = int(r.get("dynamic_lds_bytes_per_block") or 0)
            cres[r["kernel"]] = dict(vgpr=int(r["vgprs"] or 0), agpr=int(r["agprs"] or 0), sgpr=int(r["sgprs"] or 0), scratch=int(r["scratch_bytes_per_lane"] or 0),
                                     lds=int(r["static_lds_bytes_per_block"] or 0) + max(dyn, 0), lds_known=dyn >= 0)
        break
sq = {}
meta = {}
for sub in ("sq1", "sq2"):
    a, m = pmc_all(sub)
    for k, v in a.items():
        sq.setdefault(k, {}).update(v)
    meta.update(m)
CLK_GHZ, SIMDS = 2.4, 1024
per_launch = {}
for k in set(list(traffic) + list(sq)):
    per_launch[k] = {"hbm_bytes": round(traffic[k]) if k in traffic else None, "avg_us_rocprof_stats": round(avg_ns.get(k, 0) / 1e3, 2) or None}
    for c, v in sq.get(k, {}).items():
        per_launch[k][c] = round(v)
if sq:
    with open(os.path.join(rdir, name + "_occupancy_valu.md"), "w") as o:
        o.write(f"# Occupancy and VALU issue per kernel, `python bench.py --steps 5 --warmup 2 --lean` ({pairs} pairs/step), MI355X\n\n"
                "Two `rocprofv3 --pmc SQ_* --kernel-trace` passes (tools/collect_profiles.sh), values averaged per launch; `avg us` from the\n"
                "separate `--kernel-trace --stats` pass.  Registers, scratch and LDS (static + the dynamic bytes the library launches with) are the\n"
                "COMPILER's (kernel_resources.csv; the dispatch rows are only used for kernels it does not list; VGPR/AGPR allocation granule 8).  `waves/SIMD limit` = min(8, floor(512 / (VGPR+AGPR)), LDS: floor(160 KB / LDS per WG) x waves per WG / 4).\n"
                "`achieved waves/SIMD` = SQ_WAVE_CYCLES x 4 (the counter ticks in quad-cycles) / (avg duration x 2.4 GHz x 1024 SIMDs): the\n"
                "time-averaged number of resident waves per SIMD while the kernel runs (GRBM_GUI_ACTIVE spans the profiler's dispatch envelope,\n"
                "~0.4 ms even for a 5 us kernel, and is not used; LDS_Block_Size of the dispatch row misses static __shared__ arrays: the LDS\n"
                "column of kernel_resources.csv, from the compiler, is authoritative).  `VALU issue` = SQ_INSTS_VALU x 2 cycles (a wave64\n"
                "VALU instruction issues over 2 cycles on a SIMD-32, MI355X_MICROARCH.md 'Wave scheduling') / (avg duration x 2.4 GHz x 1024\n"
                "SIMDs): the fraction of the chip's VALU issue slots the kernel fills.  `VALU busy` = SQ_ACTIVE_INST_VALU x 4 /\n"
                "SQ_WAVE_CYCLES x 4: the share of its resident wave-time a wave spends issuing VALU.  `LDS stall` = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES.\n\n"
                "| kernel | WG | VGPR | AGPR | SGPR | LDS B/WG | scratch | waves/SIMD limit | achieved waves/SIMD | avg us | SQ_INSTS_VALU | VALU issue | VALU busy | SQ_INSTS_LDS | LDS stall | bank-conflict cycles / LDS active | SQ_INSTS_SALU | SQ_INSTS_VMEM |\n"
                "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|\n")
        for k in sorted(sq, key=lambda k: -sq[k].get("SQ_INSTS_VALU", 0) * 1.0):
            m, v = dict(meta.get(k, {})), sq[k]
            if not m:
                continue
            if k in cres:                                  # the compiler's numbers win
                m.update({kk: vv for kk, vv in cres[k].items() if kk != "lds_known"})
            regs = max(((m["vgpr"] + m["agpr"] + 7) // 8) * 8, 8)
            wpw = max(m["wg"] // 64, 1)
            lim_v = min(8, 512 // regs)
            lim_l = 8 if m["lds"] == 0 else (160 * 1024 // m["lds"]) * wpw / 4.0
            lim = min(lim_v, lim_l, 8)
            dur_us = avg_ns.get(k, 0) / 1e3
            occ = v.get("SQ_WAVE_CYCLES", 0) * 4 / (dur_us * 1e-6 * CLK_GHZ * 1e9 * SIMDS) if dur_us else float("nan")
            issue = v.get("SQ_INSTS_VALU", 0) * 2 / (dur_us * 1e-6 * CLK_GHZ * 1e9 * SIMDS) if dur_us else float("nan")
            busy = v.get("SQ_ACTIVE_INST_VALU", 0) / v["SQ_WAVE_CYCLES"] if v.get("SQ_WAVE_CYCLES") else float("nan")
            stall = v.get("SQ_WAIT_INST_LDS", 0) / v["SQ_WAVE_CYCLES"] if v.get("SQ_WAVE_CYCLES") else float("nan")
            bank = v.get("SQ_LDS_BANK_CONFLICT", 0) / v["SQ_LDS_IDX_ACTIVE"] if v.get("SQ_LDS_IDX_ACTIVE") else float("nan")
            per_launch[k]["valu_issue_frac"] = round(issue, 4) if issue == issue else None
            per_launch[k]["achieved_waves_per_simd"] = round(occ, 3) if occ == occ else None
            o.write(f"| {k} | {m['wg']} | {m['vgpr']} | {m['agpr']} | {m['sgpr']} | {m['lds']} | {m['scratch']} | {lim:.1f} | {occ:.2f} | {dur_us:.1f} | "
                    f"{v.get('SQ_INSTS_VALU', 0):.0f} | {issue:.3f} | {busy:.3f} | {v.get('SQ_INSTS_LDS', 0):.0f} | {stall:.3f} | {bank:.3f} | "
                    f"{v.get('SQ_INSTS_SALU', 0):.0f} | {v.get('SQ_INSTS_VMEM', 0):.0f} |\n")
    print(open(os.path.join(rdir, name + "_occupancy_valu.md")).read())
json.dump({"source": f"{rdir}/{name}_pmc_hbm.md + {rdir}/{name}_occupancy_valu.md", "pairs_per_step": pairs,
           "ba_groups": 1, "per_launch": per_launch},
          open(os.path.join("profiles", "pmc_counters.json"), "w"), indent=1)
json.dump(bj, open(os.path.join(rdir, name + ".json"), "w"), indent=1)
json.dump({"source": f"{rdir}/{name}_pmc_hbm.md", "pairs_per_step": bj["config"]["pairs_per_step_per_gpu"],
           "bytes_per_launch": {k: round(v) for k, v in traffic.items()}},
          open(os.path.join("profiles", "pmc_traffic.json"), "w"), indent=1)
print(open(os.path.join(rdir, name + "_pmc_hbm.md")).read())
