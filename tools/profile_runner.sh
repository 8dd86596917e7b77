#!/bin/bash
# Run ON THE GPU BOX from the repo root: rocprofv3 kernel statistics of the headless runner over the rendered
# forward-drive sequence (reference settings).  Output: gpurun_out/prof/<tag>/runner_kernel_stats.csv (+ the runner's
# own summary); copy what should be kept into profiles/.
set -u
TAG=${1:-runner}
FRAMES=${2:-120}
R=$PWD
OUT=$R/gpurun_out/prof/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
SEQ=/tmp/ssx_corridor_$$
python - <<PY
import sys
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tests")
import host_util as hu
hu.build_test_binaries()
seq = hu.write_corridor_sequence("$SEQ", n_frames=$FRAMES)
hu.write_config("$SEQ/cfg.yaml", {})
PY
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- "$R/ssvio_amd/host/ssx_run_kitti" \
    --config_yaml_path=$SEQ/cfg.yaml --kitti_dataset_path=$SEQ/seq --trajectory=$SEQ/t.txt --decode_threads=16 > "$OUT/runner.log" 2>&1
cd "$R"
python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("::")[-1]
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a = agg.setdefault(n, [0, 0, 1 << 62, 0]); a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
tot = sum(a[1] for a in agg.values())
with open("$OUT/runner_kernel_stats.csv", "w") as o:
    o.write("kernel,calls,total_ns,avg_ns,min_ns,max_ns,percent\n")
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        o.write(f"{n},{a[0]},{a[1]},{a[1] / a[0]:.1f},{a[2]},{a[3]},{100.0 * a[1] / tot:.2f}\n")
print(open("$OUT/runner_kernel_stats.csv").read())
PY
grep -v "^W2\|^E2\|rocprofv3" "$OUT/runner.log" | tail -14
rm -rf "$SEQ"
