#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root: three rocprofv3 passes of the default bench command.
#   1. --kernel-trace --stats            -> per-kernel durations
#   2. --pmc FETCH_SIZE  (own pass)      -> HBM/L2 read traffic per kernel
#   3. --pmc WRITE_SIZE  (own pass)      -> write traffic per kernel
# Output under gpurun_out/prof/<tag>/ ; tools/summarize_profiles.py turns it into profiles/<round>/ files.
set -u
TAG=${1:-run}
R=$PWD
OUT=$R/gpurun_out/prof/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
CMD="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $CMD > "$OUT/stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- $CMD > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- $CMD > "$OUT/write.log" 2>&1
cd "$R"
python bench.py --steps 20 --warmup 3 > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -c 600 "$OUT/bench.json"
