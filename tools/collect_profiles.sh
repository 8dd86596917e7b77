#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root: rocprofv3 passes of the default bench command.
#   1. --kernel-trace --stats            -> per-kernel durations
#   2. --pmc FETCH_SIZE  (own pass)      -> HBM/L2 read traffic per kernel
#   3. --pmc WRITE_SIZE  (own pass)      -> write traffic per kernel
#   4. --pmc SQ_* (two passes of <= 8 SQ counters + GRBM) -> VALU instructions / issue cycles, wave cycles (occupancy),
#      LDS stalls and bank conflicts; the rows also carry each kernel's VGPR / AGPR / SGPR / LDS / scratch allocation
# (counters always in their own run with --kernel-trace only, never with a sys/runtime/hip/hsa trace)
# Output under gpurun_out/prof/<tag>/ ; tools/summarize_profiles.py turns it into profiles/<round>/ files.
set -u
TAG=${1:-run}
R=$PWD
OUT=$R/gpurun_out/prof/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
# one group of BA windows per launch while the counters run: a batched kernel then covers the whole batch and has the chip
# to itself (the default, two groups on two streams, is what the bench line after the passes is taken with)
export SSX_BA_GROUPS=1
CMD="python $R/bench.py --steps 5 --warmup 2 --lean --profile-kernels"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $CMD > "$OUT/stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- $CMD > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- $CMD > "$OUT/write.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d "$OUT/sq1" -- $CMD > "$OUT/sq1.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_BUSY_CU_CYCLES GRBM_COUNT \
    --kernel-trace --output-format csv -d "$OUT/sq2" -- $CMD > "$OUT/sq2.log" 2>&1
cd "$R"
unset SSX_BA_GROUPS
python bench.py --steps 20 --warmup 3 > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -c 600 "$OUT/bench.json"
# the raw per-dispatch tables are hundreds of MB (gpurun merges <= 64 MiB back): summarise HERE, keep the summaries
mkdir -p "$OUT/summary"; cp "profiles/${3:-r03}/kernel_resources.csv" "$OUT/summary/" 2>/dev/null
python tools/summarize_profiles.py "$TAG" "$OUT/summary" "${2:-bench}" > "$OUT/summary.log" 2>&1
cp profiles/pmc_counters.json "$OUT/summary/" 2>/dev/null
for d in stats fetch write sq1 sq2; do
  find "$OUT/$d" -name "*_agent_info.csv" -exec cp {} "$OUT/summary/agent_info_$d.csv" \; 2>/dev/null
  rm -rf "$OUT/$d"
done
ls -la "$OUT/summary"
