#!/bin/bash
# rocprofv3 passes behind profiles/<tag>_<cfg>_*: kernel trace + stats of `bench.py --config <cfg>`, then the two HBM-traffic
# PMC passes (FETCH_SIZE, WRITE_SIZE - one counter per run, never combined with a trace domain) plus one pass with the MFMA /
# VALU instruction counters.  Run on the GPU box from the repo root:   bash scripts/profile_config.sh r03 cfg4
# Raw output goes to gpurun_out/prof_<tag>_<cfg>/, summaries to gpurun_out/prof_<tag>_<cfg>/summary/ (copy to profiles/).
set -u
TAG=${1:-r03}; CFG=${2:-cfg3}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_${TAG}_$CFG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python $ROOT/bench.py --config $CFG --steps 20 --warmup 3 --cpu-budget 0 > "$OUT/stats.log" 2>&1
grep "^{\"metric\"" "$OUT/stats.log" | tail -1 > "$OUT/bench_line.json"
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
  name=$(echo "$grp" | tr ' ' '+')
  rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc_$name" -- python $ROOT/bench.py --config $CFG --steps 3 --warmup 1 --cpu-budget 0 --greedy 0 > "$OUT/pmc_$name.log" 2>&1
done
cd "$ROOT" && python scripts/profile_summarize.py "$OUT" "${TAG}_$CFG"
