#!/bin/bash
# rocprofv3 passes behind profiles/<tag>_<cfg>_*: kernel trace + stats of `bench.py --config <cfg>`, then the two HBM-traffic
# PMC passes (FETCH_SIZE, WRITE_SIZE - one counter per run, never combined with a trace domain) plus one pass with the MFMA /
# VALU instruction counters.  Run on the GPU box from the repo root:   bash scripts/profile_config.sh r03 cfg4
# Raw output goes to gpurun_out/prof_<tag>_<cfg>/, summaries to gpurun_out/prof_<tag>_<cfg>/summary/ (copy to profiles/).
set -u
TAG=${1:-r04}; CFG=${2:-cfg3}; EXTRA=${3:-}; SUFFIX=${4:-}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_${TAG}_$CFG$SUFFIX
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python $ROOT/bench.py --config $CFG --steps 20 --warmup 3 --cpu-budget 0 $EXTRA > "$OUT/stats.log" 2>&1
grep "^{\"metric\"" "$OUT/stats.log" | tail -1 > "$OUT/bench_line.json"
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
  name=$(echo "$grp" | tr ' ' '+')
  rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc_$name" -- python $ROOT/bench.py --config $CFG --steps 3 --warmup 1 --cpu-budget 0 --greedy 0 --e2e 0 $EXTRA > "$OUT/pmc_$name.log" 2>&1
done
# the VALU-bound kernels of a greedy batch (joint q'-batch kernels) and of the selection tail: instruction counts and the clock, for
# the pipe-level issue fraction (scripts/profile_summarize.py + profiles/r04_isa_class_mix.json)
for grp in "SQ_INSTS_VALU SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
  name=$(echo "$grp" | tr ' ' '+')
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/pmcg_$name" -- python $ROOT/bench.py --config $CFG --steps 3 --warmup 1 --cpu-budget 0 --greedy 5 --e2e 0 $EXTRA > "$OUT/pmcg_$name.log" 2>&1
done
cd "$ROOT" && python scripts/profile_summarize.py "$OUT" "${TAG}_$CFG$SUFFIX"
