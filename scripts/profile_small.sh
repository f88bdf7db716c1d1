#!/bin/bash
# Pipe-level issue fraction of the register-resident small-model kernel (bbh_small_posterior_kernel, n = 64 / 32 at 1e6 candidates):
# SQ_INSTS_VALU / SQ_INSTS_MFMA and GRBM_GUI_ACTIVE in their own PMC passes (with --kernel-trace for the durations), summarised by
# scripts/profile_summarize.py with the static class mix of profiles/r04_isa_class_mix.json.   bash scripts/profile_small.sh r04
set -u
TAG=${1:-r04}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_${TAG}_small
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
  name=$(echo "$grp" | tr ' ' '+')
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/pmcg_$name" -- python $ROOT/scripts/prof_small.py > "$OUT/pmcg_$name.log" 2>&1
done
cd "$ROOT" && python scripts/profile_summarize.py "$OUT" "${TAG}_small"
