#!/bin/bash
# rocprofv3 passes behind profiles/rNN_*: kernel trace + stats, then one PMC pass per counter group
# (never combined with a trace domain), all on `bench.py --steps 3 --warmup 1 --cpu-budget 0`.
# Run on the GPU box from the repo root:  bash scripts/profile_bench.sh r01
# Raw output goes to gpurun_out/prof_<tag>/, the summaries are aggregated by scripts/profile_summarize.py.
set -u
TAG=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 3 --warmup 1 --cpu-budget 0 --greedy 5"
# the kernel-trace pass uses the default step count so that its average covers the same launches as bench.py's
# HIP-event average (3 warm-up + 20 timed); the PMC passes below use the short run
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python $ROOT/bench.py --steps 20 --warmup 3 --cpu-budget 0 --greedy 5 > "$OUT/stats.log" 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_INSTS_SALU SQ_WAIT_INST_LDS"; do
  name=$(echo "$grp" | tr ' ' '+')
  rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc_$name" -- $CMD > "$OUT/pmc_$name.log" 2>&1
done
# the fp64 pipe probe under the co-execution counter (does VALU work ever overlap an fp64 MFMA?)
if [ -x "$ROOT/scripts/_bin/ovl" ]; then
  rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d "$OUT/pmc_probe_coexec" -- "$ROOT/scripts/_bin/ovl" > "$OUT/pmc_probe_coexec.log" 2>&1
fi
cd "$ROOT" && python scripts/profile_summarize.py "$OUT" "$TAG"
