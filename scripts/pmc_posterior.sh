#!/bin/bash
# PMC passes (one counter group per run, never combined with a trace domain) over scripts/prof_posterior.py.
# usage: bash scripts/pmc_posterior.sh <tag> [ENV=VAL ...]   -> gpurun_out/pmc_<tag>.csv (kernel, counter, mean per dispatch)
TAG=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmcraw_$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
for grp in "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_WAVE32_LDS SQ_LDS_BANK_CONFLICT" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo "$grp" | tr ' ' '+')
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d "$OUT/$name" -- python $ROOT/scripts/prof_posterior.py > "$OUT/$name.log" 2>&1
done
cd "$ROOT"
python - "$OUT" "$TAG" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out, tag = sys.argv[1], sys.argv[2]
acc = defaultdict(list)
for fn in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(fn)):
        if "posterior_kernel" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"].split("(")[0][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
with open(f"gpurun_out/pmc_{tag}.csv", "w") as f:
    f.write("kernel,counter,dispatches,mean_per_dispatch\n")
    for (k, c), v in sorted(acc.items()):
        f.write(f"\"{k}\",{c},{len(v)},{sum(v)/len(v):.6g}\n")
print(open(f"gpurun_out/pmc_{tag}.csv").read())
PY
