#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/wt
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_baseline_configs_gpu.py -x -q -k "fit or data_term or cfg4 or n512 or n1024 or definite" 2>&1 | tail -5 > gpurun_out/wt/tests.log
echo "--- write-through hand-offs (default)" > gpurun_out/wt/eval.log
timeout 300 python scripts/gpu_fit_eval_large.py >> gpurun_out/wt/eval.log 2>&1
echo "--- BBH_TILE_WT=0" >> gpurun_out/wt/eval.log
BBH_TILE_WT=0 timeout 300 python scripts/gpu_fit_eval_large.py 128 512 1024 >> gpurun_out/wt/eval.log 2>&1
timeout 100 python scripts/gpu_tile_stamps.py 512 > gpurun_out/wt/stamps.log 2>&1
cat gpurun_out/wt/tests.log; cat gpurun_out/wt/eval.log; head -12 gpurun_out/wt/stamps.log
