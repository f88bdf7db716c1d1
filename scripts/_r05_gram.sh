#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/mt
echo "--- K^-1 tiles in the factorisation launch (default)" > gpurun_out/mt/eval.log
timeout 300 python scripts/gpu_fit_eval_large.py >> gpurun_out/mt/eval.log 2>&1
echo "--- BBH_TILE_MT=0" >> gpurun_out/mt/eval.log
BBH_TILE_MT=0 timeout 300 python scripts/gpu_fit_eval_large.py 128 256 512 >> gpurun_out/mt/eval.log 2>&1
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_baseline_configs_gpu.py -x -q -k "fit or data_term or cfg4 or n512 or n1024 or definite" 2>&1 | tail -5 > gpurun_out/mt/tests.log
cat gpurun_out/mt/eval.log; cat gpurun_out/mt/tests.log
