#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/mt
timeout 300 python scripts/gpu_fit_eval_large.py > gpurun_out/mt/eval2.log 2>&1
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_baseline_configs_gpu.py -x -q -k "fit or data_term or cfg4 or n512 or n1024 or definite" 2>&1 | tail -5 > gpurun_out/mt/tests2.log
cat gpurun_out/mt/eval2.log; cat gpurun_out/mt/tests2.log
