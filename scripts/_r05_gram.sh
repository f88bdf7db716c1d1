#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/gram
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_baseline_configs_gpu.py tests/test_small_gpu.py -x -q -k "fit or data_term or cfg4 or n512 or n1024 or task or definite" 2>&1 | tail -8 > gpurun_out/gram/tests3.log
cat gpurun_out/gram/tests3.log
