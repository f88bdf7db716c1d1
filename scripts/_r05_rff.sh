#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/rff
timeout 900 python -m pytest tests/test_rff_gpu.py -x -q 2>&1 | tail -40 > gpurun_out/rff/tests.log
timeout 600 python scripts/gpu_rff_probe.py > gpurun_out/rff/probe.log 2>&1
tail -5 gpurun_out/rff/tests.log; cat gpurun_out/rff/probe.log | tail -8
