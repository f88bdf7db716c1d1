set -u
mkdir -p gpurun_out/r05s
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r05s/gpu_suite.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05s/gpu_suite.log
tail -25 gpurun_out/r05s/gpu_suite.log
