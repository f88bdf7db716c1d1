set -u
mkdir -p gpurun_out/r06s
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06s/smoke.log 2>&1; echo "smoke rc $?" >> gpurun_out/r06s/smoke.log
tail -2 gpurun_out/r06s/smoke.log
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r06s/gpu_suite.log 2>&1; echo "pytest rc $?" >> gpurun_out/r06s/gpu_suite.log
tail -12 gpurun_out/r06s/gpu_suite.log
