set -u
mkdir -p gpurun_out/r05g
timeout 900 python -m pytest tests/test_nehvi_gpu.py tests/test_baseline_configs_gpu.py tests/test_reference_replay_gpu.py tests/test_plugin_gpu.py -q > gpurun_out/r05g/tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05g/tests.log
tail -8 gpurun_out/r05g/tests.log
