set -u
mkdir -p gpurun_out/r05f
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_baseline_configs_gpu.py tests/test_small_gpu.py -x -q > gpurun_out/r05f/tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05f/tests.log
tail -8 gpurun_out/r05f/tests.log
timeout 300 python scripts/gpu_fit_eval_large.py > gpurun_out/r05f/fit_eval.log 2>&1; cat gpurun_out/r05f/fit_eval.log
