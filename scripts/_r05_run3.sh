set -u
mkdir -p gpurun_out/r05c
timeout 900 python -m pytest tests/test_nehvi_gpu.py tests/test_baseline_configs_gpu.py -x -q -k "nehvi or cfg5 or device or pruning or scores or pareto" > gpurun_out/r05c/test_nehvi.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05c/test_nehvi.log
timeout 300 python scripts/gpu_nehvi_setup_probe.py > gpurun_out/r05c/nehvi_setup.log 2>&1
timeout 600 python bench.py --config cfg5 --cpu-budget 6 > gpurun_out/r05c/bench_cfg5.log 2>&1
tail -12 gpurun_out/r05c/test_nehvi.log; cat gpurun_out/r05c/nehvi_setup.log; tail -3 gpurun_out/r05c/bench_cfg5.log | cut -c1-6000
