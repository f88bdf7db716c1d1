#!/bin/bash
# round 6, first GPU call: the new parity tests, the set_model probe, cfg5 / cfg4 bench lines with recommend_e2e_ms
mkdir -p gpurun_out
python -m pytest tests/test_nehvi_gpu.py tests/test_fitted_parity_gpu.py tests/test_reference_replay_gpu.py tests/test_reference_events_gpu.py -x -q -m gpu -k "not cfg3" 2>&1 | tail -25 > gpurun_out/r06_run1_tests.log
python scripts/gpu_set_model_probe.py > gpurun_out/r06_set_model_probe.log 2>&1
python bench.py --config cfg5 --cpu-budget 0 > gpurun_out/r06_cfg5_bench_a.json 2> gpurun_out/r06_cfg5_bench_a.err
tail -3 gpurun_out/r06_cfg5_bench_a.err
cat gpurun_out/r06_run1_tests.log
