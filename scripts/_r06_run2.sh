#!/bin/bash
# round 6, second GPU call: whole GPU suite, set_model probe, latency script, cfg5 / cfg4 bench lines
mkdir -p gpurun_out
python scripts/gpu_set_model_probe.py > gpurun_out/r06_set_model_probe_after.log 2>&1
python scripts/gpu_small_space_latency.py > gpurun_out/r06_small_space_latency.log 2>&1
python bench.py --config cfg5 --cpu-budget 0 > gpurun_out/r06_cfg5_bench_b.json 2> gpurun_out/r06_cfg5_bench_b.err
python bench.py --config cfg4 --cpu-budget 0 > gpurun_out/r06_cfg4_bench_b.json 2> gpurun_out/r06_cfg4_bench_b.err
python -m pytest tests -q -m gpu 2>&1 | tail -40 > gpurun_out/r06_gpu_suite_b.log
tail -15 gpurun_out/r06_gpu_suite_b.log
grep -E "^\{" gpurun_out/r06_small_space_latency.log
tail -12 gpurun_out/r06_set_model_probe_after.log
