#!/bin/bash
mkdir -p gpurun_out
python scripts/gpu_cfg5_e2e_profile.py > gpurun_out/r06_cfg5_e2e_profile.log 2>&1
BBH_SETMODEL_TRACE=1 python scripts/gpu_small_space_latency.py > gpurun_out/r06_small_space_latency_trace.log 2>&1
python -m pytest tests/test_reference_events_gpu.py tests/test_reference_replay_gpu.py -q -m gpu 2>&1 | tail -5
grep -v "^bbh_set_model" gpurun_out/r06_cfg5_e2e_profile.log | head -120
