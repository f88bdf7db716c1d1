set -u
mkdir -p gpurun_out/r05e
BBH_FIT_FLOW=1 BBH_FLOW_TRACE=1 timeout 300 python scripts/gpu_flow_trace.py 512 > gpurun_out/r05e/flow_trace_512_tail.log 2>&1
cat gpurun_out/r05e/flow_trace_512_tail.log
BBH_TILE_TRACE=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -s -k "one_launch" > gpurun_out/r05e/test_flow.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05e/test_flow.log
grep -E 'evaluation|passed|failed|rc|Error|assert' gpurun_out/r05e/test_flow.log | head -30
