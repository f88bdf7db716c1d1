set -u
mkdir -p gpurun_out/r05h
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -s -k "one_launch or tile_dataflow or n512 or small_model" > gpurun_out/r05h/test_flow.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05h/test_flow.log
grep -E 'one launch|passed|failed|rc |Error|assert' gpurun_out/r05h/test_flow.log | head -30
BBH_FIT_FLOW=1 python scripts/gpu_flow_trace2.py 512 2>&1 | grep -E "span|MT:|VEC|GT"
BBH_FIT_FLOW=1 python scripts/gpu_flow_trace2.py 1024 2>&1 | grep -E "span|MT:|VEC|GT"
python scripts/gpu_fit_eval_large.py 2>&1 | tail -5
