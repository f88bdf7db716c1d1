set -u
mkdir -p gpurun_out/r05d
BBH_TILE_TRACE=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -s -k "one_launch" > gpurun_out/r05d/test_flow.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05d/test_flow.log
tail -40 gpurun_out/r05d/test_flow.log
