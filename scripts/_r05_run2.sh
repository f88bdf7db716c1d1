set -u
mkdir -p gpurun_out/r05b
timeout 600 python -m pytest tests/test_nehvi_gpu.py -x -q > gpurun_out/r05b/test_nehvi.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05b/test_nehvi.log
timeout 300 python scripts/gpu_nehvi_setup_probe.py > gpurun_out/r05b/nehvi_setup.log 2>&1
BBH_NEHVI_HOST=1 timeout 300 python scripts/gpu_nehvi_setup_probe.py > gpurun_out/r05b/nehvi_setup_host.log 2>&1
timeout 300 python scripts/gpu_tile_giveup_probe.py > gpurun_out/r05b/tile_giveup.log 2>&1
tail -15 gpurun_out/r05b/test_nehvi.log; cat gpurun_out/r05b/nehvi_setup.log gpurun_out/r05b/nehvi_setup_host.log gpurun_out/r05b/tile_giveup.log
