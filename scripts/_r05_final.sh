set -u
ROOT=$(pwd)
mkdir -p gpurun_out/r05s gpurun_out/r05final
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r05s/gpu_suite.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05s/gpu_suite.log
tail -4 gpurun_out/r05s/gpu_suite.log
bash scripts/_r05_bench.sh > gpurun_out/r05final/bench_summary.log 2>&1
cat gpurun_out/r05final/bench_summary.log
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r05final/fit512 -- python $ROOT/scripts/gpu_fit_eval_large.py 512 > $ROOT/gpurun_out/r05final/fit512.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r05final/rff -- python $ROOT/scripts/gpu_rff_probe.py > $ROOT/gpurun_out/r05final/rff.log 2>&1
cd $ROOT
timeout 100 python scripts/gpu_tile_stamps.py 512 > gpurun_out/r05final/tile_stamps_512.log 2>&1
BBH_FIT_FLOW=1 timeout 100 python scripts/gpu_flow_trace2.py 512 > gpurun_out/r05final/post_trace_512.log 2>&1
BBH_FIT_FLOW=1 timeout 100 python scripts/gpu_flow_trace2.py 1024 > gpurun_out/r05final/tail_trace_1024.log 2>&1
timeout 200 python scripts/gpu_small_space_latency.py > gpurun_out/r05final/small_space_latency.log 2>&1
tail -3 gpurun_out/r05final/small_space_latency.log
