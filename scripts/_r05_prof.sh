set -u
ROOT=$(pwd)
for cfg in cfg2 cfg4 cfg5; do bash scripts/profile_config.sh r05 $cfg > gpurun_out/prof_r05_$cfg.log 2>&1; done
bash scripts/profile_config.sh r05 cfg3 "--rows 125000" _125k > gpurun_out/prof_r05_cfg3_125k.log 2>&1
mkdir -p gpurun_out/r05fit
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r05fit/fit512 -- python $ROOT/scripts/gpu_fit_eval_large.py 512 > $ROOT/gpurun_out/r05fit/fit512.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r05fit/fiticm -- python $ROOT/scripts/gpu_fit_eval_large.py icm > $ROOT/gpurun_out/r05fit/fiticm.log 2>&1
cd $ROOT
python scripts/gpu_fit_eval_large.py > gpurun_out/r05fit/fit_eval.log 2>&1
BBH_FIT_FLOW=1 python scripts/gpu_flow_trace2.py 512 > gpurun_out/r05fit/flow_tail_trace_512.log 2>&1
BBH_FIT_FLOW=1 python scripts/gpu_flow_trace2.py 1024 > gpurun_out/r05fit/flow_tail_trace_1024.log 2>&1
BBH_FIT_FLOW=2 python scripts/gpu_flow_trace.py 512 > gpurun_out/r05fit/flow_full_trace_512.log 2>&1
python scripts/gpu_nehvi_setup_probe.py > gpurun_out/r05fit/nehvi_setup.log 2>&1
BBH_NEHVI_HOST=1 python scripts/gpu_nehvi_setup_probe.py > gpurun_out/r05fit/nehvi_setup_host.log 2>&1
python scripts/gpu_greedy_breakdown.py 125000 > gpurun_out/r05fit/greedy_125k.log 2>&1
python scripts/gpu_small_space_latency.py > gpurun_out/r05fit/small_space_latency.log 2>&1
cat gpurun_out/r05fit/fit_eval.log; tail -3 gpurun_out/r05fit/small_space_latency.log
ls gpurun_out/prof_r05_*/summary | head -40
