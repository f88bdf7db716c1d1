set -u
ROOT=$(pwd)
mkdir -p gpurun_out/r06misc
bash scripts/_r06_bench.sh > gpurun_out/r06misc/bench_summary_untraced.log 2>&1
bash scripts/profile_config.sh r06 cfg3 > gpurun_out/prof_r06_cfg3.log 2>&1
for cfg in cfg2 cfg4 cfg5; do bash scripts/profile_config.sh r06 $cfg > gpurun_out/prof_r06_$cfg.log 2>&1; done
bash scripts/profile_config.sh r06 cfg3 "--rows 125000" _125k > gpurun_out/prof_r06_cfg3_125k.log 2>&1
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r06misc/fit512 -- python $ROOT/scripts/gpu_fit_eval_large.py 512 > $ROOT/gpurun_out/r06misc/fit512.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r06misc/fit1088 -- python $ROOT/scripts/gpu_fit_eval_large.py 1088 icm1040 > $ROOT/gpurun_out/r06misc/fit1088.log 2>&1
cd $ROOT
timeout 300 python scripts/gpu_fit_eval_large.py > gpurun_out/r06misc/fit_eval.log 2>&1
timeout 600 python scripts/gpu_rff_probe.py > gpurun_out/r06misc/rff_probe.log 2>&1
timeout 200 python scripts/gpu_nehvi_setup_probe.py > gpurun_out/r06misc/nehvi_setup.log 2>&1
timeout 200 python scripts/gpu_greedy_breakdown.py 125000 > gpurun_out/r06misc/greedy_125k.log 2>&1
timeout 200 python scripts/gpu_small_space_latency.py > gpurun_out/r06misc/small_space_latency.log 2>&1
timeout 200 python scripts/gpu_cfg5_e2e_profile.py > gpurun_out/r06misc/cfg5_e2e_profile.log 2>&1
cat gpurun_out/r06misc/bench_summary_untraced.log
cat gpurun_out/r06misc/fit_eval.log gpurun_out/r06misc/rff_probe.log | grep -v amdgpu.ids
grep "^{" gpurun_out/r06misc/small_space_latency.log
ls gpurun_out/prof_r06_*/summary | head -60
