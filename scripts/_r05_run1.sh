set -u
mkdir -p gpurun_out/r05a
python scripts/gpu_nehvi_setup_probe.py > gpurun_out/r05a/nehvi_setup.log 2>&1
python scripts/gpu_fit_eval_large.py > gpurun_out/r05a/fit_eval.log 2>&1
ROOT=$(pwd); cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r05a/fit512 -- python $ROOT/scripts/gpu_fit_eval_large.py 512 > $ROOT/gpurun_out/r05a/fit512.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r05a/fiticm -- python $ROOT/scripts/gpu_fit_eval_large.py icm > $ROOT/gpurun_out/r05a/fiticm.log 2>&1
cd $ROOT; nproc > gpurun_out/r05a/nproc.txt; lscpu | head -20 >> gpurun_out/r05a/nproc.txt
cat gpurun_out/r05a/nehvi_setup.log gpurun_out/r05a/fit_eval.log
