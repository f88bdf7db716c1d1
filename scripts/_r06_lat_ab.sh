run() { python scripts/gpu_small_space_latency.py 2>gpurun_out/lat.err | grep -E "^\{" | tail -1 | cut -c100-215; grep stages gpurun_out/lat.err | tail -3 | cut -c1-170; }
for i in 1 2 3; do
echo "kernel upload + sync (default)"; BBH_SETMODEL_TRACE=2 run
echo "kernel upload + poll"; BBH_SETMODEL_TRACE=2 BBH_SETMODEL_SYNC=poll run
done
