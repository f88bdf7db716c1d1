set -u
ROOT=$(pwd)
mkdir -p gpurun_out/lat_trace
cd /tmp; export TMPDIR=/tmp
for i in 1 2; do
rm -rf $ROOT/gpurun_out/lat_trace/raw
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/lat_trace/raw -- python $ROOT/scripts/gpu_small_space_latency.py > $ROOT/gpurun_out/lat_trace/run.log 2>&1
cd $ROOT
grep "^{" gpurun_out/lat_trace/run.log | tail -1 | cut -c100-220
python - <<'PY'
import glob, csv
files=glob.glob('gpurun_out/lat_trace/raw/**/*kernel_trace.csv', recursive=True)
rows=list(csv.DictReader(open(files[0])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
print(len(rows),'kernels')
longk=[(r['Kernel_Name'][:70],(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6) for r in rows if int(r['End_Timestamp'])-int(r['Start_Timestamp'])>2_000_000]
print('kernels longer than 2 ms:', longk[:10], len(longk))
# scatter kernels: gap since the previous kernel's end, and what the previous kernel was
for k,r in enumerate(rows):
    if 'bbh_scatter_kernel' in r['Kernel_Name'] and k>0:
        prev=rows[k-1]
        gap=(int(r['Start_Timestamp'])-int(prev['End_Timestamp']))/1e6
        nxt=rows[k+1] if k+1<len(rows) else None
        gap2=(int(nxt['Start_Timestamp'])-int(r['End_Timestamp']))/1e6 if nxt else -1
        print(f"scatter: {gap:9.3f} ms after {prev['Kernel_Name'][:40]!r} (queue {prev.get('Queue_Id')}); own queue {r.get('Queue_Id')}; next kernel {gap2:8.3f} ms later: {nxt['Kernel_Name'][:40] if nxt else None!r}")
PY
cd /tmp
done
