set -u
mkdir -p gpurun_out/r06final
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r06final/gpu_suite.log 2>&1; echo "pytest rc $?" >> gpurun_out/r06final/gpu_suite.log
grep -E "passed|failed|rc" gpurun_out/r06final/gpu_suite.log | tail -3
for cfg in cfg3 cfg2 cfg4 cfg5; do
  timeout 900 python bench.py --config $cfg > gpurun_out/r06final/$cfg.log 2>&1
  grep '^{"metric"' gpurun_out/r06final/$cfg.log | tail -1 > gpurun_out/r06final/$cfg.json
done
timeout 600 python bench.py --rows 125000 --cpu-budget 0 > gpurun_out/r06final/cfg3_125k.log 2>&1
grep '^{"metric"' gpurun_out/r06final/cfg3_125k.log | tail -1 > gpurun_out/r06final/cfg3_125k.json
timeout 300 python scripts/gpu_fit_eval_large.py > gpurun_out/r06final/fit_eval.log 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06final/*.json')):
    try: d=json.loads(open(f).read())
    except Exception as e: print(f, 'unparsed', e); continue
    ex=d['extra']
    keys=['fit_ms','fit_nfev','fit_ms_sequential','fit_ms_per_evaluation','greedy_q5_ms','recommend_e2e_ms','nehvi_setup_ms','nehvi_prune_ms','nehvi_prune_ms_steady','ms_per_selection_step']
    print(f.split('/')[-1], 'value %.4g' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'frac', d['roofline'].get('frac'), 'traffic', d['roofline'].get('traffic'), {k: ex.get(k) for k in keys if k in ex}, 'cpu', (d.get('cpu_baseline') or {}).get('value'))
PY
grep -v amdgpu gpurun_out/r06final/fit_eval.log
