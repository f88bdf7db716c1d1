set -u
mkdir -p gpurun_out/r06bench
for cfg in cfg3 cfg2 cfg4 cfg5; do
  timeout 900 python bench.py --config $cfg > gpurun_out/r06bench/$cfg.log 2>&1
  grep '^{"metric"' gpurun_out/r06bench/$cfg.log | tail -1 > gpurun_out/r06bench/$cfg.json
done
timeout 600 python bench.py --rows 125000 --cpu-budget 0 > gpurun_out/r06bench/cfg3_125k.log 2>&1
grep '^{"metric"' gpurun_out/r06bench/cfg3_125k.log | tail -1 > gpurun_out/r06bench/cfg3_125k.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06bench/*.json')):
    try: d=json.loads(open(f).read())
    except Exception as e: print(f, 'unparsed', e); continue
    ex=d['extra']
    keys=['fit_ms','fit_nfev','fit_ms_sequential','fit_ms_per_evaluation','factorize_ms','greedy_q5_ms','recommend_e2e_ms','nehvi_setup_ms','nehvi_first_setup_ms','nehvi_prune_ms','nehvi_prune_parts_ms','ms_per_selection_step','cpu_fit']
    print(f.split('/')[-1], 'value %.4g' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'frac', d['roofline'].get('frac'), 'traffic', d['roofline'].get('traffic'), d['roofline'].get('traffic_source'), {k: ex.get(k) for k in keys if k in ex}, 'cpu', (d.get('cpu_baseline') or {}).get('value'))
PY
