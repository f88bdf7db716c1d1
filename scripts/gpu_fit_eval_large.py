"""Cost of one fit-objective evaluation at 64 < n <= 1024 (bbh_fit_value_grad): wall time per call incl. launch + synchronisation,
and a whole fit; single task MLL (d = 20) and the configs[3] model (ICM, 4 tasks, LOO, n = 1024)."""
import sys, time, math
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
from bench import synth_problem, synth_tl_problem
from baybe_amd import engine, gp_spec

def probe(tag, g, spec):
    th = gp_spec.theta_from_params(spec, gp_spec.initial_params(spec))
    for _ in range(10): g._data_term_theta(th)
    ts = []
    for rnd in range(5):
        t0 = time.perf_counter()
        for _ in range(40): g._data_term_theta(th)
        ts.append((time.perf_counter() - t0) / 40 * 1e6)
    t0 = time.perf_counter(); info = g.fit(); tf = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter(); info = g.fit(); tf2 = (time.perf_counter() - t0) * 1e3
    print(f"{tag}: {np.median(ts):7.1f} us per evaluation (min {min(ts):7.1f});  fit {tf:6.2f} / {tf2:6.2f} ms / {info.nfev} evaluations", flush=True)

which = sys.argv[1:] or ["128", "256", "512", "1024", "icm"]
for w in which:
    if w.startswith("icm") and w != "icm":  # e.g. icm1040: the configs[3] model a few measurements later (beyond the dataflow forms: np > 1024)
        T, d, n = 4, 15, int(w[3:])
        X, Xt, y = synth_tl_problem(4096, d, -(-n // T), T)
        Xt, y = Xt[:n], y[:n]
        spec = gp_spec.GPSpec.baybe_default(d + 1, np.zeros(d + 1), np.ones(d + 1), task_idx=d, n_tasks=T)
        g = engine.HipGP(0); g.set_model(spec, Xt, y); probe(f"icm n={n} d=15+task LOO", g, spec); g.close()
    elif w == "icm":
        T, d = 4, 15
        X, Xt, y = synth_tl_problem(4096, d, 1024 // T, T)
        spec = gp_spec.GPSpec.baybe_default(d + 1, np.zeros(d + 1), np.ones(d + 1), task_idx=d, n_tasks=T)
        g = engine.HipGP(0); g.set_model(spec, Xt, y); probe("icm n=1024 d=15+task LOO", g, spec); g.close()
    else:
        n = int(w); d = 20 if n >= 512 else 15
        X, Xt, y = synth_problem(4096, d, n, 0)
        spec = gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d))
        g = engine.HipGP(0); g.set_model(spec, Xt, y); probe(f"n={n:4d} d={d}", g, spec); g.close()
