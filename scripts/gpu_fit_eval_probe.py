"""Cost of one fit-objective evaluation of a small model (bbh_fit_value_grad -> bbh_fit_small_kernel, one workgroup): wall time per call
(launch + kernel + one stream synchronisation) for n = 12 ... 64, d = 3 and 8; with BBH_FIT_SMALL=0 the launch-by-launch path."""
import sys, time, math
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
from baybe_amd import engine, gp_spec

rng = np.random.default_rng(0)
for d in (3, 8):
    for n in (12, 20, 33, 48, 64):
        Xt = rng.random((n, d)); y = np.sin(3 * Xt.sum(1)) + 0.05 * rng.standard_normal(n)
        g = engine.HipGP(0)
        spec = gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d))
        g.set_model(spec, Xt, y)
        th = gp_spec.theta_from_params(spec, gp_spec.initial_params(spec))
        for _ in range(20): g._data_term_theta(th)
        ts = []
        for rnd in range(5):
            t0 = time.perf_counter()
            for _ in range(200): g._data_term_theta(th)
            ts.append((time.perf_counter() - t0) / 200 * 1e6)
        t0 = time.perf_counter(); info = g.fit(); tf = (time.perf_counter() - t0) * 1e3
        print(f"d={d} n={n:3d}: {np.median(ts):6.1f} us per evaluation (min {min(ts):6.1f});  fit {tf:5.2f} ms / {info.nfev} evaluations", flush=True)
        g.close()
