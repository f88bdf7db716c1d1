"""One device fit on the bench model (n = 512, d = 20) - for rocprofv3 --kernel-trace --stats."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from bench import synth_problem
from baybe_amd import engine, gp_spec
d, n = 20, 512
X, Xt, y = synth_problem(4096, d, n, 0)
g = engine.HipGP(0)
g.set_model(gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d)), Xt, y)
g.data_term(gp_spec.initial_params(g.spec))
t0 = time.perf_counter(); fi = g.fit(); t1 = time.perf_counter()
print(f"fit {1e3 * (t1 - t0):.1f} ms nfev {fi.nfev} nit {fi.nit}")
t0 = time.perf_counter()
for _ in range(20): g.data_term(fi.params)
print(f"data_term {1e3 * (time.perf_counter() - t0) / 20:.3f} ms per evaluation")
