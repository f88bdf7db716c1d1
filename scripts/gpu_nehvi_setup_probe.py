"""Where does a qLogNEHVI selection step's set-up go?  BASELINE configs[4] (3 targets, 1e5 x 15, n = 256, S = 512):
wall-clock of each part of ``HipNEHVI.prepare`` (after one untimed call), then the whole call and a greedy batch of 5."""
import sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from bench import synth_problem, synth_pareto_targets
from baybe_amd import engine, gp_spec
from baybe_amd.nehvi import HipNEHVI, compute_ref_point
import math

N, d, n, S = 100_000, 15, 256, int(sys.argv[1]) if len(sys.argv) > 1 else 512
X, Xt, y = synth_problem(N, d, n, 0)
ys = synth_pareto_targets(Xt)
spec = gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d))
params = gp_spec.GPParams(np.full(d, math.exp(math.sqrt(2.0) - 3.0) * math.sqrt(d)), math.exp(-5.0), 0.0)
engines = []
for yo in ys:
    g = engine.HipGP(0); g.set_model(spec, Xt, yo); g.factorize(params); engines.append(g)
ref = compute_ref_point(np.stack(ys, 1))
hv = HipNEHVI(engines, np.ones(3), Xt, ref, n_mc_samples=S, prune_baseline=True)
print("device set-up:", hv.device_setup)
Xd = torch.from_numpy(X).cuda()
t0 = time.perf_counter(); hv.prepare(1234, prune_seed=4321); torch.cuda.synchronize()
print("prune parts:", {k: round(v, 3) for k, v in getattr(hv, "last_prune_ms", {}).items()})
print("parts of the first call:", {k: round(v, 3) for k, v in hv.last_setup_ms.items()})
print(f"first prepare (incl. pruning) {1e3 * (time.perf_counter() - t0):.2f} ms; baseline points {len(hv.X_b_current)}, cells/sample {hv.n_cells / S:.1f}")
for rep in range(3):
    t0 = time.perf_counter(); hv.prepare(1234, prune_seed=4321); torch.cuda.synchronize()
    print(f"prepare again {1e3 * (time.perf_counter() - t0):.2f} ms", {k: round(v, 3) for k, v in hv.last_setup_ms.items()})
if hasattr(hv, "last_setup_ms"):
    print("parts:", {k: round(v, 3) for k, v in hv.last_setup_ms.items()})
s = hv.score(Xd); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    s = hv.score(Xd)
torch.cuda.synchronize()
print(f"score pass {1e3 * (time.perf_counter() - t0) / 5:.2f} ms")
for rep in range(2):
    torch.manual_seed(0)
    t0 = time.perf_counter(); r = hv.greedy(Xd, 5, seed=1234, prune_seed=4321); torch.cuda.synchronize()
    print(f"greedy q=5 {1e3 * (time.perf_counter() - t0):.2f} ms  picks {r.indices}")
