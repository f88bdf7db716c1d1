"""Fused posterior per kernel kind (1e6 x 20, n = 512): the pipelined instantiations cover Matérn-5/2, -3/2
and RBF; Matérn-1/2 and RBF with a task / outputscale table take the plain form."""
import sys, time, math
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from bench import synth_problem
from baybe_amd import engine, gp_spec
N, d, n = 1_000_000, 20, 512
X, Xt, y = synth_problem(N, d, n, 0)
Xd = torch.from_numpy(X).cuda()
for kind in ("matern52", "matern32", "rbf", "matern12"):
    g = engine.HipGP(0)
    g.set_model(gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d), kernel=kind), Xt, y)
    g.factorize(gp_spec.GPParams(np.full(d, math.exp(math.sqrt(2) - 3) * math.sqrt(d)), math.exp(-5.0), 0.0))
    g.posterior(Xd); g.posterior(Xd)
    t = []
    for rnd in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): g.posterior(Xd)
        torch.cuda.synchronize(); t.append((time.perf_counter() - t0) / 5 * 1e3)
    print(f"{kind:9s}: posterior {np.median(t):.3f} ms")
    g.close()
