"""Three launches of the fused posterior on the bench shape (for rocprofv3 --pmc / --kernel-trace passes)."""
import math, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from bench import synth_problem
from baybe_amd import engine, gp_spec
N, d, n = 1_000_000, 20, 512
X, Xt, y = synth_problem(N, d, n, 0)
g = engine.HipGP(0)
g.set_model(gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d)), Xt, y)
g.factorize(gp_spec.GPParams(np.full(d, math.exp(math.sqrt(2) - 3) * math.sqrt(d)), math.exp(-5.0), 0.0))
Xd = torch.from_numpy(X).cuda()
for _ in range(3): g.posterior(Xd)
torch.cuda.synchronize()
