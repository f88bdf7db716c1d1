"""Profiling target: the register-resident small-model posterior kernel alone (n = 64, d = 10 and n = 32, d = 6 at 1e6 candidates)."""
import math, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from bench import synth_problem
from baybe_amd import engine, gp_spec

for d, n in ((10, 64), (6, 32)):
    Xall, Xt, y = synth_problem(1_000_000, d, n, 0)
    g = engine.HipGP(0)
    g.set_model(gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d)), Xt, y)
    g.factorize(gp_spec.GPParams(np.full(d, math.exp(math.sqrt(2) - 3) * math.sqrt(d)), math.exp(-5.0), 0.0))
    Xd = torch.from_numpy(Xall).cuda()
    for _ in range(5):
        g.posterior(Xd)
    torch.cuda.synchronize()
    g.close()
