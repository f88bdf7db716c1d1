"""Times the fused posterior kernel on the bench workload (median of 6 x 10 launches)."""
import sys, time, math
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np, torch
from bench import synth_problem
from baybe_amd import engine, gp_spec
for (N, d, n) in ((1_000_000, 20, 512), (100_000, 15, 256)):
    X, Xt, y = synth_problem(N, d, n, 0)
    g = engine.HipGP(0)
    g.set_model(gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d)), Xt, y)
    g.factorize(gp_spec.GPParams(np.full(d, math.exp(math.sqrt(2) - 3) * math.sqrt(d)), math.exp(-5.0), 0.0))
    Xd = torch.from_numpy(X).cuda()
    g.posterior(Xd); g.posterior(Xd)
    t = []
    for rnd in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): g.posterior(Xd)
        torch.cuda.synchronize(); t.append((time.perf_counter() - t0) / 10 * 1e3)
    print(f"N={N} d={d} n={n}: posterior {np.median(t):.3f} ms (min {min(t):.3f})")
