"""Fused posterior at larger input dimension (descriptor-style encodings): d = 40 and 60, n = 512."""
import sys, time, math
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from bench import synth_problem
from baybe_amd import engine, gp_spec
for (N, d, n) in ((500_000, 40, 512), (500_000, 60, 512)):
    X, Xt, y = synth_problem(N, d, n, 0)
    g = engine.HipGP(0)
    g.set_model(gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d)), Xt, y)
    g.factorize(gp_spec.GPParams(np.full(d, math.exp(math.sqrt(2) - 3) * math.sqrt(d)), math.exp(-5.0), 0.0))
    Xd = torch.from_numpy(X).cuda()
    g.posterior(Xd); g.posterior(Xd)
    t = []
    for rnd in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): g.posterior(Xd)
        torch.cuda.synchronize(); t.append((time.perf_counter() - t0) / 5 * 1e3)
    fl = N * (n * n + 2 * n * d + 16 * n) / (np.median(t) * 1e-3) / 1e12
    print(f"N={N} d={d} n={n}: posterior {np.median(t):.3f} ms  ({fl:.1f} TFLOP/s algorithmic)")
