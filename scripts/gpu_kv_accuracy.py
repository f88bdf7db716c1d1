"""Accuracy of the staged Matérn-5/2 evaluation inside the pipelined fused kernel: posterior of the
bench workload with BBH_PIPELINE=1 (rsq/Goldschmidt sqrt + Taylor exp) against BBH_PIPELINE=0 (libm
sqrt/exp in the same kernel structure).  Prints max abs differences scaled by the output scale."""
import math, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np, torch
from bench import synth_problem
from baybe_amd import engine, gp_spec

def run(pipe, N, d, n):
    os.environ["BBH_PIPELINE"] = pipe
    X, Xt, y = synth_problem(N, d, n, 0)
    g = engine.HipGP(0)
    g.set_model(gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d)), Xt, y)
    g.factorize(gp_spec.GPParams(np.full(d, math.exp(math.sqrt(2) - 3) * math.sqrt(d)), math.exp(-5.0), 0.0))
    m, v = g.posterior(torch.from_numpy(X).cuda())
    return m.cpu().numpy(), v.cpu().numpy(), float(np.std(y))

for (N, d, n) in ((1_000_000, 20, 512), (200_000, 6, 1024), (100_000, 15, 256)):
    m1, v1, s = run("1", N, d, n)
    m0, v0, _ = run("0", N, d, n)
    print(f"N={N} d={d} n={n}: max|dmean|/ysd = {np.max(np.abs(m1 - m0)) / s:.3e}   "
          f"max|dvar|/ysd^2 = {np.max(np.abs(v1 - v0)) / s**2:.3e}   min var/ysd^2 = {v0.min() / s**2:.3e}")
