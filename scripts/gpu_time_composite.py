"""Composite kernels take the materialised-K* posterior path (bbh_launch_unfused_ext); how much slower is it than the
fused single-kernel path at the bench size (1e6 x 20 candidates, n = 512)?"""
import math, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np, torch
from bench import synth_problem
from baybe_amd import engine, gp_spec
from baybe_amd.kernels import MaternKernel, ProductKernel, AdditiveKernel, RBFKernel, ScaleKernel, apply_kernel_spec

N, d, n = 1_000_000, 20, 512
X, Xt, y = synth_problem(N, d, n, 0)
Xd = torch.from_numpy(X).cuda()
ls0 = math.exp(math.sqrt(2) - 3) * math.sqrt(d)
for name, kern in (("single Matern-5/2 (fused cooperative kernel)", None),
                   ("ProductKernel(Matern-5/2, RBF)", ProductKernel([MaternKernel(2.5), RBFKernel()])),
                   ("AdditiveKernel(Scale(Matern-5/2), Scale(RBF))", AdditiveKernel([ScaleKernel(MaternKernel(2.5)), ScaleKernel(RBFKernel())]))):
    spec = gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d))
    if kern is not None:
        apply_kernel_spec(spec, kern)
    g = engine.HipGP(0)
    g.set_model(spec, Xt, y)
    p = gp_spec.initial_params(spec)
    p.lengthscale = np.full(d, ls0)
    p.noise = math.exp(-5.0)
    if kern is not None:
        p.factor_ls = [np.full(d, 2.0 * ls0) for _ in p.factor_ls]
    g.factorize(p)
    g.posterior(Xd)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        g.posterior(Xd)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3 * 1e3
    g.greedy_qlogei(Xd, 3, seed=3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = g.greedy_qlogei(Xd, 3, seed=3)
    torch.cuda.synchronize(); dg = (time.perf_counter() - t0) * 1e3
    print(f"{name}: posterior {dt:.1f} ms ({N / dt * 1e3:.3g} cand/s, form {g.posterior_kernel_form()}), greedy batch of 3: {dg:.1f} ms")
    g.close()
