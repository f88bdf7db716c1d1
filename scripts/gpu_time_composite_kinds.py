"""Posterior pass (1e6 x 20, n = 512) of every kernel of baybe/kernels/basic.py that has no software-pipelined instantiation -
Linear, Polynomial, Periodic, RQ, piecewise polynomial - and of a sum with a dot-product factor, on the cooperative form with the
generic production (csrc/bbh_coopg.h) against the materialised-K* path (``unfused=True``).  d = 14 for the periodic kernel (its
cos / sin features need 2 d + 1 <= 32)."""
import os, sys, time, math
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from bench import synth_problem
from baybe_amd import engine, gp_spec
from baybe_amd.kernels import (AdditiveKernel, GammaPrior, LinearKernel, MaternKernel, PeriodicKernel, PolynomialKernel, ProductKernel, RBFKernel, RQKernel,
                               ScaleKernel, apply_kernel_spec)


def run(name, kern, d, N=1_000_000, n=512):
    class Space:
        comp_rep_columns = tuple(f"x{j}" for j in range(d))

    X, Xt, y = synth_problem(N, d, n, 0)
    Xd = torch.from_numpy(X).cuda()
    spec = gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d))
    apply_kernel_spec(spec, kern, Space())
    g = engine.HipGP(0)
    g.set_model(spec, Xt, y)
    p = gp_spec.initial_params(spec)
    p.noise = 1e-2
    g.factorize(p)
    out = {}
    for unfused in (False, True):
        Xs = Xd if not unfused else Xd[:200_000]
        g.posterior(Xs, unfused=unfused)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): g.posterior(Xs, unfused=unfused)
        torch.cuda.synchronize()
        out[unfused] = ((time.perf_counter() - t0) / 3 * 1e3 * (N / len(Xs)), g.posterior_kernel_form())
    # steps >= 2 of a greedy batch: one cross-covariance pass (3 pending points) and the whole greedy batch of 5
    g.set_pending(X[[11, 5000, 90000]])
    g.cross_cov(Xd)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): g.cross_cov(Xd)
    torch.cuda.synchronize(); t_cross = (time.perf_counter() - t0) / 3 * 1e3
    g.set_pending(None)
    g.greedy_qlogei(Xd, 5, seed=3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    g.greedy_qlogei(Xd, 5, seed=3)
    torch.cuda.synchronize(); t_greedy = (time.perf_counter() - t0) * 1e3
    print(f"{name:28s} d={d}: fused {out[False][0]:.2f} ms [{out[False][1]}]   materialised {out[True][0]:.1f} ms per 1e6 [{out[True][1]}]   "
          f"cross pass {t_cross:.2f} ms   greedy batch of 5: {t_greedy:.1f} ms   (BBH_COOPG_CROSS={os.environ.get('BBH_COOPG_CROSS', '1')})", flush=True)
    g.close()


run("Linear (scaled)", ScaleKernel(LinearKernel(GammaPrior(2, 1)), GammaPrior(2, 0.5)), 20)
run("Polynomial(2)", PolynomialKernel(2, GammaPrior(2, 2)), 20)
run("Periodic (scaled)", ScaleKernel(PeriodicKernel(GammaPrior(3, 2), 1.0, GammaPrior(4, 3), 1.5), GammaPrior(2, 0.5)), 14)
run("RQ", RQKernel(GammaPrior(3, 1)), 20)
run("Matern x RBF (product)", ProductKernel([MaternKernel(2.5, GammaPrior(3, 1)), ScaleKernel(RBFKernel(), GammaPrior(2, 0.5))]), 20)
run("Matern + Linear (sum)", AdditiveKernel([MaternKernel(2.5, GammaPrior(3, 1)), ScaleKernel(LinearKernel(GammaPrior(3, 2)))]), 20)
run("(Matern * Matern) + (Matern + RBF)", AdditiveKernel([ProductKernel([MaternKernel(2.5, GammaPrior(3, 1)), MaternKernel(1.5)]),
                                                       AdditiveKernel([ScaleKernel(MaternKernel(2.5), GammaPrior(2, 0.5)), RBFKernel()])]), 20)
