"""Composite / rational-quadratic / piecewise-polynomial models: materialised-K* posterior path (BBH_COOP=0) against the
cooperative form with the generic kernel-value production (bbh_coopg.h, default) - results compared, kernel times from HIP
events.  Bench size: 1e6 x 20 candidates, n = 512; and a small model (1e5 x 8, n = 100)."""
import math, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np, torch
from bench import synth_problem
from baybe_amd import engine, gp_spec
from baybe_amd.kernels import (AdditiveKernel, MaternKernel, PiecewisePolynomialKernel, ProductKernel, RBFKernel, RQKernel,
                               ScaleKernel, apply_kernel_spec)

KERNELS = (("ProductKernel(Matern-5/2, RBF)", lambda: ProductKernel([MaternKernel(2.5), RBFKernel()])),
           ("AdditiveKernel(Scale(Matern-5/2), Scale(RBF))", lambda: AdditiveKernel([ScaleKernel(MaternKernel(2.5)), ScaleKernel(RBFKernel())])),
           ("ProductKernel(Matern-3/2, Matern-5/2, RBF)", lambda: ProductKernel([MaternKernel(1.5), MaternKernel(2.5), RBFKernel()])),
           ("Scale(RQKernel)", lambda: ScaleKernel(RQKernel())),
           ("PiecewisePolynomialKernel(q=2)", lambda: PiecewisePolynomialKernel(q=2)))
for (N, d, n) in ((1_000_000, 20, 512), (100_000, 8, 100)):
    X, Xt, y = synth_problem(N, d, n, 0)
    Xd = torch.from_numpy(X).cuda()
    ls0 = math.exp(math.sqrt(2) - 3) * math.sqrt(d)
    for name, mk in KERNELS:
        out, ms, forms = {}, {}, {}
        for flag in ("0", "1"):
            os.environ["BBH_COOP"] = flag
            spec = gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d))
            apply_kernel_spec(spec, mk())
            g = engine.HipGP(0)
            g.set_model(spec, Xt, y)
            p = gp_spec.initial_params(spec)
            p.lengthscale = np.full(d, (3.0 if "Piecewise" in name else 1.0) * ls0)
            p.noise = math.exp(-5.0)
            if p.factor_ls is not None:
                p.factor_ls = [np.full(d, 2.0 * ls0) for _ in p.factor_ls]
            g.factorize(p)
            m, v = g.posterior(Xd)
            out[flag] = (m.cpu().numpy(), v.cpu().numpy()); forms[flag] = g.posterior_kernel_form()
            g.timing(True); g.timing_read(reset=True)
            torch.cuda.synchronize()
            import time
            t0 = time.perf_counter()
            for _ in range(3): g.posterior(Xd)
            torch.cuda.synchronize()
            ms[flag] = (time.perf_counter() - t0) / 3 * 1e3
            g.close()
        dm = np.abs(out["0"][0] - out["1"][0]).max(); dv = np.abs(out["0"][1] - out["1"][1]).max()
        print(f"N={N} d={d} n={n} {name}: {forms['0']} {ms['0']:.2f} ms  {forms['1']} {ms['1']:.2f} ms  max|dmean| {dm:.2e} max|dvar| {dv:.2e}"
              f" (std y {np.std(y):.2f})", flush=True)
