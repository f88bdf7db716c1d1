"""RFF kernel (csrc/bbh_rff.hip): cost of a scoring pass over 1e6 x 20 candidates at n = 512 (<= 8 ms), of a fit
evaluation and of a whole fit, for num_samples in {5, 16, 32, 64}; greedy batch of 5; the RBF model of the same size next to it."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import torch
from bench import synth_problem
from baybe_amd import engine, gp_spec
from baybe_amd.kernels import GammaPrior, RBFKernel, RFFKernel, ScaleKernel, apply_kernel_spec

N, d, n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 20, 512
X, Xt, y = synth_problem(N, d, n, 0)
Xd = torch.from_numpy(X).cuda()

class Space:
    comp_rep_columns = tuple(f"x{j}" for j in range(d))

def run(tag, kern):
    spec = gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d))
    apply_kernel_spec(spec, kern, Space())
    g = engine.HipGP(0)
    torch.manual_seed(0)
    g.set_model(spec, Xt, y)
    th = gp_spec.theta_from_params(g.spec, gp_spec.initial_params(g.spec))
    for _ in range(5): g._data_term_theta(th)
    t0 = time.perf_counter()
    for _ in range(40): g._data_term_theta(th)
    ev = (time.perf_counter() - t0) / 40 * 1e6
    t0 = time.perf_counter(); info = g.fit(); tf = (time.perf_counter() - t0) * 1e3
    m = torch.empty(N, dtype=torch.float64, device="cuda"); v = torch.empty_like(m)
    for _ in range(3): g.posterior(Xd, out=(m, v))
    torch.cuda.synchronize()
    g.timing(True, ["posterior"])
    t0 = time.perf_counter()
    for _ in range(10): g.posterior(Xd, out=(m, v))
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 10 * 1e3
    ms, cnt = g.timing_read(family="posterior")
    g.timing(False)
    t0 = time.perf_counter(); res = g.greedy_qlogei(Xd, 5, seed=3); tg = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter(); res = g.greedy_qlogei(Xd, 5, seed=3); tg = (time.perf_counter() - t0) * 1e3
    print(f"{tag:28s} fit evaluation {ev:7.1f} us, fit {tf:7.1f} ms / {info.nfev} evaluations; posterior {N:.0e} rows: {ms / max(cnt, 1):6.3f} ms "
          f"(HIP events; wall {wall:6.3f}) form {g.posterior_kernel_form()}; greedy q = 5: {tg:6.2f} ms", flush=True)
    g.close()

for D in (5, 16, 32, 64, 100, 128, 256):  # (beyond 64: the chunked candidate form and the blocked m x m factorisation, round 6)
    run(f"rff D={D}", ScaleKernel(RFFKernel(D, GammaPrior(3, 2)), GammaPrior(2, 0.5)))
run("rbf (n x n form)", ScaleKernel(RBFKernel(GammaPrior(3, 2)), GammaPrior(2, 0.5)))
