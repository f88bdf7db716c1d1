"""A/B in one process (interleaved rounds): fused posterior+qLogEI kernel vs posterior kernel + qLogEI kernel."""
import sys, time, math
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np, torch
from bench import synth_problem
from baybe_amd import engine, gp_spec
N, d, n, S = 1_000_000, 20, 512, 512
X, Xt, y = synth_problem(N, d, n, 0)
gp = engine.HipGP(0)
gp.set_model(gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d)), Xt, y)
gp.factorize(gp_spec.GPParams(np.full(d, math.exp(math.sqrt(2) - 3) * math.sqrt(d)), math.exp(-5.0), 0.0))
bf = gp.best_f(); z = engine.sobol_normal_base_samples(S, 1, 1234)[:, 0]
Xd = torch.from_numpy(X).cuda()
def fused():
    s, _, _ = gp.score_qlogei(Xd, z, bf, 1.0, want_posterior=False); return gp.topk(s, 8)
def split():
    m, v = gp.posterior(Xd); s = gp.qlogei(m, v, z, bf, 1.0); return gp.topk(s, 8)
for f in (fused, split): f(); f()
res = {"fused": [], "split": []}
for rnd in range(6):
    for name, f in (("fused", fused), ("split", split)):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): f()
        torch.cuda.synchronize(); res[name].append((time.perf_counter() - t0) / 10 * 1e3)
for k, v in res.items(): print(k, "ms/step median %.3f min %.3f" % (np.median(v), min(v)), [round(x, 3) for x in v])
