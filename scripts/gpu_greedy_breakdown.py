"""Wall-clock breakdown of one greedy batch of 5 on the bench shape (1e6 x 20, n = 512): which host calls the 2.8 ms between
the device's 17.9 ms and the 20.7 ms wall time go to.  Every engine method is wrapped with a synchronising timer."""
import math, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from bench import synth_problem
from baybe_amd import engine, gp_spec

N, d, n = (int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000), 20, 512
X, Xt, y = synth_problem(N, d, n, 0)
g = engine.HipGP(0)
g.set_model(gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d)), Xt, y)
g.factorize(gp_spec.GPParams(np.full(d, math.exp(math.sqrt(2) - 3) * math.sqrt(d)), math.exp(-5.0), 0.0))
Xd = torch.from_numpy(X).cuda()
best_f = g.best_f()
acc = {}
def wrap(name):
    fn = getattr(g, name)
    def inner(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
        return r
    setattr(g, name, inner)
for _ in range(20): g.greedy_qlogei(Xd, 5, seed=1234, best_f=best_f)
ts = []
for _ in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter(); g.greedy_qlogei(Xd, 5, seed=1234, best_f=best_f); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
print(f"N={N}: greedy q=5 median {np.median(ts):.3f} ms (min {min(ts):.3f})")
torch.cuda.synchronize(); t0 = time.perf_counter(); g.greedy_qlogei(Xd, 5, seed=1234, best_f=best_f); torch.cuda.synchronize()
print(f"plain wall {1e3 * (time.perf_counter() - t0):.2f} ms")
for name in ("posterior", "mc_acq", "set_pending", "cross_cov", "topk", "argmax", "qlogei_topk", "qlogei_pending_big"):
    wrap(name)
orig = engine.sobol_normal_base_samples
def sob(*a, **k):
    t0 = time.perf_counter(); r = orig(*a, **k); acc["sobol (host)"] = acc.get("sobol (host)", 0.0) + (time.perf_counter() - t0) * 1e3; return r
engine.sobol_normal_base_samples = sob
torch.cuda.synchronize(); t0 = time.perf_counter(); g.greedy_qlogei(Xd, 5, seed=1234, best_f=best_f); torch.cuda.synchronize()
tot = 1e3 * (time.perf_counter() - t0)
print(f"instrumented wall {tot:.2f} ms; " + ", ".join(f"{k} {v:.2f}" for k, v in acc.items()) + f"; other {tot - sum(acc.values()):.2f}")
