"""A/B of the pending-points qLogEI kernel (register-resident form vs the generic LDS form): greedy batch of 5
on 1e6 x 20 candidates, n = 512; same batch, step values equal to ~1e-13."""
import os, sys, time, math
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np, torch
from bench import synth_problem
from baybe_amd import engine, gp_spec
N, d, n = 1_000_000, 20, 512
X, Xt, y = synth_problem(N, d, n, 0)
Q = int(sys.argv[1]) if len(sys.argv) > 1 else 5
out = {}
for form in ("0", "1"):
    os.environ["BBH_PENDING_LDS"] = form
    g = engine.HipGP(0)
    g.set_model(gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d)), Xt, y)
    g.factorize(gp_spec.GPParams(np.full(d, math.exp(math.sqrt(2) - 3) * math.sqrt(d)), math.exp(-5.0), 0.0))
    Xd = torch.from_numpy(X).cuda()
    g.greedy_qlogei(Xd, Q, seed=11)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = g.greedy_qlogei(Xd, Q, seed=11)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    out[form] = r
    print(f"BBH_PENDING_LDS={form}: greedy batch of {Q}: {dt:.1f} ms  -> {list(r.indices)}")
    g.close()
a, b = out["0"], out["1"]
print("identical indices:", list(a.indices) == list(b.indices), " max |value difference|:", float(np.max(np.abs(np.asarray(a.values) - np.asarray(b.values)))))
