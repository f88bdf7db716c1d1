"""How the cooperative posterior kernel's time grows with the number of workgroup rounds (512 resident workgroups of 16 candidates at
n = 512): t(r) for r rounds separates a fixed ramp (cold L2 at kernel start, lock-step reads of the operand stream) from the per-round
time.  Wall time over 20 back-to-back launches (launch overhead overlaps the previous kernel)."""
import sys, time, math
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np, torch
from bench import synth_problem
from baybe_amd import engine, gp_spec

for (d, n, per_round) in ((20, 512, 512 * 16), (15, 256, 1024 * 16)):
    X, Xt, y = synth_problem(per_round * 64, d, n, 0)
    g = engine.HipGP(0)
    g.set_model(gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d)), Xt, y)
    g.factorize(gp_spec.GPParams(np.full(d, math.exp(math.sqrt(2) - 3) * math.sqrt(d)), math.exp(-5.0), 0.0))
    Xd = torch.from_numpy(X).cuda()
    for r in (0.25, 0.5, 1, 2, 3, 4, 8, 15.26, 16, 32, 64):
        N = int(per_round * r)
        Xr = Xd[:N]
        m = torch.empty(N, dtype=torch.float64, device="cuda"); v = torch.empty_like(m)
        for _ in range(3): g.posterior(Xr, out=(m, v))
        ts = []
        for rnd in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): g.posterior(Xr, out=(m, v))
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 20 * 1e6)
        print(f"n={n} d={d} rounds={r:6.2f} N={N:8d}: {np.median(ts):8.1f} us  per round {np.median(ts)/max(r,1):7.1f} us  form={g.posterior_kernel_form()}", flush=True)
