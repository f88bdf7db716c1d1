"""A/B of the fused posterior kernel forms in one process: windowed form (one wave per tile, BBH_COOP=0) against the
cooperative form (one workgroup per tile, BBH_COOP=1); interleaved rounds, results compared."""
import math, os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from bench import synth_problem
from baybe_amd import engine, gp_spec

def handle(flag, d, Xt, y):
    os.environ["BBH_COOP"] = flag
    g = engine.HipGP(0)
    g.set_model(gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d)), Xt, y)
    g.factorize(gp_spec.GPParams(np.full(d, math.exp(math.sqrt(2) - 3) * math.sqrt(d)), math.exp(-5.0), 0.0))
    return g

for (N, d, n) in ((1_000_000, 20, 512), (1_000_000, 20, 500), (500_000, 18, 256), (500_000, 20, 128), (300_000, 20, 64), (1003, 20, 320)):
    X, Xt, y = synth_problem(max(N, 4 * n), d, n, 0)
    X = X[:N]
    Xd = torch.from_numpy(X).cuda()
    gs = {f: handle(f, d, Xt, y) for f in ("0", "2")}
    out = {}
    for f, g in gs.items():
        m, v = g.posterior(Xd); m, v = g.posterior(Xd)
        out[f] = (m.cpu().numpy(), v.cpu().numpy())
    dm = np.abs(out["0"][0] - out["2"][0]).max(); dv = np.abs(out["0"][1] - out["2"][1]).max()
    t = {f: [] for f in gs}
    for rnd in range(5):
        for f, g in gs.items():
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): g.posterior(Xd)
            torch.cuda.synchronize(); t[f].append((time.perf_counter() - t0) / 10 * 1e3)
    fl = N * (n * n + 2 * n * d + 16 * n + 16 * 512)
    print(f"N={N} d={d} n={n}: windowed {np.median(t['0']):.3f} ms  coop {np.median(t['2']):.3f} ms "
          f"({fl / (np.median(t['2']) * 1e-3) / 78.6e12:.3f} of peak)  max|dmean| {dm:.2e} max|dvar| {dv:.2e}", flush=True)
    for g in gs.values(): g.close()
