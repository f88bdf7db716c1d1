"""A/B of the fused posterior kernel forms for 512 < n <= 1024 in one process: windowed form (BBH_COOP=0) against the
two-sweep cooperative form (bbh_coop2.h, default); interleaved rounds, results compared.  Shapes: BASELINE configs[3]
(ICM over 4 tasks, 1e5 x 16, n = 1024) and plain Matérn-5/2 models at n = 1024 / 768 / 576."""
import math, os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import numpy as np, torch
from bench import synth_problem
from _problems import make_tl_problem
from baybe_amd import engine, gp_spec


def handle(flag, spec, Xt, y, params):
    os.environ["BBH_COOP"] = flag
    g = engine.HipGP(0)
    g.set_model(spec, Xt, y)
    g.factorize(params if params is not None else gp_spec.initial_params(spec))
    return g


cases = []
for N in (100_000, 300_000):
    X, Xt, y = make_tl_problem(N, 15, 256, T=4, seed=0)
    spec = gp_spec.GPSpec.baybe_default(16, np.zeros(16), np.ones(16), task_idx=15, n_tasks=4)
    cases.append((f"cfg4 ICM N={N} d=15+task n=1024", X, Xt, y, spec, None, 15, 1024))
for (N, d, n) in ((100_000, 20, 1024), (1_000_000, 20, 1024), (300_000, 20, 768), (300_000, 12, 576)):
    X, Xt, y = synth_problem(max(N, 4 * n), d, n, 0)
    spec = gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d))
    prm = gp_spec.GPParams(np.full(d, math.exp(math.sqrt(2) - 3) * math.sqrt(d)), math.exp(-5.0), 0.0)
    cases.append((f"plain N={N} d={d} n={n}", X[:N], Xt, y, spec, prm, d, n))

for name, X, Xt, y, spec, prm, d, n in cases:
    N = len(X)
    Xd = torch.from_numpy(X).cuda()
    gs = {f: handle(f, spec, Xt, y, prm) for f in ("0", "1")}
    out, forms = {}, {}
    for f, g in gs.items():
        m, v = g.posterior(Xd); m, v = g.posterior(Xd)
        out[f] = (m.cpu().numpy(), v.cpu().numpy()); forms[f] = g.posterior_kernel_form()
    dm = np.abs(out["0"][0] - out["1"][0]).max(); dv = np.abs(out["0"][1] - out["1"][1]).max()
    t = {f: [] for f in gs}
    for g in gs.values():
        g.timing(True); g.timing_read(reset=True)
    for rnd in range(5):
        for f, g in gs.items():
            for _ in range(10): g.posterior(Xd)
            torch.cuda.synchronize()
            ms, cnt = g.timing_read(reset=True)
            t[f].append(ms / cnt)
    fl = N * (n * n + 2 * n * d + 16 * n)  # the posterior kernel's own algorithmic flops (SURVEY §8d without the 16 S of qLogEI)
    a, b = np.median(t["0"]), np.median(t["1"])
    print(f"{name}: {forms['0']} {a:.3f} ms ({fl / (a * 1e-3) / 78.6e12:.3f})  {forms['1']} {b:.3f} ms ({fl / (b * 1e-3) / 78.6e12:.3f} of peak)"
          f"  max|dmean| {dm:.2e} max|dvar| {dv:.2e}", flush=True)
    for g in gs.values(): g.close()
